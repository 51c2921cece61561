#!/usr/bin/env python3
"""How busy the GPU is during the device-triangulation batch call: union of the kernels' intervals over the last call of
    cd /tmp; rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/e2e_tl -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 32768 2000
    python profiles/e2e_gpu_busy.py gpurun_out/e2e_tl/e2e_kernel_trace.csv"""
import csv, sys
from collections import defaultdict
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last call: the second half of the launches (the script runs the batch twice)
rows = rows[len(rows) // 2:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
busy, cur_s, cur_e = 0, rows[0][0], rows[0][1]
gaps = []
for s, e, _ in rows[1:]:
    if s > cur_e:
        busy += cur_e - cur_s
        gaps.append((s - cur_e, cur_e - t0))
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
print("kernels %d, span %.2f ms, busy %.2f ms (%.1f %%)" % (len(rows), (t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0)))
per = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    per[k.split("(")[0][-60:]][0] += e - s; per[k.split("(")[0][-60:]][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:10]:
    print("  %-62s %8.2f ms  %5d launches" % (k, t / 1e6, c))
gaps.sort(reverse=True)
print("idle %.2f ms in %d gaps; the largest (us, at ms):" % (sum(g for g, _ in gaps) / 1e6, len(gaps)), ", ".join("%.0f@%.1f" % (g / 1e3, a / 1e6) for g, a in gaps[:12]))
small = sum(g for g, _ in gaps if g < 50e3)
print("gaps below 50 us: %.2f ms in total" % (small / 1e6))
