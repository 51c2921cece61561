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
def short(k):
    """A kernel's name without its argument list ('(anonymous namespace)::' is not one)."""
    k = k.replace("(anonymous namespace)::", "")
    return k.split("(")[0][-62:]


per = defaultdict(lambda: [0, 0])
for s, e, k in rows:
    per[short(k)][0] += e - s; per[short(k)][1] += 1
for k, (t, c) in sorted(per.items(), key=lambda kv: -kv[1][0])[:10]:
    print("  %-62s %8.2f ms  %5d launches" % (k, t / 1e6, c))
gaps.sort(reverse=True)
print("idle %.2f ms in %d gaps; the largest (us, at ms):" % (sum(g for g, _ in gaps) / 1e6, len(gaps)), ", ".join("%.0f@%.1f" % (g / 1e3, a / 1e6) for g, a in gaps[:12]))
small = sum(g for g, _ in gaps if g < 50e3)
print("gaps below 50 us: %.2f ms in total" % (small / 1e6))

# the exact path: where a call's time goes around the Qhull replay — the head before the first replay starts (pack +
# upload of the first chunk), the union of the big replays (the machine's real work), the tail after the last big replay ends (vote,
# stand-in triangulation, product kernels, the list replay of the exact pass's frames: one wavefront per frame, a whole run long)
rep = [(s_, e_) for s_, e_, k in rows if "qhull_rows_kernel" in k]
if rep:
    # (the list walk is the instantiation whose last template argument is true)
    lst = [(s_, e_) for s_, e_, k in rows if "qhull_rows_kernel" in k and k.replace(" ", "").split("(")[0].rstrip(">").endswith(",true")]
    big = [r for r in rep if r not in lst]
    u, cs, ce = 0, big[0][0], big[0][1]
    for s_, e_ in sorted(big)[1:]:
        if s_ > ce:
            u += ce - cs; cs, ce = s_, e_
        else:
            ce = max(ce, e_)
    u += ce - cs
    print("qhull replay: %d big launches (union %.2f ms, sum %.2f ms), %d list launches (sum %.2f ms); head before the first replay %.2f ms, "
          "tail after the last big replay %.2f ms" % (len(big), u / 1e6, sum(e_ - s_ for s_, e_ in big) / 1e6, len(lst), sum(e_ - s_ for s_, e_ in lst) / 1e6,
                                                     (min(s_ for s_, _ in big) - t0) / 1e6, (t1 - max(e_ for _, e_ in big)) / 1e6))
