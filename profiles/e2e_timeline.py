#!/usr/bin/env python3
"""Chunk by chunk: when the uploads and the kernels of the device-triangulation batch call ran (the last call of
    cd /tmp; rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/e2e_tl -o e2e -- python3 $R/profiles/e2e_gpu_profile.py 32768 2000 [rescale]
    python profiles/e2e_timeline.py gpurun_out/e2e_tl)
A chunk = the kernels from one first-triangulation launch (delaunay_kernel without seeds comes first) to the next."""
import csv, glob, os, sys
d = sys.argv[1]
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
mt = glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True)
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(kt)))
M = []
if mt:
    for r in csv.DictReader(open(mt[0])):
        M.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Direction", r.get("Name", "")), 0))
M.sort()
# the last call: after the largest gap between kernels in the second half of the trace
half = K[len(K) // 2:]
gaps = [(half[i + 1][0] - max(x[1] for x in half[:i + 1][-8:]), i) for i in range(len(half) - 1)]
g, i = max(gaps)
K = half[i + 1:] if g > 2e6 else half
t0 = min(K[0][0], min([m[0] for m in M if m[0] > K[0][0] - 20e6] or [K[0][0]]))
M = [m for m in M if m[0] >= t0 - 1e6]
t0 = min(t0, M[0][0]) if M else t0
# chunks: a delaunay launch that follows a non-delaunay kernel (or starts the call) and is the FIRST of its pair
chunks, cur = [], None
prev_dt = 0
for s, e, k in K:
    is_dt = "delaunay_kernel" in k
    if is_dt and prev_dt % 2 == 0:
        cur = {"k0": s, "k1": e, "busy": 0, "dt": 0}
        chunks.append(cur)
    if is_dt: prev_dt += 1
    if cur is None: continue
    cur["k1"] = max(cur["k1"], e); cur["busy"] += e - s
    if is_dt: cur["dt"] += e - s
print("call: first upload/kernel at 0, last kernel ends at %.2f ms; %d chunks" % ((max(x[1] for x in K) - t0) / 1e6, len(chunks)))
big = [m for m in M if m[1] - m[0] > 200e3]
print("uploads longer than 0.2 ms: %d, %.2f ms in total" % (len(big), sum(m[1] - m[0] for m in big) / 1e6))
print("%5s %22s %22s %9s %9s %9s" % ("chunk", "upload (start-end ms)", "kernels (start-end ms)", "busy ms", "dt ms", "idle before"))
last_end = None
bi = 0
for n, c in enumerate(chunks):
    ups = [m for m in big if m[1] <= c["k0"] + 1e5 and (n == 0 or m[1] > chunks[n - 1]["k0"])]
    u = "%.2f-%.2f" % ((min(m[0] for m in ups) - t0) / 1e6, (max(m[1] for m in ups) - t0) / 1e6) if ups else "-"
    idle = (c["k0"] - last_end) / 1e6 if last_end is not None else (c["k0"] - t0) / 1e6
    print("%5d %22s %22s %9.2f %9.2f %9.2f" % (n, u, "%.2f-%.2f" % ((c["k0"] - t0) / 1e6, (c["k1"] - t0) / 1e6), c["busy"] / 1e6, c["dt"] / 1e6, idle))
    last_end = c["k1"]
