#!/usr/bin/env python3
"""Kernel timeline of the last steps of a rocprofv3 --kernel-trace run: start/end (us, relative), queue, name.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o bench -- python3 bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-e2e
    python profiles/timeline.py gpurun_out/tl/bench_kernel_trace.csv [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n:]
t0 = int(rows[0]["Start_Timestamp"])
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    print("%10.1f %10.1f  %8.1f us  q%-3s %s" % (s, e, e - s, r.get("Queue_Id", "?"), r["Kernel_Name"][:70]))
