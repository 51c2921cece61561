#!/usr/bin/env python3
"""profiles/collect_rescale.sh's CSVs -> profiles/<tag>_rescale_kernel_stats.csv, <tag>_rescale_bench.json (with a
roofline per kernel) and <tag>_rescale_summary.md.

The graph vote moves bytes (HBM roof).  flat_selection, the RANSAC plane fit and triangle_batch are fp64 arithmetic:
their roof is the vector fp64 rate, 78.6 TFLOP/s on MI355X (half the 157.3 TFLOP/s fp32 vector rate of
MI355X_MICROARCH.md) = 39.3 T lane-instructions/s with an FMA counted once — against it stand the fp64 VALU
instructions the kernels really issue (SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64 x 64 lanes)."""
import csv, glob, json, os, shutil, statistics, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
out = os.path.join(ROOT, "gpurun_out")
FP64_LANE_INSTR_PEAK = 39.3e12
stats = os.path.join(out, tag + "_rescale_stats", "bench_kernel_stats.csv")
shutil.copy(stats, os.path.join(HERE, tag + "_rescale_kernel_stats.csv"))
rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
bench = json.loads(open(os.path.join(out, tag + "_rescale_bench.json")).read().strip().splitlines()[-1])
F = bench["frames"]
pmc = {}
for d in sorted(glob.glob(os.path.join(out, tag + "_rescale_pmc*"))):
    f = os.path.join(d, "bench_counter_collection.csv")
    if os.path.isfile(f):
        for r in csv.DictReader(open(f)):
            pmc.setdefault(r["Kernel_Name"], {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
names = {"graph_inliers": "graph_inliers_kernel", "flat_selection": "flat_selection_kernel", "ransac_plane": "ransac_plane_kernel",
         "triangle_batch": "triangle_batch_kernel"}
roof = {}
lines = ["# rocprofv3 summary %s — `python profiles/bench_rescale.py` (%d frames x %.0f features per launch)" % (tag, F, bench["features_per_frame"]), "",
         "| kernel | rocprof avg ms | HIP-event ms (same run) | algorithmic GB/s (of 8 TB/s) | fp64 VALU lane-instr/frame | fp64 rate (of 39.3 T/s) | 2xFETCH+WRITE per frame | bound |",
         "|---|---|---|---|---|---|---|---|"]
for key, kname in names.items():
    krow = [r for n, r in rows.items() if kname in n]
    avg_ms = float(krow[0]["AverageNs"]) / 1e6 if krow else float("nan")
    c = {}
    for n, cs in pmc.items():
        if kname in n:
            c = {k: statistics.median(v) for k, v in cs.items()}
    f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64")) * 64.0
    rate = f64 / (avg_ms * 1e-3) if avg_ms == avg_ms and f64 else float("nan")
    gbps = bench["algorithmic_GBps_per_kernel"][key]
    traffic = (2.0 * c["FETCH_SIZE"] + c.get("WRITE_SIZE", 0.0)) * 1024.0 / F if "FETCH_SIZE" in c else float("nan")
    hbm_frac, f64_frac = gbps / 8000.0, rate / FP64_LANE_INSTR_PEAK
    bound = "hbm" if not (f64_frac > hbm_frac) else "fp64_valu"
    roof[key] = {"bound": bound, "achieved": gbps if bound == "hbm" else rate / 1e12, "peak": 8000.0 if bound == "hbm" else FP64_LANE_INSTR_PEAK / 1e12,
                 "unit": "GB/s" if bound == "hbm" else "T fp64 lane-instr/s", "frac": hbm_frac if bound == "hbm" else f64_frac,
                 "hbm_frac": hbm_frac, "fp64_valu_frac": f64_frac, "kernel_ms_rocprof": avg_ms, "kernel_ms_events": bench["kernel_ms"][key],
                 "fp64_lane_instr_per_frame": f64 / F, "valu_wave_instr_per_frame": c.get("SQ_INSTS_VALU", float("nan")) / F,
                 "traffic_bytes_per_frame": traffic}
    lines.append("| `%s` | %.4f | %.4f | %.0f (%.3f) | %.4g | %.3g T/s (%.3f) | %.0f B | %s |"
                 % (kname, avg_ms, bench["kernel_ms"][key], gbps, hbm_frac, f64 / F, rate / 1e12, f64_frac, traffic, bound))
bench["roofline_per_kernel"] = roof
json.dump(bench, open(os.path.join(HERE, tag + "_rescale_bench.json"), "w"), indent=1)
lines += ["", "fp64 peak: 78.6 TFLOP/s vector fp64 = 39.3 T lane-instructions/s (FMA = one instruction); instruction counts from "
          "`SQ_INSTS_VALU_{ADD,MUL,FMA,TRANS}_F64` (wave instructions x 64 lanes).  The three stages together: %.2f M frames/s." % (bench["value"] / 1e6)]
open(os.path.join(HERE, tag + "_rescale_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
