#!/usr/bin/env python3
"""profiles/collect_delaunay.sh's CSVs -> profiles/<tag>_delaunay_kernel_stats.csv, <tag>_delaunay_bench.json and
<tag>_delaunay_summary.md.

delaunay_kernel is instruction-bound: a frame's points are read from HBM once (16 B per point) and its rows written once
(12 B per row, ~2 rows per point) — 40 B per point against ~5 000 vector instructions per point-lane.  Its roof is the
vector-instruction issue rate of the CUs: 256 CUs x 4 SIMDs, one wave-instruction per 4 cycles per SIMD for the fp64 /
VOP3 class this kernel consists of (MI355X_MICROARCH.md, per-instruction constants: v_fma_f32 4 cycles for one wave,
fp64 the same issue slot; measured 4.4 in profiles/micro/valu_rates.hip) at the 2.4 GHz peak clock = 6.1e11 wave-instructions/s;
beside it the fp64 arithmetic roof (39.3 T fp64 lane-instructions/s)."""
import csv, glob, json, os, shutil, statistics, sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
out = os.path.join(ROOT, "gpurun_out")
ISSUE_PEAK = 256 * 4 * 2.4e9 / 4.0          # wave-instructions per second
FP64_LANE_INSTR_PEAK = 39.3e12
stats = os.path.join(out, tag + "_dt_stats", "bench_kernel_stats.csv")
shutil.copy(stats, os.path.join(HERE, tag + "_delaunay_kernel_stats.csv"))
rows = {r["Name"]: r for r in csv.DictReader(open(stats))}
bench = json.loads(open(os.path.join(out, tag + "_dt_bench.json")).read().strip().splitlines()[-1])
F, n = bench["sets"], bench["points_per_set"]
krow = [r for k, r in rows.items() if "delaunay_kernel" in k][0]
avg_ms = float(krow["AverageNs"]) / 1e6
c = {}
for d in sorted(glob.glob(os.path.join(out, tag + "_dt_pmc*"))):
    f = os.path.join(d, "bench_counter_collection.csv")
    if os.path.isfile(f):
        acc = {}
        for r in csv.DictReader(open(f)):
            if "delaunay_kernel" in r["Kernel_Name"]:
                acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
        for k, v in acc.items():
            c[k] = statistics.median(v)
g = lambda k: c.get(k, float("nan"))
valu = g("SQ_INSTS_VALU")
f64 = sum(c.get(k, 0.0) for k in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_TRANS_F64"))
issue_rate = valu / (avg_ms * 1e-3)
roof = {"bound": "valu_issue", "achieved": issue_rate / 1e9, "peak": ISSUE_PEAK / 1e9, "unit": "G wave-instr/s", "frac": issue_rate / ISSUE_PEAK,
        "fp64_lane_instr_frac": f64 * 64 / (avg_ms * 1e-3) / FP64_LANE_INSTR_PEAK,
        "hbm_frac": (16.0 * n + 12.0 * bench["rows_per_set"]) * F / (avg_ms * 1e-3) / 8e12,
        "kernel_ms_rocprof": avg_ms, "kernel_ms_events": bench["kernel_ms"],
        "valu_wave_instr_per_point": valu / (F * n), "fp64_wave_instr_per_point": f64 / (F * n), "salu_per_point": g("SQ_INSTS_SALU") / (F * n),
        "lds_instr_per_point": g("SQ_INSTS_LDS") / (F * n), "lds_bank_conflict_share": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE"),
        "wait_any_share": g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), "active_valu_share_of_busy": g("SQ_ACTIVE_INST_VALU") / g("SQ_BUSY_CYCLES"),
        "traffic_bytes_per_set": (2.0 * g("FETCH_SIZE") + g("WRITE_SIZE")) * 1024.0 / F, "algorithmic_bytes_per_set": 16.0 * n + 12.0 * bench["rows_per_set"]}
bench["roofline"] = roof
json.dump(bench, open(os.path.join(HERE, tag + "_delaunay_bench.json"), "w"), indent=1)
lines = ["# rocprofv3 summary %s — `python profiles/bench_delaunay.py` (%d point sets x %d points per launch)" % (tag, F, n), "",
         "| quantity | value |", "|---|---|",
         "| `delaunay_kernel` average (rocprofv3 --kernel-trace --stats) | %.4f ms (%d calls) |" % (avg_ms, int(krow["Calls"])),
         "| HIP events around the launches, same run | %.4f ms |" % bench["kernel_ms"],
         "| point sets / s, points / s | %.0f, %.3g |" % (F / (avg_ms * 1e-3), F * n / (avg_ms * 1e-3)),
         "| declined sets | %d of %d |" % (bench["declined"], F),
         "| VALU wave-instructions per point (SQ_INSTS_VALU / points) | %.1f (fp64 arithmetic: %.1f; scalar: %.1f; LDS: %.1f) |"
         % (roof["valu_wave_instr_per_point"], roof["fp64_wave_instr_per_point"], roof["salu_per_point"], roof["lds_instr_per_point"]),
         "| **bound: vector-instruction issue** — achieved / peak | **%.1f / %.1f G wave-instr/s = %.3f** |" % (roof["achieved"], roof["peak"], roof["frac"]),
         "| fp64 arithmetic rate (lane-instructions, of 39.3 T/s) | %.3f |" % roof["fp64_lane_instr_frac"],
         "| HBM: algorithmic bytes per set %.0f, measured 2·FETCH+WRITE %.0f B; of the 8 TB/s peak | %.4f |"
         % (roof["algorithmic_bytes_per_set"], roof["traffic_bytes_per_set"], roof["hbm_frac"]),
         "| wave cycles waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES) | %.2f |" % roof["wait_any_share"],
         "| LDS bank-conflict cycles / LDS-active cycles | %.2f |" % roof["lds_bank_conflict_share"], ""]
open(os.path.join(HERE, tag + "_delaunay_summary.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
