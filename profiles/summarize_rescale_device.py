#!/usr/bin/env python3
"""Per-frame counter medians of the device-resident rescale path's kernels from profiles/collect_rescale_device.sh:
    python profiles/summarize_rescale_device.py r04  ->  profiles/r04_rescale_device_summary.md"""
import csv, glob, os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__)); ROOT = os.path.dirname(HERE)
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
per = {}
for d in sorted(glob.glob(os.path.join(ROOT, "gpurun_out", tag + "_rsd_pmc*"))):
    f = os.path.join(d, "e2e_counter_collection.csv")
    if not os.path.isfile(f):
        continue
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void mvosr::", "")
        per.setdefault(k, {}).setdefault(r["Counter_Name"], []).append((float(r["Counter_Value"]), int(r.get("Grid_Size", 0) or 0), int(r.get("Workgroup_Size", 1) or 1)))
lines = ["# rocprofv3 counters, device-resident rescale path (%s): per FRAME (= per workgroup), median over the dispatches of full chunks" % tag, ""]
for k in sorted(per):
    if not any(x in k for x in ("flat_selection", "graph_inliers", "delaunay")):
        continue
    lines += ["## `%s`" % k, "", "| counter | per frame |", "|---|---|"]
    vals = {}
    for c, items in sorted(per[k].items()):
        pf = [v / max(g // max(w, 1), 1) for v, g, w in items if g // max(w, 1) >= 1024]
        if pf:
            vals[c] = statistics.median(pf)
            lines.append("| %s | %.4g |" % (c, vals[c]))
    if "SQ_ACTIVE_INST_VALU" in vals and "SQ_BUSY_CYCLES" in vals:
        lines += ["", "VALU-active share of wave cycles: %.2f; waiting share: %.2f; fp64 arithmetic instructions: %.0f of %.0f VALU wave-instructions per frame."
                  % (vals["SQ_ACTIVE_INST_VALU"] / vals["SQ_WAVE_CYCLES"], vals["SQ_WAIT_ANY"] / vals["SQ_WAVE_CYCLES"],
                     sum(vals.get(x, 0) for x in ("SQ_INSTS_VALU_FMA_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_TRANS_F64")), vals.get("SQ_INSTS_VALU", 0))]
    if "FETCH_SIZE" in vals:
        lines += ["HBM traffic per frame: 2 x FETCH_SIZE + WRITE_SIZE = %.0f B." % (2048 * vals["FETCH_SIZE"] + 1024 * vals.get("WRITE_SIZE", 0))]
    lines.append("")
open(os.path.join(HERE, tag + "_rescale_device_summary.md"), "w").write("\n".join(lines))
print("\n".join(lines))
