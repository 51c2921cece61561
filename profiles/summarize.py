#!/usr/bin/env python3
"""Turn the raw rocprofv3 CSVs of profiles/collect.sh (gpurun_out/<tag>_*) into the committed
summaries: profiles/<tag>_kernel_stats.csv (verbatim --stats table), profiles/<tag>_summary.md
and profiles/traffic.json (HBM bytes per frame of the fused kernel, read by bench.py).

    python profiles/summarize.py r01 [--kernel scale_frames_kernel] [--frames 16384] [--features 2000]

--kernel takes a ";"-separated list of name fragments: the dominant stage of a ragged batch is three launches (one per
size class); their averages and per-dispatch counter medians are added up.  --entry NAME merges the traffic figure into
profiles/traffic.json's "entries" under that key (what bench.py looks up) instead of writing a one-entry file.
"""
import argparse
import csv
import glob
import json
import os
import shutil
import statistics

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("tag")
    ap.add_argument("--kernel", default="scale_frames_kernel")
    ap.add_argument("--frames", type=int, default=16384)
    ap.add_argument("--features", type=int, default=2000)
    ap.add_argument("--traffic-name", default="traffic.json")
    ap.add_argument("--entry", default="")
    a = ap.parse_args()
    kernels = [k for k in a.kernel.split(";") if k]
    out = os.path.join(ROOT, "gpurun_out")
    stats = os.path.join(out, a.tag + "_stats", "bench_kernel_stats.csv")
    shutil.copy(stats, os.path.join(HERE, a.tag + "_kernel_stats.csv"))
    rows = list(csv.DictReader(open(stats)))
    krows = [[r for r in rows if k in r["Name"]][0] for k in kernels]
    bench = json.loads(open(os.path.join(out, a.tag + "_bench_under_rocprof.json")).read().strip().splitlines()[-1])
    counters, road = {}, {}
    for d in sorted(glob.glob(os.path.join(out, a.tag + "_pmc*"))):
        f = os.path.join(d, "bench_counter_collection.csv")
        if not os.path.isfile(f):
            continue
        for r in csv.DictReader(open(f)):
            for k in kernels:
                if k in r["Kernel_Name"]:
                    counters.setdefault(r["Counter_Name"], {}).setdefault(k, []).append(float(r["Counter_Value"]))
            if "road_model_kernel" in r["Kernel_Name"]:
                road.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    med = {c: sum(statistics.median(v) for v in per.values()) for c, per in counters.items()}
    road = {k: statistics.median(v) for k, v in road.items()}
    F = int(bench.get("config", {}).get("frames_per_step_per_gpu", a.frames))
    lines = ["# rocprofv3 summary %s — `python bench.py` (N=1, %d frames x %d features per launch)" % (a.tag, F, a.features), ""]
    lines += ["## kernel trace (`rocprofv3 --kernel-trace --stats`, 10 timed + 2 warm-up steps)", "",
              "| kernel | calls | avg ns | min ns | max ns | % |", "|---|---|---|---|---|---|"]
    for r in rows:
        lines.append("| `%s` | %s | %.0f | %s | %s | %s |" % (r["Name"], r["Calls"], float(r["AverageNs"]), r["MinNs"], r["MaxNs"], r["Percentage"]))
    avg_ms = sum(float(r["AverageNs"]) for r in krows) / 1e6
    lines += ["", "bench.py in the same run (HIP events on the launch stream): kernel_ms_avg = %.4f ms; rocprof average = %.4f ms."
              % (bench["roofline"]["kernel_ms_avg"], avg_ms),
              "roofline: achieved %.1f GB/s algorithmic = %.3f of the 8 TB/s HBM peak; %.2f M frames/s."
              % (bench["roofline"]["achieved"], bench["roofline"]["frac"], bench["value"] / 1e6), ""]
    lines += ["## PMC (separate passes, median over the dispatches of `%s`)" % a.kernel, "", "| counter | per launch | per frame |", "|---|---|---|"]
    for k in sorted(med):
        lines.append("| %s | %.6g | %.6g |" % (k, med[k], med[k] / F))
    traffic = None
    if "FETCH_SIZE" in med and "WRITE_SIZE" in med:
        # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies the 128-B
        # requests of wide coalesced reads at 64 B -> double it.  The request-size breakdown
        # (TCC_EA0_RDREQ_{32B,64B,128B}) is the calibration of that factor for THIS access pattern.
        fetch2 = 2.0 * med["FETCH_SIZE"] * 1024.0
        write = med["WRITE_SIZE"] * 1024.0
        traffic = fetch2 + write
        lines += ["", "HBM traffic per launch = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> B) = %.4g + %.4g = **%.4g B** = %.0f B/frame; "
                  "algorithmic bytes per launch %.4g (%.0f B/frame): traffic/algorithmic = %.3f."
                  % (fetch2, write, traffic, traffic / F, bench["roofline"]["algorithmic_bytes_per_launch"],
                     bench["roofline"]["algorithmic_bytes_per_frame"], traffic / bench["roofline"]["algorithmic_bytes_per_launch"])]
    if "TCC_EA0_RDREQ" in med:
        n32, n64, n128 = med.get("TCC_EA0_RDREQ_32B", 0), med.get("TCC_EA0_RDREQ_64B", 0), med.get("TCC_EA0_RDREQ_128B", 0)
        tot = med["TCC_EA0_RDREQ"]
        exact = 32 * n32 + 64 * n64 + 128 * n128
        lines += ["", "Calibration of the read side from request sizes: RDREQ=%.6g of which 32B=%.6g, 64B=%.6g, 128B=%.6g "
                  "-> 32*n32+64*n64+128*n128 = %.4g B (%.0f B/frame); FETCH_SIZE*1024 = %.4g B, ratio %.3f."
                  % (tot, n32, n64, n128, exact, exact / F, med.get("FETCH_SIZE", 0) * 1024, exact / max(med.get("FETCH_SIZE", 1) * 1024, 1))]
        if exact > 0:
            traffic_exact = exact + med.get("WRITE_SIZE", 0) * 1024
            lines += ["Read bytes by request size + WRITE_SIZE = %.4g B per launch (%.0f B/frame)." % (traffic_exact, traffic_exact / F)]
    if "FETCH_SIZE" in road and "WRITE_SIZE" in road:
        rt = 2.0 * road["FETCH_SIZE"] * 1024.0 + road["WRITE_SIZE"] * 1024.0
        lines += ["", "Second kernel of the step, `road_model_kernel` (reads the dense selected-y lists the scale kernel wrote): "
                  "2 x FETCH_SIZE + WRITE_SIZE = %.4g B per launch (%.0f B/frame)." % (rt, rt / F)]
    if "SQ_WAVE_CYCLES" in med:
        wc = med["SQ_WAVE_CYCLES"]
        lines += ["", "Wave-cycle shares: ACTIVE_INST_ANY %.2f, WAIT_ANY %.2f, WAIT_INST_ANY %.2f; VALU %.2f, LDS %.2f of wave cycles."
                  % (med.get("SQ_ACTIVE_INST_ANY", 0) / wc, med.get("SQ_WAIT_ANY", 0) / wc, med.get("SQ_WAIT_INST_ANY", 0) / wc,
                     med.get("SQ_ACTIVE_INST_VALU", 0) / wc, med.get("SQ_ACTIVE_INST_LDS", 0) / wc)]
    open(os.path.join(HERE, a.tag + "_summary.md"), "w").write("\n".join(lines) + "\n")
    if traffic is not None and a.entry:
        path = os.path.join(HERE, a.traffic_name)
        doc = json.load(open(path)) if os.path.isfile(path) else {}
        doc.setdefault("entries", {})[a.entry] = {
            "tag": a.tag, "frames": F, "features": a.features, "hbm_bytes_per_launch": traffic, "hbm_bytes_per_frame": traffic / F,
            "fetch_size_kib": med["FETCH_SIZE"], "write_size_kib": med["WRITE_SIZE"], "kernels": kernels,
            "method": "2*FETCH_SIZE+WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section), separate --pmc passes"}
        json.dump(doc, open(path, "w"), indent=1)
    elif traffic is not None:
        json.dump({"tag": a.tag, "frames": F, "features": a.features, "hbm_bytes_per_launch": traffic,
                   "hbm_bytes_per_frame": traffic / F, "fetch_size_kib": med["FETCH_SIZE"], "write_size_kib": med["WRITE_SIZE"],
                   "method": "2*FETCH_SIZE+WRITE_SIZE (gfx950 correction, MI355X_MICROARCH.md HBM section), separate --pmc passes"},
                  open(os.path.join(HERE, a.traffic_name), "w"), indent=1)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
