#!/usr/bin/env python3
"""Render the measured-numbers tables of BASELINE.md, README.md and DESIGN.md from the committed measurement files of a
round (profiles/r0N_*.json / .jsonl / .csv / .txt), so that no number in those documents is typed by hand:

    python profiles/make_tables.py [r04]        # rewrites the text between <!-- TAG-tables:begin --> and <!-- TAG-tables:end -->

Sources: r0N_bench.json / r0N_bench_kitti.json (the bench lines: `python bench.py [--workload kitti]`), r0N_bench_c4_1gpu.json
(BASELINE configs[3] literally, on one GPU), r0N_summary.md's source r0N_kernel_stats.csv (rocprofv3 --kernel-trace --stats),
r0N_delaunay_bench.jsonl (profiles/bench_delaunay.py), r0N_e2e_{scale,rescale}_busy.txt (profiles/e2e_gpu_busy.py),
r0N_fixed_mode_accuracy.json (profiles/fixed_mode_accuracy.py), traffic.json (PMC passes).
"""
import csv
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def load_line(name):
    p = os.path.join(HERE, name)
    if not os.path.isfile(p):
        return None
    lines = [ln for ln in open(p).read().splitlines() if ln.startswith("{")]
    return json.loads(lines[-1]) if lines else None        # (an empty file: a run that produced no line)


def k(x):
    if x < 1e4:
        return "%.2f k" % (x / 1e3)
    return "%.0f k" % (x / 1e3) if x < 1e6 else "%.2f M" % (x / 1e6)


def render(tag):
    out = []
    b = load_line(tag + "_bench.json")
    kt = load_line(tag + "_bench_kitti.json")
    c4 = load_line(tag + "_bench_c4_1gpu.json")
    rows = []
    if b:
        r = b["roofline"]
        rows.append(("headline step (`python bench.py`): %s" % b["config"]["workload"].split(",")[0],
                     "**%s frames/s**, %.3f ms per step; `%s` %.3f ms by HIP events = %.0f GB/s algorithmic = **%.3f of 8 TB/s**; both kernels of the step %.3f"
                     % (k(b["value"]), b["ms_per_step"], r["kernel"], r["kernel_ms_avg"], r["achieved"], r["frac"], r.get("step_frac", float("nan"))),
                     "`profiles/%s_bench.json`" % tag))
    stats = os.path.join(HERE, tag + "_kernel_stats.csv")
    if os.path.isfile(stats) and b:
        for r_ in csv.DictReader(open(stats)):
            if "scale_frames_kernel<8, 4, 0, false>" in r_["Name"]:
                ms = float(r_["AverageNs"]) / 1e6
                bytes_launch = b["roofline"]["algorithmic_bytes_per_launch"]
                tj = json.load(open(os.path.join(HERE, "traffic.json")))["entries"].get("c2_2000", {})
                rows.append(("the same kernel under `rocprofv3 --kernel-trace --stats`", "%.3f ms average over %s launches = %.0f GB/s = **%.3f of 8 TB/s**; HBM traffic (2 x FETCH_SIZE + WRITE_SIZE, separate --pmc passes) %.3f x the algorithmic bytes"
                             % (ms, r_["Calls"], bytes_launch / ms / 1e6, bytes_launch / ms / 1e6 / 8000.0, tj.get("hbm_bytes_per_frame", float("nan")) / b["roofline"]["algorithmic_bytes_per_frame"]),
                             "`profiles/%s_kernel_stats.csv`, `%s_summary.md`, `traffic.json`" % (tag, tag)))
    if b:
        for key, label in (("e2e_rescale", "end to end from per-frame arrays, `rescale.ScaleEstimator(triangulation=\"gpu\")` — the estimator `/root/reference/src/main.py:20` imports; no declared deviation"),
                           ("e2e_gpu_triangulation", "end to end, `scale_calculator.ScaleEstimator(triangulation=\"gpu\")` (`check_triangle=\"fixed\"`: a declared deviation)"),
                           ("e2e", "end to end, host SciPy triangulations (`triangulation=\"scipy\"`: reference-exact; the default of rounds 1-4), worker pool on the box's CPUs")):
            if key in b and "value" in b[key]:
                e = b[key]
                extra = ""
                if kt and key in kt and "value" in kt[key]:
                    extra = "; KITTI-sized frames (300-1500 features): %s" % k(kt[key]["value"])
                if "hip_malloc_calls_in_timed_call" in e:
                    extra += "; hipMalloc / hipHostMalloc calls in the timed call: %d / %d" % (e["hip_malloc_calls_in_timed_call"], e["hip_host_malloc_calls_in_timed_call"])
                rows.append((label, "**%s frames/s** at 2000 features (%d frames, %s distinct)%s" % (k(e["value"]), e["frames"], e.get("distinct_frames", "all"), extra), "`%s` in the bench line" % key))
        if "e2e_gpu_exact" in b and "value" in b["e2e_gpu_exact"]:
            e = b["e2e_gpu_exact"]
            dk = e.get("delaunay_kernel") or {}
            extra = "; KITTI-sized frames (300-1500 features): %s" % k(kt["e2e_gpu_exact"]["value"]) if (kt and "value" in kt.get("e2e_gpu_exact", {})) else ""
            declined = ("declined to the host in the whole call: %d" % e["declined_total"]) if "declined_total" in e else ("declined to the host in the last chunk: %d" % e.get("declined_last_chunk", 0))
            sc = e.get("qhull_selfcheck") or {}
            guard = ("; first-use self-check of the replay against the installed SciPy %s: %s (device %d / host replay %d sets compared, 0 different)"
                     % (sc.get("scipy"), "passed" if sc.get("ok") else "FAILED", (sc.get("device") or {}).get("compared", 0), (sc.get("host") or {}).get("compared", 0))) if sc else ""
            rf = e.get("roofline") or {}
            roof = ("; its roof is vector instruction issue: %.0f VALU + %.0f scalar instructions per insertion, **%.2f of the VALU issue peak** (%.2e of %.2e wave-instructions/s), HBM at %.0f %% of 8 TB/s"
                    % (rf["valu_per_insertion"], rf["salu_per_insertion"], rf["frac"], rf["achieved"], rf["peak"], 100 * rf["hbm_frac_of_8TBps"])) if "frac" in rf else ""
            rows.append(("end to end, `scale_calculator.ScaleEstimator(...)` as the reference constructs it (= `triangulation=\"gpu\", check_triangle=\"reference\"`, the default since round 5) — **the reference's result, bit for bit, both triangulations on the device** (Qhull's rows by `qhull_rows_kernel`)",
                         "**%s frames/s** at 2000 features (%d frames, %s distinct; %s)%s; `qhull_rows_kernel` alone: %s sets/s (%d resident sets of %d points)%s%s"
                         % (k(e["value"]), e["frames"], e.get("distinct_frames", "all"), declined, extra,
                            k(dk.get("sets_per_s", float("nan"))), dk.get("sets", 0), dk.get("points_per_set", 0), roof, guard), "`e2e_gpu_exact` in the bench line"))
        if "two_streams" in b and "ms_per_step" in b["two_streams"]:
            t2 = b["two_streams"]
            rows.append(("the headline steps alternating between two streams (road model of step k under the scale kernel of step k+1)",
                         "%.3f ms per step = %.3f of 8 TB/s (one stream: %.3f ms = %.3f); raw scales identical: %s" % (t2["ms_per_step"], t2["step_frac"], b["ms_per_step"], b["roofline"].get("step_frac", float("nan")), t2.get("raw_scales_equal")), "`two_streams` in the bench line"))
        if "latency" in b:
            la = b["latency"]
            extra = (", `rescale` estimator device-resident %.2f ms" % la["rescale_gpu"]["median_ms"]) if "rescale_gpu" in la else ""
            if "gpu_exact" in la:
                if la["gpu_exact"].get("host_replay"):
                    extra += (", `check_triangle=\"reference\"` with `triangulation=\"gpu\"` **%.2f ms** (the DEFAULT construction, the reference's result: the first triangulation by the host replay of Qhull's run `mvosr_qhull_rows_host` — SciPy's own rows, "
                              "held to the installed SciPy by the first-use self-check; %d of the calls' sets declined to SciPy —, the second by the fast kernel as a stand-in; round 5, with SciPy for the first triangulation: 3.3 ms)"
                              % (la["gpu_exact"]["median_ms"], la["gpu_exact"].get("host_replay_declined_to_scipy", 0)))
                else:
                    extra += ", `check_triangle=\"reference\"` with `triangulation=\"gpu\"` %.2f ms (the DEFAULT construction, the reference's result: SciPy for the first triangulation only, the second by the fast kernel as a stand-in — the device's replay of Qhull is 20 ms per triangulation for a single frame)" % la["gpu_exact"]["median_ms"]
            rows.append(("per-frame `scale_calculation` latency, 2000 features", "SciPy triangulations %.2f ms, device triangulations %.2f ms%s (median; 0 allocations per call)" % (la["scipy"]["median_ms"], la["gpu"]["median_ms"], extra), "`latency` in the bench line"))
        cb = b.get("cpu_baseline")
        if cb:
            rows.append(("CPU oracle on the GPU box's host cores (baseline, not target)", "%.0f frames/s on %d core (vectorised NumPy port); all cores: %s; reference-shaped Python loops: %s"
                         % (cb["value"], cb["cores"], ("%.0f on %d" % (b["cpu_baseline_all_cores"]["value"], b["cpu_baseline_all_cores"]["cores"])) if "cpu_baseline_all_cores" in b else "n/a",
                            ("%.1f frames/s/core" % b["cpu_baseline_reference_shaped"]["value"]) if "cpu_baseline_reference_shaped" in b else "n/a"), "bench line"))
    if kt:
        r = kt["roofline"]
        rows.append(("config C3's sizes (`--workload kitti`: ragged 300-1500 features, one launch per size class)", "%s frames/s, kernels %.3f of 8 TB/s, step %.3f" % (k(kt["value"]), r["frac"], r.get("step_frac", float("nan"))), "`profiles/%s_bench_kitti.json`" % tag))
    dn = load_line(tag + "_bench_dense.json")
    if dn:
        r = dn["roofline"]
        ex = dn.get("e2e_gpu_exact", {})
        rows.append(("config C5's shape (`--features 20000`: dense frames, tiled kernel)",
                     "%s frames/s, `%s` %.3f of 8 TB/s, step %.3f; end to end: SciPy on the host %s, device triangulations (fixed mode) %s%s frames/s"
                     % (k(dn["value"]), r["kernel"], r["frac"], r.get("step_frac", float("nan")), k(dn.get("e2e", {}).get("value", float("nan"))),
                        k(dn.get("e2e_gpu_triangulation", {}).get("value", float("nan"))),
                        (", the reference's result with device triangulations (`qhull_rows_kernel<uint32_t>`) **%s**" % k(ex["value"])) if "value" in ex else ""),
                     "`profiles/%s_bench_dense.json`" % tag))
    gr = load_line(tag + "_bench_gridded.json")
    if gr:
        ex, fx, ho = gr.get("e2e_gpu_exact", {}), gr.get("e2e_gpu_triangulation", {}), gr.get("e2e", {})
        if "value" in ex:
            rows.append(("the decline path (`--workload gridded`: 2000-feature frames with pixel coordinates rounded to 1/4 px — collinear and cocircular sites, where Qhull merges facets and both device triangulations DECLINE to the host's SciPy)",
                         "the default estimator end to end **%s frames/s** with %.1f %% of the frames declined (%d of %d; triangulated by SciPy on the worker pool under the GPU's queued chunks, re-run on a context of their own), 0 rows different from SciPy by construction; "
                         "the fixed-mode estimator %s (declined %.1f %%); host SciPy for every frame: %s"
                         % (k(ex["value"]), 100 * ex.get("declined_fraction", 0), ex.get("declined_total", 0), ex["frames"], k(fx.get("value", float("nan"))), 100 * fx.get("declined_fraction", 0), k(ho.get("value", float("nan")))),
                         "`profiles/%s_bench_gridded.json`" % tag))
    g16, g16b = load_line(tag + "_bench_grid16.json"), load_line(tag + "_bench_grid16_before.json")
    if g16 and "value" in g16.get("e2e_gpu_exact", {}):
        ex, fx = g16["e2e_gpu_exact"], g16.get("e2e_gpu_triangulation", {})
        before = ""
        if g16b and "value" in g16b.get("e2e_gpu_exact", {}):
            bx, bf = g16b["e2e_gpu_exact"], g16b.get("e2e_gpu_triangulation", {})
            before = " (the same box with the kernel as it was: %s with %.0f %% declined / %s with %.0f %%)" % (
                k(bx["value"]), 100 * bx.get("declined_fraction", 0), k(bf.get("value", float("nan"))), 100 * bf.get("declined_fraction", 0))
        rows.append(("... coordinates on a 1/16 px grid (`--workload gridded --snap-grid 0.0625`: what a detector refining to 1/16 px hands over — exactly collinear triples in two frames of three, few cocircular quadruples)",
                     "the default estimator end to end **%s frames/s** with %.1f %% of the frames declined (%d of %d), the fixed-mode estimator **%s** (%.1f %% declined)%s: "
                     "`delaunay_kernel` no longer declines a collinear site beyond q of an interior edge, nor an EXACTLY collinear one on the hull (round 6)"
                     % (k(ex["value"]), 100 * ex.get("declined_fraction", 0), ex.get("declined_total", 0), ex["frames"], k(fx.get("value", float("nan"))),
                        100 * fx.get("declined_fraction", 0), before), "`profiles/%s_bench_grid16.json`" % tag))
    g5 = load_line(tag + "_bench_gridded_0005.json")
    if g5 and "value" in g5.get("e2e_gpu_exact", {}):
        ex = g5["e2e_gpu_exact"]
        rows.append(("... a FEW declined frames per chunk (`--workload gridded --snap-fraction 0.005`; round 5: 72 declined frames in 16 384 took the call from 53 k to 30 k frames/s, re-run in the chunk's epilogue on the chunk's own stream)",
                     "the default estimator end to end **%s frames/s** with %d of %d frames declined (%.2f %%), the fixed-mode estimator **%s** (without declines: the rows above): the declined frames' re-runs on a context of their own, a chunk's few frames STARTED "
                     "when the chunk is collected and advanced while later chunks run (first triangulations on the worker pool, vote, second triangulations, product kernels), more than 16 per chunk in one merged re-run after the call's last chunk"
                     % (k(ex["value"]), ex.get("declined_total", 0), ex["frames"], 100 * ex.get("declined_fraction", 0), k(g5.get("e2e_gpu_triangulation", {}).get("value", float("nan")))), "`profiles/%s_bench_gridded_0005.json`" % tag))
    for nm, what in (("share2", "`bench.py --gpus 2 --share-gpu` (two ranks sharing this one GPU over gloo: the N-rank code path as a dry run, not a scaling number)"),
                     ("share2_c4", "`bench.py --gpus 2 --share-gpu --c4 --total-frames 100000` (configs[3]'s split, two ranks on one GPU, dry run)")):
        sh = load_line(tag + "_bench_%s.json" % nm)
        if sh:
            rows.append((what, "ran: %s frames/s whole-job, %d ranks, scaling \"%s\", one collective per step" % (k(sh["value"]), sh["n_gpus"], sh["scaling"]), "`profiles/%s_bench_%s.json`" % (tag, nm)))
    if c4:
        rows.append(("BASELINE configs[3] literally on ONE GPU (`bench.py --c4 --total-frames 1000000 --gpus 1`: %.0f GB resident)" % (c4["roofline"]["algorithmic_bytes_per_launch"] / 1e9),
                     "%s frames/s, %.1f ms per step of 1 000 000 frames, kernel %.3f of 8 TB/s" % (k(c4["value"]), c4["ms_per_step"], c4["roofline"]["frac"]), "`profiles/%s_bench_c4_1gpu.json`" % tag))
    dj = os.path.join(HERE, tag + "_delaunay_bench.jsonl")
    if os.path.isfile(dj):
        items = [json.loads(ln) for ln in open(dj) if ln.startswith("{")]
        txt = "; ".join("%s, %d points: **%s sets/s**" % (d["what"], d["points_per_set"], k(d["sets_per_s"])) for d in items if d["points_per_set"] == 2000)
        rows.append(("`delaunay_kernel` alone, 4096 resident sets", txt, "`profiles/%s_delaunay_bench.jsonl`" % tag))
        small = "; ".join("%s, <= %d points: %s" % (d["what"], d["points_per_set"], k(d["sets_per_s"])) for d in items if d["points_per_set"] != 2000)
        if small:
            rows.append(("... smaller sets (8192 per launch)", small, "same"))
    qc = os.path.join(HERE, tag + "_qhull_check.txt")
    if os.path.isfile(qc):
        launches = [ln.strip() for ln in open(qc) if ln.startswith("launch of")]
        rows.append(("`qhull_rows_kernel` alone (resident point sets, one launch)", "; ".join(launches), "`profiles/%s_qhull_check.txt`" % tag))
    for w in ("scale", "rescale"):
        p = os.path.join(HERE, "%s_e2e_%s_busy.txt" % (tag, w))
        if os.path.isfile(p):
            t = open(p).read()
            m = re.search(r"span ([\d.]+) ms, busy ([\d.]+) ms \(([\d.]+) %\)", t)
            d = re.search(r"delaunay_kernel.*?\s([\d.]+) ms\s+(\d+) launches", t)
            if m and d:
                rows.append(("GPU timeline of one 32 768-frame end-to-end call (%s estimator)" % w,
                             "kernel span %s ms, busy %s ms (%s %%); `delaunay_kernel` %s ms = %.2f us per frame for both triangulations" % (m.group(1), m.group(2), m.group(3), d.group(1), float(d.group(1)) * 1e3 / 32768),
                             "`profiles/%s_e2e_%s_busy.txt`, `%s_e2e_%s_kernel_stats.csv`" % (tag, w, tag, w)))
    fa = os.path.join(HERE, tag + "_fixed_mode_accuracy.json")
    if os.path.isfile(fa):
        for r in json.load(open(fa))["rows"]:
            a, d = r["all_frames"], r["differing_frames"]
            rows.append(("declared deviation `check_triangle=\"fixed\"` vs the reference, raw scales, %s" % r["set"],
                         "%.1f %% bit-equal; mean relative error against the generator's true scale: reference %.4f / fixed %.4f over all frames, %.3f / %.3f over the %d frames that differ (fixed closer on %d, reference on %d)"
                         % (100 * r["bit_equal_fraction"], a["reference_mean_rel_err"], a["fixed_mean_rel_err"], d["reference_mean_rel_err"], d["fixed_mean_rel_err"], d["count"], d["fixed_closer_to_truth"], d["reference_closer_to_truth"]),
                         "`profiles/%s_fixed_mode_accuracy.json`" % tag))
    out.append("| quantity (round %s, one MI355X) | value | source |" % tag[1:].lstrip("0"))
    out.append("|---|---|---|")
    for a_, b_, c_ in rows:
        out.append("| %s | %s | %s |" % (a_, b_, c_))
    return "\n".join(out)


def test_counts():
    """`pytest --collect-only` of the suite as it stands (the counts in the documents are rendered, not typed), and the GPU box's pass
    count of the round's collection (profiles/r06_gputest_count.txt) next to it."""
    import subprocess
    out = {}
    for name, expr in (("gpu", "gpu"), ("cpu", "not gpu")):
        r = subprocess.run([sys.executable, "-m", "pytest", "tests", "--collect-only", "-q", "-m", expr], cwd=ROOT, capture_output=True, text=True)
        m = re.search(r"(\d+)/(\d+) tests collected", r.stdout) or re.search(r"(\d+) tests? collected", r.stdout)
        out[name] = int(m.group(1)) if m else -1
    return out


def render_counts(tag):
    c = test_counts()
    txt = "`-m gpu`: %d tests, `-m \"not gpu\"`: %d (`pytest --collect-only`, rendered by `profiles/make_tables.py`)" % (c["gpu"], c["cpu"])
    p = os.path.join(HERE, tag + "_gputest_count.txt")
    if os.path.isfile(p):
        txt += "; on the GPU box, the suite as shipped (no `MVOSR_TRIANGULATION` anywhere): %s (`profiles/%s_gputest_count.txt`)" % (open(p).read().strip().strip("=").strip(), tag)
    return txt


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
    text = render(tag)
    open(os.path.join(HERE, tag + "_tables.md"), "w").write(text + "\n")
    begin, end = "<!-- %s-tables:begin -->" % tag, "<!-- %s-tables:end -->" % tag
    counts = render_counts(tag) if tag >= "r06" else None
    for doc in ("BASELINE.md", "README.md", "DESIGN.md"):
        p = os.path.join(ROOT, doc)
        s = open(p).read()
        if begin in s and end in s:
            s = s[:s.index(begin) + len(begin)] + "\n" + text + "\n" + s[s.index(end):]
            open(p, "w").write(s)
            print("updated", doc)
        cb, ce = "<!-- test-counts:begin -->", "<!-- test-counts:end -->"
        if counts and cb in s and ce in s:
            s = s[:s.index(cb) + len(cb)] + counts + s[s.index(ce):]
            open(p, "w").write(s)
    print(text)


if __name__ == "__main__":
    main()
