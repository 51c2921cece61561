#!/usr/bin/env python3
"""Is the declared deviation check_triangle="fixed" benign?  Raw-scale error against the synthetic generator's TRUE scale
(1.75 / h_cam of each frame) for the reference as written and for the reference with the one line of check_triangle patched
(/root/reference/src/scale_calculator.py:113-115) — both taken from the committed goldens the reference itself produced
(tests/golden/seq4541.npz / seq4541_fixed.npz: config C3's ragged 300-1500-feature frames; seq200.npz / seq200_fixed.npz:
1800-2200-feature frames), so nothing is recomputed here and no oracle is involved.

    python profiles/fixed_mode_accuracy.py            # prints the table, writes profiles/r04_fixed_mode_accuracy.json
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
ABS_REF = 1.75


def h_cam(frame_idx):
    """mvoscalerecovery_amd.synth.synth_frame's camera height of frame `frame_idx` (VO units)."""
    return 0.6 + 0.4 * ((frame_idx * 0.6180339887498949) % 1.0)


def load(name):
    z = np.load(os.path.join(GOLD, name), allow_pickle=False)
    kinds = z["kinds"]
    frames = np.nonzero(kinds == 1)[0]                       # frames that reached scale_calculation, in order
    raws = z["raw_scales"]
    assert len(frames) == len(raws), (name, len(frames), len(raws))
    return frames, raws


def table(ref_name, fix_name, label):
    fr, ref = load(ref_name)
    fr2, fix = load(fix_name)
    assert np.array_equal(fr, fr2)
    truth = ABS_REF / h_cam(fr.astype(np.float64))
    ok = np.isfinite(ref) & np.isfinite(fix)
    e_ref, e_fix = np.abs(ref - truth) / truth, np.abs(fix - truth) / truth
    differ = ok & (ref != fix)
    row = {"set": label, "frames": int(ok.sum()), "bit_equal_fraction": float(np.mean(ref[ok] == fix[ok])),
           "all_frames": {"reference_mean_rel_err": float(e_ref[ok].mean()), "fixed_mean_rel_err": float(e_fix[ok].mean()),
                          "reference_median_rel_err": float(np.median(e_ref[ok])), "fixed_median_rel_err": float(np.median(e_fix[ok]))},
           "differing_frames": {"count": int(differ.sum()),
                                "reference_mean_rel_err": float(e_ref[differ].mean()) if differ.any() else None,
                                "fixed_mean_rel_err": float(e_fix[differ].mean()) if differ.any() else None,
                                "fixed_closer_to_truth": int((e_fix[differ] < e_ref[differ]).sum()),
                                "reference_closer_to_truth": int((e_ref[differ] < e_fix[differ]).sum()),
                                "median_rel_difference_between_modes": float(np.median(np.abs(fix[differ] - ref[differ]) / ref[differ])) if differ.any() else None}}
    return row


def main():
    rows = [table("seq4541.npz", "seq4541_fixed.npz", "C3 sequence, 300-1500 features (4541 frames)"),
            table("seq200.npz", "seq200_fixed.npz", "1800-2200 features (200 frames)")]
    out = {"what": "raw scale vs the generator's true scale 1.75/h_cam; 'reference' = /root/reference as written, 'fixed' = the same "
                   "with check_triangle's (v0,v2) pair marking vertices 0 and 2 on canonical rows (tests/golden/make_golden.py: "
                   "fixed_reference); values are the reference's own outputs from the committed goldens",
           "note": "the synthetic scenes have no camera pitch while the path assumes -0.5 deg (scale_calculator.py:24), and the height is "
                   "quantised to histogram bins of 0.05-0.1: both modes carry the same systematic error; what the table answers is "
                   "whether the modes differ in accuracy where they differ at all",
           "rows": rows}
    with open(os.path.join(ROOT, "profiles", "r04_fixed_mode_accuracy.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print("| set | frames | bit-equal | mean rel. err reference / fixed (all) | differing frames | mean rel. err reference / fixed (differing) | fixed closer / reference closer |")
    print("|---|---|---|---|---|---|---|")
    for r in rows:
        a, d = r["all_frames"], r["differing_frames"]
        print("| %s | %d | %.1f %% | %.4f / %.4f | %d | %.3f / %.3f | %d / %d |" % (
            r["set"], r["frames"], 100 * r["bit_equal_fraction"], a["reference_mean_rel_err"], a["fixed_mean_rel_err"], d["count"],
            d["reference_mean_rel_err"] or 0, d["fixed_mean_rel_err"] or 0, d["fixed_closer_to_truth"], d["reference_closer_to_truth"]))


if __name__ == "__main__":
    sys.exit(main())
