"""Row f3: evaluation mirrors vs the reference's own scripts (goldens: tests/golden/eval.npz)."""
import re

import numpy as np

from conftest import load_npz


def test_eval_golden(tmp_path):
    from mvoscalerecovery_amd import evaluate, offline
    z, _ = load_npz("eval.npz")
    motions, est = z["motions"], z["est_scale"]
    gt = offline.get_path(motions, np.ones(len(est)))
    res = offline.save_outputs(str(tmp_path) + "/seq_", ".t", est, motions)
    assert np.array_equal(np.loadtxt(str(tmp_path) + "/seq_scales.txt.t"), est)
    np.testing.assert_allclose(np.loadtxt(str(tmp_path) + "/seq_path.txt.t"), res, rtol=0, atol=0)
    errors = evaluate.calculate_sequence_error(gt, res)
    np.testing.assert_allclose(np.array(errors), z["errors"], rtol=1e-9, atol=1e-15)
    rot, tra, _ = evaluate.calculate_ave_errors(errors)
    np.testing.assert_allclose(rot, z["rot"], rtol=1e-9)
    np.testing.assert_allclose(tra, z["tra"], rtol=1e-9)
    np.testing.assert_allclose(evaluate.pose2motion(gt[:50]), z["pose2motion"], atol=1e-12)
    np.testing.assert_allclose(offline.motion2pose(motions[:50]), z["motion2pose"], atol=1e-12)
    head, drift = evaluate.evaluate_scale(np.ones(len(est)), est)
    printed = [float(x) for x in re.findall(r"[-+]?\d+\.?\d*(?:[eE][-+]?\d+)?", str(z["scale_print"]).replace("np.float64", ""))]
    np.testing.assert_allclose(list(head) + list(drift), printed, rtol=1e-12)
