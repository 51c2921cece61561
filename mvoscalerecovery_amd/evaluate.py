"""Downstream scoring of a scale-recovered trajectory: the KITTI-devkit style
segment error of /root/reference/script/evaluate_vo.py:5-96, the scale error statistics of
/root/reference/script/evaluate_scale.py:4-29 and pose <-> motion conversion of
/root/reference/script/transformation.py:6-32.  Host NumPy, like the reference: this is evaluation of
the path's output (a few thousand 3x4 poses), not part of the per-frame hot path."""
from __future__ import annotations

import numpy as np

LENGTHS = [100, 200, 300, 400, 500, 600, 700, 800]      # evaluate_vo.py:46
STEP = 10                                               # evaluate_vo.py:45


def _mat(line):
    m = np.eye(4)
    m[:3, :] = np.asarray(line, dtype=np.float64).reshape(3, 4)
    return m


def trajectory_distances(poses):
    """evaluate_vo.py:5-13: cumulative path length over the translation columns."""
    t = np.asarray(poses)[:, 3:12:4]
    d = [0]
    for i in range(1, t.shape[0]):
        d.append(d[i - 1] + np.linalg.norm(t[i - 1] - t[i]))
    return d


def calculate_sequence_error(poses_gt, poses_result):
    """evaluate_vo.py:40-79: rows [first_frame, r_err/len, t_err/len, len, speed]."""
    poses_gt, poses_result = np.asarray(poses_gt), np.asarray(poses_result)
    dist = trajectory_distances(poses_gt)
    errors = []
    for first in range(0, poses_gt.shape[0], STEP):
        for length in LENGTHS:
            last = -1
            for i in range(first, len(dist)):                                   # :15-19
                if dist[i] > dist[first] + length:
                    last = i
                    break
            if last == -1:
                continue
            d_gt = np.linalg.inv(_mat(poses_gt[first])) @ _mat(poses_gt[last])            # :65
            d_re = np.linalg.inv(_mat(poses_result[first])) @ _mat(poses_result[last])    # :66
            err = np.linalg.inv(d_re) @ d_gt                                              # :67
            dd = 0.5 * (err[0, 0] + err[1, 1] + err[2, 2] - 1)
            r_err = np.arccos(max(min(dd, 1.0), -1.0))                                    # :21-27
            t_err = np.sqrt(err[0, 3] * err[0, 3] + err[1, 3] * err[1, 3] + err[2, 3] * err[2, 3])   # :29-33
            speed = length / (0.1 * float(last - first + 1))                              # :72-73
            errors.append([first, r_err / length, t_err / length, length, speed])
    return errors


def calculate_ave_errors(errors):
    """evaluate_vo.py:80-96: per-length mean rotation (deg/m) and translation (fraction) errors."""
    rot, tra, tra_all = [], [], []
    for length in LENGTHS:
        r = [e[1] for e in errors if abs(e[3] - length) < 1]
        t = [e[2] for e in errors if abs(e[3] - length) < 1]
        tra_all.append(t)
        if r:
            rot.append(sum(r) / len(r))
            tra.append(sum(t) / len(t))
    return np.array(rot) * 180 / np.pi, tra, tra_all


def patch(data, window=10, step=2):
    """evaluate_scale.py:19-23."""
    data = np.asarray(data)
    return np.abs(np.array([np.sum(data[i:i + window]) for i in range(0, data.shape[0] - window, step)])) / window


def evaluate_scale(gt, re):
    """evaluate_scale.py:4-13, returned instead of printed: (mean |err|, max |err|, share within
    0.1/0.2/0.3/0.5, windowed drift for windows 10..800)."""
    gt, re = np.asarray(gt, dtype=np.float64), np.asarray(re, dtype=np.float64)
    n = re.shape[0]
    er = np.abs(gt[:n] - re)
    head = (np.mean(er), np.max(er), 1 - np.sum(er > 0.1) / n, 1 - np.sum(er > 0.2) / n, 1 - np.sum(er > 0.3) / n,
            1 - np.sum(er > 0.5) / n)
    ers = gt[:n] - re
    drift = [np.mean(patch(ers, w, 10)) for w in [10, 20, 50, 100, 200, 300, 400, 500, 600, 700, 800]]
    return head, drift


def pose2motion(poses):
    """transformation.py:23-32."""
    poses = np.asarray(poses)
    out = np.zeros((poses.shape[0] - 1, 12))
    for i in range(poses.shape[0] - 1):
        out[i] = (np.linalg.inv(_mat(poses[i])) @ _mat(poses[i + 1]))[:3].reshape(-1)
    return out
