#!/usr/bin/env python3
"""Inputs of tools/dump_reference_vectors.cc: the seeded frames and PoseOptimization problems this repository tests with, as .npy files.
usage: python tests/golden/make_ref_inputs.py <out-dir>"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from ba_scene import pose_problem
from rumi_slam_amd.synth import synth_frame

FRAMES = [dict(seed=1234), dict(seed=1235), dict(seed=3, n_rect=60, contrast=(8, 19)), dict(seed=77, w=752, h=480)]
POSES = [100, 101, 102]


def main(out):
    os.makedirs(out, exist_ok=True)
    for i, kw in enumerate(FRAMES):
        np.save(os.path.join(out, f"frame_{i}.npy"), synth_frame(**kw))
    for i, seed in enumerate(POSES):
        p = pose_problem(seed, 300, 0.1)
        rows = np.zeros((len(p["inv_sigma2"]) + 1, 11), np.float64)
        rows[0, :4] = p["K"]; rows[0, 4:11] = p["T0"]
        rows[1:, 0:3] = p["Xw"]; rows[1:, 3:5] = p["obs"]; rows[1:, 5] = p["inv_sigma2"]
        np.save(os.path.join(out, f"pose_in_{i}.npy"), rows)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "ref_in")
