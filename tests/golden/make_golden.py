#!/usr/bin/env python3
"""Generates the committed golden vectors of the ORB extractor from the CPU oracle.

The reference holds NO golden vectors / tests for this path and cannot be built here (no OpenCV / Eigen
in the image), so these vectors are produced by oracle/ (the CPU restatement) — they pin the oracle
against regressions and give the GPU tests a fixture that does not need the oracle at run time.
Parity against the reference BINARY is unpinned (DESIGN.md "Oracle").

    python tests/golden/make_golden.py        # rewrites tests/golden/orb_*.npz
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]

import oracle_lib  # noqa: E402
from rumi_slam_amd.synth import synth_frame  # noqa: E402

CASES = [
    dict(name="orb_640x480_n1000_seed1234", seed=1234, w=640, h=480, nf=1000, lap=(0, 1000), kw={}),
    dict(name="orb_640x480_n1000_seed3_lowtex", seed=3, w=640, h=480, nf=1000, lap=(0, 1000),
         kw=dict(n_rect=60, contrast=(8, 19))),
    dict(name="orb_320x240_n500_seed11_lap00", seed=11, w=320, h=240, nf=500, lap=(0, 0), kw={}),
]


def run_case(c):
    img = synth_frame(c["seed"], w=c["w"], h=c["h"], **c["kw"])
    o = oracle_lib.OracleExtractor(c["nf"], 1.2, 8, 20, 7)
    mono, kps, desc = o.extract(img, c["lap"])
    lv = [hashlib.sha256(o.level(l).tobytes()).hexdigest() for l in range(8)]
    bl = [hashlib.sha256(o.level(l, True).tobytes()).hexdigest() if o.level(l, True) is not None else "" for l in range(8)]
    ncand = np.array([len(o.keypoints(l, False)) for l in range(8)], np.int32)
    nsel = np.array([len(o.keypoints(l, True)) for l in range(8)], np.int32)
    return dict(image_sha256=hashlib.sha256(img.tobytes()).hexdigest(), mono=np.int32(mono),
                kps=kps.view(np.uint8).reshape(-1, 28), desc=desc, level_sha256=np.array(lv), blur_sha256=np.array(bl),
                ncand=ncand, nsel=nsel, lap=np.array(c["lap"], np.int32), nfeatures=np.int32(c["nf"]),
                seed=np.int32(c["seed"]), wh=np.array([c["w"], c["h"]], np.int32))


if __name__ == "__main__":
    for c in CASES:
        out = run_case(c)
        np.savez_compressed(os.path.join(HERE, c["name"] + ".npz"), **out)
        print(c["name"], "n =", len(out["kps"]), "mono =", int(out["mono"]))
