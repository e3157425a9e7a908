#!/usr/bin/env python3
"""Regenerate the rBRIEF pattern DATA table from the reference source.

Reads the 256x4 integer table at /root/reference/src/rumi-slam/lib_src/ORBextractor.cc:145-403
(numbers only; comments dropped) and writes it, 8 test pairs per line, to
rumi_slam_amd/csrc/orb_pattern.inc and oracle/orb_pattern.inc.  Only runs where the reference
is mounted (the build container); the committed .inc files are what ships.
"""
import re, sys, pathlib

REF = pathlib.Path("/root/reference/src/rumi-slam/lib_src/ORBextractor.cc")
ROOT = pathlib.Path(__file__).resolve().parent.parent

def read_rows():
    lines = REF.read_text().split("\n")[146:402]
    rows = []
    for l in lines:
        l = re.sub(r"/\*.*?\*/", "", l)
        nums = [int(x) for x in re.findall(r"-?\d+", l)]
        if len(nums) == 4:
            rows.append(nums)
    assert len(rows) == 256, len(rows)
    return rows

def render(rows):
    out = ["// rBRIEF sampling pattern: 256 test pairs (x1,y1,x2,y2), int8, row i = descriptor bit i.",
           "// DATA ONLY. Values are the learned ORB pattern (Rublee et al. 2011) as tabulated in the",
           "// reference at lib_src/ORBextractor.cc:145-403; regenerate with tools/gen_pattern.py.",
           "// 8 pairs (= one descriptor byte) per line."]
    for i in range(0, 256, 8):
        out.append("  " + " ".join(",".join(f"{v:d}" for v in rows[j]) + "," for j in range(i, i + 8)))
    return "\n".join(out) + "\n"

if __name__ == "__main__":
    text = render(read_rows())
    if "--check" in sys.argv:
        for p in ("rumi_slam_amd/csrc/orb_pattern.inc", "oracle/orb_pattern.inc"):
            assert (ROOT / p).read_text() == text, p
        print("pattern tables match the reference")
    else:
        for p in ("rumi_slam_amd/csrc/orb_pattern.inc", "oracle/orb_pattern.inc"):
            (ROOT / p).write_text(text)
