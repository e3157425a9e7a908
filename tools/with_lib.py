#!/usr/bin/env python3
"""Run a tool / bench script against a differently built library (experiments): with_lib.py <lib.so> <script.py> [args...]"""
import os, runpy, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rumi_slam_amd.capi as capi
capi.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
