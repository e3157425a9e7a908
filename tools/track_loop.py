"""100 frames through rumi_track_frame (for rocprofv3 --kernel-trace --stats: the kernels of one device-resident Tracking step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import track_probe
r = track_probe.measure(100)
print(r)
