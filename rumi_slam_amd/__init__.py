"""MI355X-native ORB front-end / Hamming matcher / local-BA hot path of RUMI-SLAM.

Host side = ctypes over the C-ABI in ``include/*.h`` (``librumi_hip.so``, hand-written HIP for
gfx950).  There is no CPU fallback: every operator raises if the HIP library is missing.
"""
__version__ = "0.1.0"
