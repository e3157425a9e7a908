import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        from rumi_slam_amd import capi
        return capi.lib().rumi_device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    # -m gpu on a box without a GPU must fail loudly, not skip silently: only auto-skip when the user did
    # not ask for the gpu marker explicitly.
    if "gpu" in (config.getoption("-m") or ""):
        return
    if not _has_gpu():
        skip = pytest.mark.skip(reason="no GPU here")
        for it in items:
            if "gpu" in it.keywords:
                it.add_marker(skip)


@pytest.fixture(scope="session", autouse=True)
def _build_libs():
    import subprocess
    so = os.path.join(ROOT, "rumi_slam_amd", "librumi_hip.so")
    if not os.path.exists(so):
        subprocess.check_call([sys.executable, os.path.join(ROOT, "__graft_entry__.py")])
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
