import os
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The GPU parity tests evaluate the fp32 torch oracle on the GPU: without this MIOpen times every solver (including
# naive ones) on the first call of each new conv shape -- 17 s for a 48x128x128 layer, 5 minutes at 48x512x512.
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

_T0 = time.time()


def pytest_runtest_logreport(report):
    """Per-test wall-clock into gpurun_out/test_times.log (when that scratch directory exists): a GPU run that is
    killed at its time limit still shows where the time went."""
    if report.when != "call":
        return
    out = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "test_times.log"), "a") as f:
            f.write(f"{time.time() - _T0:8.1f}s  {report.duration:7.2f}s  {report.outcome:7s} {report.nodeid}\n")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `pytest -m gpu` on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np
    return np.load(os.path.join(ROOT, "tests", "golden", "golden_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def pkg():
    import importlib
    return importlib.import_module("video-to-video-diffusion_amd")
