import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
if GOLDEN not in sys.path:
    sys.path.insert(0, GOLDEN)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _gpu_run_selected(config) -> bool:
    expr = config.getoption("markexpr", "") or ""
    return "gpu" in expr and "not gpu" not in expr


def pytest_sessionstart(session):
    """The two-rank data-parallel rehearsal (tests/test_gpu_distributed.py) needs its ranks started as FRESH processes
    BEFORE this process initialises the GPU (test modules call torch.cuda.is_available() when they are collected), so
    it runs here, once, and the tests only read what the ranks wrote.  torch.cuda.device_count() does not initialise
    the GPU on this image."""
    session.config._nsg_dp_dir = None
    if not _gpu_run_selected(session.config):
        return
    import tempfile
    import torch
    if torch.cuda.device_count() < 1:
        return
    from tests.helpers.spawn import run_ranks
    out = tempfile.mkdtemp(prefix="nsg_dp_")
    rcs, logs = run_ranks([os.path.join(ROOT, "tests", "helpers", "dp_rank.py"), os.path.join(GOLDEN, "model_tiny.npz"), out], 2,
                          extra_env={"NSG_DIST_BACKEND": "gloo", "NSG_DEVICE_INDEX": "0"}, log_dir=out)
    with open(os.path.join(out, "rcs.txt"), "w") as f:
        f.write(" ".join(str(r) for r in rcs))
    session.config._nsg_dp_dir = out


@pytest.fixture(scope="session")
def dp_rehearsal_dir(request):
    return getattr(request.config, "_nsg_dp_dir", None)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
