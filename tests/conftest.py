"""pytest configuration: `gpu` marker, import paths, shared fixtures."""
import os
import subprocess
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "collaborative-distillation_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLD = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Plain `pytest tests` on a box without a GPU runs the CPU suite: gpu-marked tests are skipped there.  An explicit
    `-m gpu` on such a box is a configuration error, not a skip -- those tests then fail loudly (no CPU fallback exists)."""
    if config.getoption("-m"):
        return
    try:
        import torch
        have_gpu = torch.cuda.is_available()
    except Exception:
        have_gpu = False
    if have_gpu:
        return
    skip = pytest.mark.skip(reason="needs the MI355X (select with -m gpu on the GPU box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU checker (test infrastructure).  Builds oracle/liboracle_conv.so if missing."""
    so = os.path.join(REPO, "oracle", "liboracle_conv.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(REPO, "oracle")])
    from oracle import wct_oracle
    return wct_oracle


@pytest.fixture(scope="session")
def weights16x():
    from wct_hip import model_zoo
    return model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))


def load_golden(name):
    with np.load(os.path.join(GOLD, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max|a-b| / max|b|  -- the gate of BASELINE.md section 3.5"""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
