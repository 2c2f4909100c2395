"""CPU-only checks of the drop-in boundary: the built library loads and exports every symbol that
include/wct_hip.h declares (no compute calls without a GPU), and the host mirror's error behaviour."""
import os
import re
import types

import pytest

from tests.conftest import REPO


def _build():
    import __graft_entry__ as g
    g.build()


def test_library_exports_every_declared_symbol():
    _build()
    from wct_hip import lib
    hdr = open(os.path.join(REPO, "include", "wct_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(wct_[a-z_0-9]+)\s*\(", hdr)) - {"wct_ctx"})
    assert declared and sorted(lib.SYMBOLS) == declared
    L = lib.load()
    for s in declared:
        assert hasattr(L, s), s
    assert L.wct_version() >= 1


def test_header_cites_reference_interfaces():
    hdr = open(os.path.join(REPO, "include", "wct_hip.h")).read()
    for cite in ("WCT.py:98-106", "util_wct.py:210-223", "WCT.py:120-125", "util_wct.py:68-70", "model_cd.py:724-743"):
        assert cite in hdr


def test_create_without_gpu_reports_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _build()
    import ctypes
    from wct_hip import lib
    L = lib.load()
    ctx = ctypes.c_void_p()
    assert L.wct_create(0, ctypes.byref(ctx)) == lib.WCT_ERR_HIP and not ctx.value
    assert L.wct_sync(None) == lib.WCT_ERR_INVALID  # NULL context is rejected, never dereferenced


def test_host_mirror_error_behaviour():
    import torch
    from wct_hip import WCT
    with pytest.raises(ValueError, match="Wrong mode"):          # util_wct.py:57-59 prints this and exits
        WCT(types.SimpleNamespace(mode="32x"))
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU path"):   # the product never falls back to the CPU
            WCT(types.SimpleNamespace(mode="16x"))


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(REPO, "collaborative-distillation_amd")
    bad = re.compile(r"(from|import)\s+oracle|wct_oracle|liboracle|oracle/")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert not bad.search(src), os.path.join(root, f)
