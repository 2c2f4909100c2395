"""transforms.Resize of the reference harness (PytorchWCT/data_loader.py:52-56 -> Pillow's bilinear Image.resize): the CPU checker
against the Pillow goldens (G12) and Pillow itself, then the device path against the checker -- bit for bit (uint8 work)."""
import ctypes
import os

import numpy as np
import pytest

from oracle import resize_oracle as R
from tests.conftest import load_golden


def _cases(g):
    for name in ("nat", "noise_p", "noise_l"):
        for key in g:
            if key.startswith(name + ".resize"):
                yield key, g[name], int(key.split("resize")[1]), None
    for key in g:
        if key.startswith("nat.to"):
            oh, ow = key[len("nat.to"):].split("x")
            yield key, g["nat"], None, (int(oh), int(ow))


def test_oracle_matches_pillow_goldens():
    g = load_golden("g12_resize.npz")
    n = 0
    for key, img, size, target in _cases(g):
        got = R.resize(img, size) if target is None else R.resize_bilinear_u8(img, *target)
        assert got.shape == g[key].shape and np.array_equal(got, g[key]), key
        n += 1
    assert n == 22


def test_oracle_matches_pillow_live():
    """Random shapes against the installed Pillow (the same 8-bit resampler since Pillow 3): shrinking by up to 60x, enlarging, 1-pixel edges."""
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(5)
    shapes = [tuple(int(v) for v in rng.integers(1, 160, 4)) for _ in range(60)]
    shapes += [(240, 426, 4, 7), (3, 500, 3, 499), (500, 3, 7, 3), (64, 64, 64, 64), (270, 480, 135, 240)]
    for (h, w, oh, ow) in shapes:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.asarray(Image.fromarray(img).resize((ow, oh), Image.BILINEAR))
        assert np.array_equal(R.resize_bilinear_u8(img, oh, ow), ref), (h, w, oh, ow)


def test_size_rule_host_function_matches_checker():
    """wct_resize_shape is host arithmetic of the C ABI (no device call): torchvision 0.2.1's rule on every small shape and some large ones."""
    from wct_hip import lib
    L = lib.load()
    oh, ow = ctypes.c_int(), ctypes.c_int()
    rng = np.random.default_rng(1)
    cases = [(h, w, s) for h in range(1, 24) for w in range(1, 24) for s in (0, 1, 5, 16, 23)]
    cases += [tuple(int(v) for v in rng.integers(1, 12000, 3)) for _ in range(2000)]
    cases += [(2160, 3840, 512), (3840, 2160, 512), (4096, 10240, 1024), (100, 100, 50)]
    for (h, w, s) in cases:
        assert L.wct_resize_shape(h, w, s, ctypes.byref(oh), ctypes.byref(ow)) == 0
        assert (oh.value, ow.value) == R.resize_shape(h, w, s), (h, w, s)
    assert L.wct_resize_shape(0, 5, 3, ctypes.byref(oh), ctypes.byref(ow)) != 0
    assert L.wct_resize_shape(5, 5, -1, ctypes.byref(oh), ctypes.byref(ow)) != 0


# ---------------------------------------------------------------------------------------------------------------- device
@pytest.fixture(scope="module")
def wct(weights16x):
    import types
    from wct_hip import WCT
    return WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)


@pytest.mark.gpu
def test_device_resize_matches_pillow_goldens(wct):
    import torch
    g = load_golden("g12_resize.npz")
    for key, img, size, target in _cases(g):
        x = torch.from_numpy(img).cuda()
        got = wct.resize_u8(x, size if target is None else target)
        assert tuple(got.shape) == g[key].shape and np.array_equal(got.cpu().numpy(), g[key]), key
        # the fused ToTensor form: exactly to_tensor_u8 of the uint8 result
        f = wct.resize_u8(x, size if target is None else target, to_tensor=True)
        assert torch.equal(f, wct.to_tensor_u8(got)), key


@pytest.mark.gpu
def test_device_resize_full_size_vs_checker(wct):
    """BASELINE config-2 sized content (3840x2160) to the usual working sizes, and the 2048^2 style enlarged: against the CPU checker
    (and Pillow where importable), bit for bit; then the property that holds at any size: resizing a constant image returns the constant."""
    import torch
    rng = np.random.default_rng(2)
    img = rng.integers(0, 256, (2160, 3840, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    for size in (512, 1080, 1000):
        got = wct.resize_u8(x, size).cpu().numpy()
        ref = R.resize(img, size)
        assert got.shape == ref.shape and np.array_equal(got, ref), size
    try:
        from PIL import Image
        ref = np.asarray(Image.fromarray(img).resize((1234, 777), Image.BILINEAR))
        assert np.array_equal(wct.resize_u8(x, (777, 1234)).cpu().numpy(), ref)
    except ImportError:
        pass
    small = rng.integers(0, 256, (300, 200, 3), dtype=np.uint8)
    up = wct.resize_u8(torch.from_numpy(small).cuda(), (1500, 1000)).cpu().numpy()
    assert np.array_equal(up, R.resize_bilinear_u8(small, 1500, 1000))
    const = torch.full((2160, 3840, 3), 173, dtype=torch.uint8, device="cuda")
    for target in ((540, 960), (4320, 7680), (2160, 100), (7, 3840)):
        out = wct.resize_u8(const, target)
        assert tuple(out.shape) == (target[0], target[1], 3) and bool((out == 173).all())


@pytest.mark.gpu
def test_device_resize_edges_and_errors(wct):
    import torch
    rng = np.random.default_rng(3)
    for (h, w, oh, ow) in ((1, 1, 5, 7), (1, 37, 1, 11), (41, 1, 9, 1), (5, 7, 1, 1), (33, 65, 33, 65), (2, 3, 200, 300)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        got = wct.resize_u8(torch.from_numpy(img).cuda(), (oh, ow)).cpu().numpy()
        assert np.array_equal(got, R.resize_bilinear_u8(img, oh, ow)), (h, w, oh, ow)
    # a contiguous uint8 view at an ODD byte offset (e.g. a row crop of a larger HWC buffer): the kernels read single bytes
    img = rng.integers(0, 256, (37, 53, 3), dtype=np.uint8)
    buf = torch.zeros(1 + img.size, dtype=torch.uint8, device="cuda")
    buf[1:] = torch.from_numpy(img).cuda().reshape(-1)
    view = buf[1:].view(37, 53, 3)
    assert view.data_ptr() % 2 == 1
    assert np.array_equal(wct.resize_u8(view, (20, 31)).cpu().numpy(), R.resize_bilinear_u8(img, 20, 31))
    assert torch.equal(wct.resize_u8(view, (20, 31), to_tensor=True), wct.to_tensor_u8(wct.resize_u8(view, (20, 31))))
    with pytest.raises(ValueError):
        wct.resize_u8(torch.zeros((4, 4, 3), dtype=torch.uint8, device="cuda"), (0, 4))
    with pytest.raises(ValueError):
        wct.resize_u8(torch.zeros((4, 4), dtype=torch.uint8, device="cuda"), 2)
    # many distinct sizes: the context's table cache is bounded and recycles
    img = rng.integers(0, 256, (50, 60, 3), dtype=np.uint8)
    x = torch.from_numpy(img).cuda()
    for k in range(1, 40):
        assert np.array_equal(wct.resize_u8(x, (k, 2 * k)).cpu().numpy(), R.resize_bilinear_u8(img, k, 2 * k)), k
        if k % 5 == 0:      # a size that keeps coming back survives the evictions (least recently used goes first)
            assert np.array_equal(wct.resize_u8(x, (3, 4)).cpu().numpy(), R.resize_bilinear_u8(img, 3, 4))
