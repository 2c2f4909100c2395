"""CPU restatement of the reference harness's Resize step -- TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench's cpu_baseline may
import it; the product never does).

PytorchWCT/data_loader.py:52-56 applies torchvision 0.2.1 `transforms.Resize(size)` to the decoded PIL image:
    functional.resize(img, size:int): (w, h) = img.size; unchanged if the smaller edge already equals size; else the smaller edge
    becomes `size` and the other int(size * long / short);  img.resize((ow, oh), Image.BILINEAR)
The arithmetic is Pillow's (requirements.txt:  Pillow==8.2.0; its source is not under /root/reference).  Restated from Pillow's
published resampler, libImaging/Resample.c:
    precompute_coeffs         per output index: centre, support = max(scale, 1), taps [xmin, xmin + n), triangle weights normalised in double
    normalize_coeffs_8bpc     weights -> int with 22 fractional bits, round half away from zero
    ImagingResampleHorizontal_8bpc / Vertical_8bpc   acc = 2^21 + sum pix * w (int32);  out = clip8(acc >> 22)
    ImagingResample           horizontal pass first (only over the rows the vertical pass reads), uint8 in between
Pinned by tests/golden/g12_resize.npz (Pillow 12.2.0 outputs, tools/make_goldens.py gen_g12) and, where Pillow is importable, against
Pillow itself (tests/test_resize.py): bit-exact.
"""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2


def resize_shape(h: int, w: int, size: int):
    """torchvision 0.2.1 functional.resize's size rule for an int `size` (0: unchanged).  Returns (oh, ow)."""
    if size == 0 or (w <= h and w == size) or (h <= w and h == size):
        return h, w
    if w < h:
        return int(size * h / w), size
    return size, int(size * w / h)


def _triangle(x: float) -> float:
    if x < 0.0:
        x = -x
    return 1.0 - x if x < 1.0 else 0.0


def axis_tables(in_size: int, out_size: int):
    """(ksize, bounds[out, 2], kk[out, ksize] int32) of one axis for the box [0, in_size)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int64)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = max(int(center - support + 0.5), 0)          # C's (int) truncates toward zero; the argument is > -1 here
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = [_triangle((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            p = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + p * (1 << PRECISION_BITS)) if p < 0 else int(0.5 + p * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _clip8(acc: np.ndarray) -> np.ndarray:
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _pass(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One resampling pass of a uint8 H x W x 3 image along `axis` (0: rows / vertical, 1: columns / horizontal)."""
    _, bounds, kk = axis_tables(img.shape[axis], out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], np.uint8)
    for i in range(out_size):
        lo, n = int(bounds[i, 0]), int(bounds[i, 1])
        acc = (1 << (PRECISION_BITS - 1)) + np.tensordot(kk[i, :n].astype(np.int64), src[lo:lo + n], axes=(0, 0))
        acc = ((acc + (1 << 31)) % (1 << 32)) - (1 << 31)      # the C accumulator is a 32-bit int (never wraps for these weights)
        out[i] = _clip8(acc)
    return np.moveaxis(out, 0, axis)


def resize_bilinear_u8(img: np.ndarray, oh: int, ow: int) -> np.ndarray:
    """Image.resize((ow, oh), Image.BILINEAR) of a uint8 H x W x 3 image."""
    assert img.dtype == np.uint8 and img.ndim == 3 and img.shape[2] == 3
    h, w = img.shape[:2]
    out = img
    if ow != w:
        out = _pass(out, ow, 1)
    if oh != h:
        out = _pass(out, oh, 0)
    return np.ascontiguousarray(out)


def resize(img: np.ndarray, size: int) -> np.ndarray:
    """transforms.Resize(size) on a uint8 H x W x 3 image."""
    oh, ow = resize_shape(img.shape[0], img.shape[1], size)
    return resize_bilinear_u8(img, oh, ow)
