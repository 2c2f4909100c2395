"""The algebraic identities the round-2 kernels rest on, in plain numpy (CPU): if one of these did not hold, the kernels that use it
could not agree with the reference however they were written.  (The kernels themselves are checked against the reference's
goldens and the CPU checker on the GPU: tests/test_hip_parity.py.)"""
import numpy as np
import pytest


def _conv3x3_reflect(x, w):
    """x [C, H, W] -> [O, H, W]; w [O, C, 3, 3]; ReflectionPad2d(1) + 3x3 conv (model_cd.py's layer), fp64."""
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    H, W = x.shape[1:]
    out = np.zeros((w.shape[0], H, W))
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, dy, dx], xp[:, dy:dy + H, dx:dx + W])
    return out


# R(a, i): taps of the 3-tap axis that fall on patch row i for output parity a (wct_api.hip pack_up_phase_f16 / pack_up_sp_f16)
RUNS = {(0, 0): (0,), (0, 1): (1, 2), (1, 0): (0, 1), (1, 1): (2,)}


@pytest.mark.parametrize("h,w", [(2, 2), (3, 5), (7, 4), (16, 9)])
def test_conv_behind_nearest_upsample_is_four_2x2_convs_of_the_low_resolution_map(h, w):
    """dec_tail_up_kernel / conv3x3_sp_up_kernel: conv3x3(reflect_pad(upsample2(x))) at output (2Y + a, 2X + b) is the 2x2 convolution of x
    with window origin (Y - 1 + a, X - 1 + b), CLAMPED coordinates and weights W_ab[i][j] = sum of the taps in R(a, i) x R(b, j)."""
    rng = np.random.default_rng(h * 31 + w)
    C, O = 5, 4
    x = rng.standard_normal((C, h, w))
    wt = rng.standard_normal((O, C, 3, 3))
    up = x.repeat(2, axis=1).repeat(2, axis=2)                     # nn.UpsamplingNearest2d(scale_factor=2), model_cd.py:90
    ref = _conv3x3_reflect(up, wt)
    got = np.zeros_like(ref)
    for a in range(2):
        for b in range(2):
            wab = np.zeros((O, C, 2, 2))
            for i in range(2):
                for j in range(2):
                    for dy in RUNS[(a, i)]:
                        for dx in RUNS[(b, j)]:
                            wab[:, :, i, j] += wt[:, :, dy, dx]
            for Y in range(h):
                for X in range(w):
                    acc = np.zeros(O)
                    for i in range(2):
                        for j in range(2):
                            yy = min(max(Y - 1 + a + i, 0), h - 1)
                            xx = min(max(X - 1 + b + j, 0), w - 1)
                            acc += wab[:, :, i, j] @ x[:, yy, xx]
                    got[:, 2 * Y + a, 2 * X + b] = acc
    assert np.abs(got - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())


def _l1_single(s, kq, u):
    """wct_common.h l1_single: (term, pos, zero) of lane group kq, K-step s, half u."""
    q, j = 4 * s + 2 * u + (kq >> 1), kq & 1
    if q == 15:
        return 0, 0, True
    r, m = divmod(q, 5)
    if m == 0:
        return 0, 3 * r + j, False
    if m == 1:
        return (0, 3 * r + 2, False) if j == 0 else (2, 3 * r, False)
    if m == 2:
        return 2, 3 * r + 1 + j, False
    if m == 3:
        return 1, 3 * r + j, False
    return 1, 3 * r + 2, j != 0


def test_conv11_single_order_covers_every_product_once_and_pairs_share_row_and_plane():
    """The K order of the 3-channel first conv: 27 (split term, window position) singles in 32 slots, each exactly once; the two lane
    groups one ds_read_b64 serves (kq = 2h, 2h + 1) always read the same window row of the same plane (term 1 = lo plane)."""
    seen = {}
    for s in range(4):
        for u in range(2):
            for h in range(2):
                a, b = _l1_single(s, 2 * h, u), _l1_single(s, 2 * h + 1, u)
                assert a[1] // 3 == b[1] // 3 and (a[0] == 1) == (b[0] == 1), (s, u, h, a, b)
                for t in (a, b):
                    if not t[2]:
                        seen[(t[0], t[1])] = seen.get((t[0], t[1]), 0) + 1
    assert len(seen) == 27 and set(seen.values()) == {1}
    assert {k[0] for k in seen} == {0, 1, 2} and {k[1] for k in seen} == set(range(9))


def test_24_channel_covariance_from_two_16x16_products():
    """l1_moments_kernel: rows ch 0..15 x cols ch 0..15 plus rows ch 8..23 x cols ch (16..23, 0..7) contain every unordered channel pair
    of 24 channels, with the scatter the kernel applies (tile (0,1) rows 8..15 direct, rows 0..7 transposed, tile (1,1) whole)."""
    rng = np.random.default_rng(3)
    X = rng.standard_normal((24, 200))
    full = X @ X.T
    p0 = X[:16] @ X[:16].T
    cols = np.r_[16:24, 0:8]
    p1 = X[8:24] @ X[cols].T
    out = np.full((24, 24), np.nan)
    out[:16, :16] = p0
    for r in range(16):
        for c in range(16):
            ra, cb = 8 + r, cols[c]
            if ra < 16 and cb >= 16:
                out[ra, cb] = out[cb, ra] = p1[r, c]
            elif ra >= 16 and cb >= 16:
                out[ra, cb] = p1[r, c]
            elif ra >= 16:
                out[cb, ra] = out[ra, cb] = p1[r, c]
    assert not np.isnan(out).any()
    assert np.array_equal(out, out.T) and np.abs(out - full).max() <= 1e-12 * np.abs(full).max()


def test_last_conv_over_2x2_pixel_blocks():
    """c3_block_compute: M = 4 (2 py + px) + cout, K over the 4 x 4 window a 2 x 2 block of outputs shares, weight
    w[cout][ch][wy - py][wx - px] where both offsets are in 0..2 and zero elsewhere -- equals the direct 3x3 convolution."""
    rng = np.random.default_rng(4)
    C = 16
    x = rng.standard_normal((C, 10, 12))
    wt = rng.standard_normal((3, C, 3, 3))
    ref = _conv3x3_reflect(x, wt)
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1)), mode="reflect")
    A = np.zeros((16, 4, 4, C))                                     # [m][wy][wx][ch]
    for py in range(2):
        for px in range(2):
            for co in range(3):
                for wy in range(4):
                    for wx in range(4):
                        dy, dx = wy - py, wx - px
                        if 0 <= dy <= 2 and 0 <= dx <= 2:
                            A[4 * (2 * py + px) + co, wy, wx] = wt[co, :, dy, dx]
    got = np.zeros_like(ref)
    for R in range(5):
        for Cc in range(6):
            win = xp[:, 2 * R:2 * R + 4, 2 * Cc:2 * Cc + 4]           # the block's 4 x 4 window (padded coordinates)
            d = np.einsum("myxc,cyx->m", A, win)
            for py in range(2):
                for px in range(2):
                    got[:, 2 * R + py, 2 * Cc + px] = d[4 * (2 * py + px):4 * (2 * py + px) + 3]
    assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
