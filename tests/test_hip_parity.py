"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libwct_hip.so), against
  (1) golden vectors generated from the reference itself (tests/golden, tools/make_goldens.py),
  (2) the CPU oracle on seeded inputs at sizes it finishes in seconds,
  (3) size-independent properties at BASELINE.json's full sizes.
Tolerances: BASELINE.json north_star = 1e-3 relative (max|d|/max|ref|) end to end; individual operators are
held much tighter (fp32 convs differ from the reference only by summation order)."""
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from tests.conftest import PKG, REPO, rel_err
from wct_hip import model_zoo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


@pytest.fixture(scope="module")
def wct16(torch_cuda, weights16x):
    from wct_hip import WCT
    return WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)


def cu(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


def test_native_library_is_loaded(wct16):
    import os
    from wct_hip import lib
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(lib.LIB_PATH) in maps  # the in-tree HIP library is what runs, no fallback


# --------------------------------------------------------------------------- G2 modules
@pytest.mark.parametrize("k", [1, 2, 3, 4, 5])
def test_g2_encoders_decoders(torch_cuda, wct16, golden, k):
    g = golden("g2_modules.npz")
    enc, dec = getattr(wct16, "e%d" % k), getattr(wct16, "d%d" % k)
    f = enc(cu(torch_cuda, g["img"])).cpu().numpy()
    assert f.shape == g["e%d.y" % k].shape
    assert rel_err(f, g["e%d.y" % k]) < 2e-5
    y = dec(cu(torch_cuda, g["d%d.x" % k])).cpu().numpy()
    assert y.shape == g["d%d.y" % k].shape
    assert rel_err(y, g["d%d.y" % k]) < 2e-5
    fo = enc(cu(torch_cuda, g["img_odd"])).cpu().numpy()   # 45x37: floor pooling drops odd rows/cols
    assert fo.shape == g["e%d.y_odd" % k].shape
    assert rel_err(fo, g["e%d.y_odd" % k]) < 2e-5


def test_nhwc_and_nchw_agree(torch_cuda, wct16, golden):
    g = golden("g2_modules.npz")
    a = wct16.encode(3, cu(torch_cuda, g["img"]), layout="nchw")
    b = wct16.encode(3, cu(torch_cuda, g["img"]), layout="nhwc")
    assert torch_cuda.equal(a, b.permute(0, 3, 1, 2))
    ya = wct16.decode(3, a, layout="nchw")
    yb = wct16.decode(3, b, layout="nhwc")
    assert torch_cuda.equal(ya, yb)


# --------------------------------------------------------------------------- G3 transform
CASES = ["fullrank24", "fullrank24_a06", "dead32", "hw_lt_C_content", "hw_lt_C_style", "illcond64"]


@pytest.mark.parametrize("case", CASES)
def test_g3_transform(torch_cuda, wct16, golden, case):
    g = golden("g3_transform.npz")
    cF, sF, a = g[case + ".cF"], g[case + ".sF"], float(g[case + ".alpha"])
    # CPU tensors in, like WCT.py:102-104
    csF = torch_cuda.empty(0, device="cuda")
    out = wct16.transform(torch_cuda.from_numpy(cF), torch_cuda.from_numpy(sF), csF, a)
    assert out is csF and tuple(out.shape) == g[case + ".out"].shape  # same object, resized (util_wct.py:221)
    assert rel_err(out.cpu().numpy(), g[case + ".out"]) < 1e-5, case


@pytest.mark.parametrize("C,n", [(24, 4000), (32, 999), (64, 2500), (128, 20000), (128, 77)])
def test_moments_and_solve_split_form(torch_cuda, wct16, oracle, C, n):
    rng = np.random.default_rng(C * 1000 + n)
    h = 7 if n % 7 == 0 else (11 if n % 11 == 0 else 1)
    w = n // h
    n = h * w
    f = np.maximum(rng.standard_normal((h, w, C)).astype(np.float32) + 0.3, 0)
    f[..., 1] = 0                      # a dead channel
    s = np.maximum(rng.standard_normal((9, 31, C)).astype(np.float32) * 1.5 + 0.1, 0)
    x0, x1 = (0, w) if w < 8 else (3, w - 2)   # window columns: what a content strip owns
    nc, sc, ssc = wct16.moments(cu(torch_cuda, f)[None], x0, x1)
    X = f[:, x0:x1].reshape(-1, C).astype(np.float64)
    assert nc == X.shape[0]
    assert rel_err(sc.cpu().numpy(), X.sum(0)) < 1e-13
    assert rel_err(ssc.cpu().numpy(), X.T @ X) < 1e-13
    assert np.array_equal(ssc.cpu().numpy(), ssc.cpu().numpy().T)
    ns, ss_, sss = wct16.moments(cu(torch_cuda, s)[None])
    if nc < 2:
        return
    M, b, info = wct16.solve(nc, sc, ssc, ns, ss_, sss, alpha=0.8, want_info=True)
    _, mc, cc = oracle.moments(np.ascontiguousarray(f[:, x0:x1].transpose(2, 0, 1)))
    _, ms, cs = oracle.moments(np.ascontiguousarray(s.transpose(2, 0, 1)))
    Mr, br = oracle.affine_from_moments(mc, cc, ms, cs, 0.8)
    # info: Newton-Schulz iterations (< 40), or 100 + Jacobi sweeps when the fallback ran (singular covariance:
    # fewer pixels than channels) -- either way it converged
    for i in info:
        assert 0 < i < 40 or 100 < i < 140
    if n > 4 * C:
        assert info[0] < 40    # well-conditioned content covariance: the GEMM path must have handled it
    assert rel_err(M.cpu().numpy(), Mr) < 1e-8 and rel_err(b.cpu().numpy(), br) < 1e-8


@pytest.mark.parametrize("C,h,w,x0,x1", [(512, 80, 64, 3, 62), (512, 134, 240, 0, 240), (512, 67, 120, 7, 120), (256, 90, 100, 4, 97), (256, 134, 240, 0, 240),
                                         (384, 64, 80, 1, 79)])
def test_moments_wide_maps_vs_numpy(torch_cuda, wct16, C, h, w, x0, x1):
    """The wide levels of --mode original (C = 256 / 512): fp64-product moments against an INDEPENDENT fp64 reference (numpy X^T X)
    at round-off -- C >= 512 with >= 4096 pixels in the window takes the channel-blocked register kernel (moments.hip moments_blk_kernel<false, 4>:
    every 128 x 128 block pair, diagonal and off-diagonal, windowed and whole maps), C = 256 / 384 and smaller windows the LDS kernel."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(C + h)
    f = torch.relu(torch.randn((1, h, w, C), device="cuda", generator=g) + 0.3)
    f[..., 1] = 0
    f[..., C - 3] *= 1e-3
    wct16.debug_set("mom32", 0)          # fp64 products at every size (the default for maps below 65 536 pixels)
    try:
        n, sm, sq = wct16.moments(f, x0, x1)
    finally:
        wct16.debug_set("mom32", 1)
    X = f[0, :, x0:x1].reshape(-1, C).double().cpu().numpy()
    assert n == X.shape[0]
    assert rel_err(sm.cpu().numpy(), X.sum(0)) < 1e-13
    q = sq.cpu().numpy()
    assert rel_err(q, X.T @ X) < 1e-13 and np.array_equal(q, q.T)


def test_moments_blocked_and_lds_kernels_agree():
    """The same C = 512 moments through the channel-blocked register kernel (default) and the LDS kernel it replaced (WCT_MOM_BLK=0, an
    environment switch read once per process: two fresh processes), fp64 products: equal to fp64 round-off, windowed and whole."""
    code = r"""
import os, sys, types
sys.path[:0] = [%r, %r]
import numpy as np, torch
from wct_hip import WCT, model_zoo
e = WCT(types.SimpleNamespace(mode='16x', alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(%r, 'weights', '16x.npz')))
g = torch.Generator(device='cuda').manual_seed(5)
f = torch.relu(torch.randn((1, 134, 240, 512), device='cuda', generator=g) + 0.3)
e.debug_set('mom32', 0)
out = {}
for k, (a, b) in {'whole': (0, 240), 'window': (16, 200)}.items():
    n, s, q = e.moments(f, a, b)
    out[k + '.s'], out[k + '.q'] = s.cpu().numpy(), q.cpu().numpy()
np.savez(sys.argv[1], **out)
""" % (REPO, PKG, PKG)
    import tempfile
    res = []
    with tempfile.TemporaryDirectory() as d:
        for blk in ("1", "0"):
            path = os.path.join(d, "m%s.npz" % blk)
            r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, WCT_DEBUG="1", WCT_MOM_BLK=blk))
            assert r.returncode == 0, r.stderr[-2000:]
            res.append(dict(np.load(path)))
    for k in res[0]:
        assert rel_err(res[0][k], res[1][k]) < 1e-13, k
    assert not all(np.array_equal(res[0][k], res[1][k]) for k in res[0])     # two different kernels really ran (summation orders differ)


@pytest.mark.parametrize("C,h,w", [(32, 540, 960), (64, 270, 480), (128, 300, 260), (256, 270, 262), (512, 135, 512), (24, 0, 0)])
def test_moments_fp32_block_products(torch_cuda, weights16x, C, h, w):
    """Maps of >= 65 536 pixels take their moments with fp32 products: 64-pixel blocks on v_mfma_f32_16x16x4_f32, block sums added in
    fp64 (moments.hip F32 variant, debug key "mom32", VERDICT r3 task 7) -- twice the matrix-core rate and no conversions in front of
    the operands.  Against the fp64 form (mom32 = 0, itself 1e-13 from numpy, test above): the raw sums to 1e-6 of their largest
    entry (expected ~1e-8: zero-mean block rounding averaged over >= 1024 blocks), the COVARIANCE it implies to 2e-6 of its largest
    entry, symmetric, the pixel count equal; smaller maps are untouched (bitwise the fp64 form).  C = 24: the fused level-1 kernel
    (image -> conv11 -> moments), through content_encode."""
    from wct_hip import WCT
    torch = torch_cuda
    e = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
    g = torch.Generator(device="cuda").manual_seed(C)
    if C == 24:
        img = torch.rand((1, 3, 600, 800), device="cuda", generator=g)
        res = {}
        for m in (1, 0):
            e.debug_set("mom32", m)
            _, _, sm, sq = e.content_encode(1, img)
            res[m] = (600 * 800, sm.clone(), sq.clone())
    else:
        f = torch.relu(torch.randn((1, h, w, C), device="cuda", generator=g) + 0.3)
        f[..., 1] = 0
        f[..., 2] *= 1e-3                  # a weak channel beside strong ones
        res = {}
        for m in (1, 0):
            e.debug_set("mom32", m)
            n, sm, sq = e.moments(f, 5, w - 3)
            res[m] = (n, sm.clone(), sq.clone())
        small = f[:, :100, :200].contiguous()      # 19 400 pixels in the window: below the threshold, fp64 either way
        e.debug_set("mom32", 1)
        a = e.moments(small, 3, 197)
        e.debug_set("mom32", 0)
        b = e.moments(small, 3, 197)
        assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    (n1, s1, q1), (n0, s0, q0) = res[1], res[0]
    assert n1 == n0
    s1, q1, s0, q0 = (t.cpu().numpy() for t in (s1, q1, s0, q0))
    assert np.array_equal(q1, q1.T)
    es, eq = rel_err(s1, s0), rel_err(q1, q0)
    cov = lambda s_, q_: (q_ - np.outer(s_, s_) / n0) / (n0 - 1)     # noqa: E731
    ec = rel_err(cov(s1, q1), cov(s0, q0))
    print("\n[moments fp32 blocks C=%d n=%d] sum %.2e  sumsq %.2e  covariance %.2e" % (C, n0, es, eq, ec))
    assert es < 1e-6 and eq < 1e-6 and ec < 2e-6
    assert not np.array_equal(q1, q0)      # the fp32 path really ran


@pytest.mark.parametrize("C,lmin,dead,rank", [(128, 1e-8, 9, None), (128, 1e-15, 0, None), (256, 3e-11, 69, None), (512, 1e-9, 0, None),
                                              (512, 1e-7, 7, 100), (256, 1e-6, 0, 255), (512, 1e-5, 0, 15)])
def test_solve_ill_conditioned_and_singular(torch_cuda, wct16, oracle, C, lmin, dead, rank):
    """Covariances with a prescribed spectrum: 1 .. lmin log-uniform over `rank` directions (None: all live ones), exact zeros
    beyond, plus exactly-dead channels.  The reference keeps every singular value >= 1e-100 (util_wct.py:25,82-86): a GENUINE
    direction 1e-11 below the top one is whitened like any other (--mode original on generated weights has them;
    tools/experiments/original_parity.py), while the null space of a singular covariance meets exactly-zero centred
    components, i.e. is dropped.  C <= 128: Newton-Schulz or the LDS Jacobi; C > 128: the deflated iteration."""
    rng = np.random.default_rng(C + dead)
    live = C - dead
    def spd(lo, scale, k):
        Q, _ = np.linalg.qr(rng.standard_normal((live, live)))
        lam = np.zeros(live)
        lam[:k] = scale * np.exp(np.linspace(0.0, np.log(lo), k))
        A = np.zeros((C, C))
        idx = np.sort(rng.permutation(C)[:live])
        A[np.ix_(idx, idx)] = (Q * lam) @ Q.T
        return (A + A.T) / 2
    cov_c, cov_s = spd(lmin, 40.0, rank or live), spd(1e-6, 3.0, live)
    mu_c, mu_s = rng.random(C) * (cov_c.diagonal() > 0), rng.random(C) * (cov_s.diagonal() > 0)
    dev = lambda a: torch_cuda.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()
    raw = lambda n, mu, cov: (n, dev(n * mu), dev((n - 1) * cov + n * np.outer(mu, mu)))
    M, b, info = wct16.solve(*raw(50000.0, mu_c, cov_c), *raw(20000.0, mu_s, cov_s), alpha=1.0, want_info=True)
    Mr, br = oracle.affine_from_moments(mu_c, cov_c, mu_s, cov_s, 1.0)
    if lmin > 1e-12 and rank is None:
        # nothing is at the round-off level: the policy keeps everything, as the reference does (SVD + 1e-100)
        Mk, _ = oracle.affine_from_moments(mu_c, cov_c, mu_s, cov_s, 1.0, rel_thresh=1e-14, abs_floor=1e-16)
        assert rel_err(Mk, Mr) < 1e-9
    if C > 128:
        assert 0 < info[0] < 60, info      # GEMMs only: the global-memory Jacobi (100 + sweeps) costs 27..68 ms per matrix
    # |M| ~ lmin^-1/2; the error is ~10 cond eps relative to the largest entry
    tol = 3e-4 if lmin < 1e-10 else 1e-6
    assert rel_err(M.cpu().numpy(), Mr) < tol and rel_err(b.cpu().numpy(), br) < tol, (info, rel_err(M.cpu().numpy(), Mr))


def test_single_launch_newton_schulz_is_the_multi_launch_one(torch_cuda, weights16x, oracle):
    """C = 128 (levels 5 and 4 of --mode 16x): the coupled iteration as ONE launch on one XCD (solve.hip ns_coop128_kernel, software
    barrier, iterates through that XCD's L2) against the 2 x 16 stage launches (debug key "nscoop" 0): the same tile products in
    the same order -> (M, b) and the iteration counts bit for bit, for a well-conditioned and an ill-conditioned pair, alone and
    while the other lane keeps the GPU busy (a whole overlapped cascade, bitwise).  And its safety net: with a participant
    reported on the wrong XCD ("nscoop" 2) everyone leaves, the outcome says "not converged" and the gated Jacobi launch solves -- and
    the aborts are counted, and stop the lane from trying again once they are the rule."""
    from wct_hip import WCT
    torch = torch_cuda
    C = 128
    rng = np.random.default_rng(5)
    def spd(lo, scale):
        Q, _ = np.linalg.qr(rng.standard_normal((C, C)))
        return (Q * (scale * np.exp(np.linspace(0.0, np.log(lo), C)))) @ Q.T
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float64)).cuda()
    raw = lambda n, mu, cov: (n, dev(n * mu), dev((n - 1) * (cov + cov.T) / 2 + n * np.outer(mu, mu)))
    cases = [(spd(1e-3, 5.0), spd(1e-2, 2.0), rng.random(C), rng.random(C)), (spd(1e-6, 40.0), spd(1e-5, 3.0), rng.random(C), rng.random(C))]
    gen = torch.Generator(device="cuda").manual_seed(9)
    c = torch.rand((1, 3, 1024, 1536), device="cuda", generator=gen)
    s = torch.rand((1, 3, 768, 1024), device="cuda", generator=gen)
    res = {}
    for mode in (1, 0, 2):
        w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        w.debug_set("nscoop", mode)
        out = []
        for cov_c, cov_s, mu_c, mu_s in cases:
            M, b, info = w.solve(*raw(50000.0, mu_c, cov_c), *raw(20000.0, mu_s, cov_s), alpha=1.0, want_info=True)
            Mr, br = oracle.affine_from_moments(mu_c, (cov_c + cov_c.T) / 2, mu_s, (cov_s + cov_s.T) / 2, 1.0)
            assert rel_err(M.cpu().numpy(), Mr) < 1e-6 and rel_err(b.cpu().numpy(), br) < 1e-6, (mode, info)
            out.append((M.clone(), b.clone(), info))
        out.append(w.stylize(c, s).clone())
        res[mode] = out
        # health counters (wct_debug_get): an aborted single-launch solve is repaired by the Jacobi net but costs the watchdog + ~2 ms, so it
        # is counted per lane, and a lane on which the single launch keeps failing goes back to the multi-launch schedule
        aborts, off, solves = w.debug_get("nscoop_aborts"), int(w.debug_get("nscoop_off")), w.debug_get("nscoop_solves")
        if mode == 1:
            assert aborts == 0 and off == 0 and solves >= 8, (aborts, off, solves)
        elif mode == 0:
            assert aborts == 0 and solves == 0
        else:
            assert aborts >= 4 and (off & 1), (aborts, off, solves)      # every solve of the injected-fault run aborted: the main lane gave up
    # channel counts that are padded to the kernel's 128 (identity block in the padding), with dead channels among the live ones
    for Cq, dead in ((98, 0), (126, 5), (112, 17)):
        live = Cq - dead
        def spd_q(lo, scale):
            Q, _ = np.linalg.qr(rng.standard_normal((live, live)))
            A = np.zeros((Cq, Cq))
            idx = np.sort(rng.permutation(Cq)[:live])
            A[np.ix_(idx, idx)] = (Q * (scale * np.exp(np.linspace(0.0, np.log(lo), live)))) @ Q.T
            return (A + A.T) / 2
        cov_c, cov_s = spd_q(1e-4, 7.0), spd_q(1e-3, 2.0)
        mu_c, mu_s = rng.random(Cq) * (cov_c.diagonal() > 0), rng.random(Cq) * (cov_s.diagonal() > 0)
        got = {}
        for mode in (1, 0):
            w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
            w.debug_set("nscoop", mode)
            got[mode] = w.solve(*raw(30000.0, mu_c, cov_c), *raw(9000.0, mu_s, cov_s), alpha=0.7, want_info=True)
        assert torch.equal(got[1][0], got[0][0]) and torch.equal(got[1][1], got[0][1]) and got[1][2] == got[0][2], (Cq, got[1][2], got[0][2])
        assert all(0 < i < 40 for i in got[1][2]), (Cq, got[1][2])
        Mr, br = oracle.affine_from_moments(mu_c, cov_c, mu_s, cov_s, 0.7)
        assert rel_err(got[1][0].cpu().numpy(), Mr) < 1e-6 and rel_err(got[1][1].cpu().numpy(), br) < 1e-6, Cq
    for (M1, b1, i1), (M0, b0, i0), (M2, b2, i2) in zip(res[1][:2], res[0][:2], res[2][:2]):
        assert torch.equal(M1, M0) and torch.equal(b1, b0) and i1 == i0 and all(0 < i < 40 for i in i1), (i1, i0)
        assert all(100 < i < 140 for i in i2), i2        # the Jacobi net ran (100 + sweeps) ...
        assert rel_err(M2.cpu().numpy(), M1.cpu().numpy()) < 1e-6   # ... and found the same map
    assert torch.equal(res[1][2], res[0][2])
    assert float((res[2][2] - res[1][2]).abs().max() / res[1][2].abs().max()) < 1e-3


# --------------------------------------------------------------------------- G4 cascade
@pytest.mark.parametrize("tag", ["a", "b"])
def test_g4_cascade(torch_cuda, wct16, golden, tag):
    from wct_hip.wct import styleTransfer
    g = golden("g4_cascade.npz")
    style = cu(torch_cuda, g[tag + ".style"])[None]
    img = g[tag + ".content"]
    for k in (5, 4, 3, 2, 1):     # level-isolated: the golden previous output is the content
        y = styleTransfer(getattr(wct16, "e%d" % k), getattr(wct16, "d%d" % k), cu(torch_cuda, img)[None], style, None)
        assert rel_err(y.cpu().numpy()[0], g["%s.L%d.out" % (tag, k)]) < 1e-4, k
        img = g["%s.L%d.out" % (tag, k)]
    out = wct16.stylize(cu(torch_cuda, g[tag + ".content"]), style).cpu().numpy()[0]
    ref = g[tag + ".final"]
    assert out.shape == ref.shape
    assert rel_err(out, ref) < 1e-3                                    # north_star tolerance
    assert np.allclose(out, ref, rtol=1e-3, atol=1e-3 * float(ref.max()))


def test_unfused_reference_sequence_matches_fused(torch_cuda, wct16, golden):
    """encoder -> transform -> decoder through the reference's literal call sequence (WCT.py:100-105)
    equals the fused level (M, b folded into the decoder's first conv) to fp32 round-off."""
    g = golden("g4_cascade.npz")
    c, s = cu(torch_cuda, g["b.content"])[None], cu(torch_cuda, g["b.style"])[None]
    for k in (5, 3, 1):
        e, d = getattr(wct16, "e%d" % k), getattr(wct16, "d%d" % k)
        sF, cF = e(s), e(c)
        csF = wct16.transform(cF.squeeze(0).cpu(), sF.squeeze(0).cpu(), torch_cuda.empty(0, device="cuda"), 1.0)
        unfused = d(csF).cpu().numpy()
        fused = wct16.style_transfer_level(k, c, s).cpu().numpy()
        assert rel_err(fused, unfused) < 2e-5, k


def test_fast_fold_matches_the_map_based_fold(torch_cuda, weights16x, golden):
    """The cascade folds csF = M cF + b into the decoder's first conv.  Default: (W Ss) Wc with the style-side product formed on
    the side stream (misc.hip fold_style_kernel / fold_fast_kernel: no T = Ss Wc, no M, no b on the critical path); debug switch
    "fastfold" 0: M and b first (launch_assemble), then W M (fold_row_kernel) -- the form the split-level API keeps.  Same fp64
    arithmetic in another association -> fp32 round-off agreement per level, alpha = 1 and the alpha-blend, a constant content
    (cov_c = 0: Wc = 0) included; and the profile shows that the default path really runs without launch_assemble."""
    from wct_hip import WCT
    torch = torch_cuda
    g = golden("g4_cascade.npz")
    c, s = cu(torch, g["b.content"])[None], cu(torch, g["b.style"])[None]
    const = torch.full((1, 3, 64, 80), 0.3, device="cuda")
    outs = {}
    for ff in (1, 0):
        w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        w.debug_set("fastfold", ff)
        w.profile_reset(); w.profile(True)
        outs[ff] = [w.style_transfer_level(k, c, s, alpha).clone() for k in (5, 4, 3, 2, 1) for alpha in (1.0, 0.6)]
        outs[ff] += [w.style_transfer_level(k, const, s).clone() for k in (3, 1)]
        outs[ff].append(w.stylize(c, s).clone())
        w.profile(False)
        names = {e["name"] for e in w.profile_read()}
        assert ("assemble_Mb" in names) == (ff == 0) and ("fold_style" in names) == (ff == 1), names
    for a, b in zip(outs[1][:-1], outs[0][:-1]):
        assert a.shape == b.shape and float((a - b).abs().max() / b.abs().max()) < 2e-6
    assert float((outs[1][-1] - outs[0][-1]).abs().max() / outs[0][-1].abs().max()) < 1e-4     # five levels chained


def test_interleaved_enqueue_changes_the_schedule_not_the_result(torch_cuda, weights16x):
    """wct_stylize enqueues the style side of level L - 1 behind the content side of level L (so that a cold call's first content
    kernel is not queued behind ~400 style-side launches); debug switch "interleave" 0 = all five style sides up front.  Only the
    order in which independent kernels are ENQUEUED changes: bitwise the same image, with two runs (the style side belongs to the
    first), in both model widths, through the uint8 entry and with the side lane switched off."""
    from wct_hip import WCT, model_zoo
    torch = torch_cuda
    gen = torch.Generator(device="cuda").manual_seed(21)
    c = torch.rand((1, 3, 208, 272), device="cuda", generator=gen)
    s = torch.rand((1, 3, 176, 240), device="cuda", generator=gen)
    c8 = (c[0].permute(1, 2, 0) * 255).round().to(torch.uint8).contiguous()
    s8 = (s[0].permute(1, 2, 0) * 255).round().to(torch.uint8).contiguous()
    for mode, weights in (("16x", weights16x), ("original", model_zoo.synth_weights("original", 7))):
        outs = []
        for interleave, overlap in ((1, True), (0, True), (1, False)):
            w = WCT(types.SimpleNamespace(mode=mode, alpha=1.0), weights=weights)
            w.debug_set("interleave", interleave)
            w.set_overlap(overlap)
            outs.append((w.stylize(c, s).clone(), w.stylize(c, s, num_run=2).clone(), w.stylize_u8(c8, s8).clone()))
        for k, other in enumerate(outs[1:]):
            for j, (a, b) in enumerate(zip(outs[0], other)):
                assert torch.equal(a, b), (mode, k, j, float((a.float() - b.float()).abs().max()))


def test_wide_fold_as_matrix_core_gemm_matches_the_valu_fold(torch_cuda):
    """--mode original, levels 5 / 4 / 3 (decoder first convs 512 -> 512, 512 -> 256, 256 -> 128): W' = W M as an fp64 matrix-core
    GEMM (solve.hip fold_gemm_kernel) against misc.hip's fold_block_kernel (debug key "foldgemm" 0) -- the same fp64 sums in another
    order, rounded to fp32 once: level outputs agree to fp32 round-off, alpha = 1 and the alpha blend; levels 2 / 1 (cin < 256) do
    not take the GEMM and agree bitwise."""
    from wct_hip import WCT, model_zoo
    torch = torch_cuda
    wo = model_zoo.synth_weights("original", 7)
    gen = torch.Generator(device="cuda").manual_seed(33)
    c = torch.rand((1, 3, 176, 208), device="cuda", generator=gen)
    s = torch.rand((1, 3, 160, 144), device="cuda", generator=gen)
    outs = {}
    for fg in (1, 0):
        w = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=wo)
        w.debug_set("foldgemm", fg)
        outs[fg] = [w.style_transfer_level(k, c, s, alpha).clone() for k in (5, 4, 3, 2, 1) for alpha in (1.0, 0.6)]
    for k, (a, b) in enumerate(zip(outs[1], outs[0])):
        if k >= 6:
            assert torch.equal(a, b), k
        else:
            assert float((a - b).abs().max() / b.abs().max()) < 2e-6, k


def test_fused_ends_match_unfused(torch_cuda, weights16x):
    """Fused ends vs the layer-by-layer path (odd sizes, image-border tiles).
    Decoder tail (conv12+conv11): conv12 has the unfused kernel's arithmetic; the final 16 -> 3 conv runs block-packed in the
    fused kernel (pairs of pixels per MFMA column, another summation order) -> fp32 round-off agreement.
    Encoder head (conv11+conv12+pool): conv11 runs as f16x3 there and as exact-fp32 MFMA unfused -> fp32-class
    agreement (the tolerance is relative to max|y|, as in the golden tests)."""
    from wct_hip import WCT
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(21)
    c = torch.rand((1, 3, 203, 333), device="cuda", generator=g)
    f5 = torch.rand((1, 128, 12, 20), device="cuda", generator=g)
    f2 = torch.rand((1, 32, 101, 166), device="cuda", generator=g)
    outs = {}
    for fuse in ("1", "0"):
        w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        w.debug_set("fuse", int(fuse))
        outs[fuse] = ([w.e5(c).clone(), w.e2(c).clone(), w.e3(c[..., :64, :32]).clone()], [w.d5(f5).clone(), w.d2(f2).clone()])
    for a, b in zip(outs["1"][0], outs["0"][0]):
        assert float((a - b).abs().max() / b.abs().max()) < 3e-6
    for a, b in zip(outs["1"][1], outs["0"][1]):
        assert a.shape == b.shape and float((a - b).abs().max() / b.abs().max()) < 3e-6


def test_head_with_two_roles_is_bitwise_the_single_role_head(torch_cuda, tmp_path):
    """Large images take the fused head in its two-role form (conv3x3_f16.hip enc_head_roles_kernel: producer waves run conv11 of
    tile t + 1 while consumer waves run conv12 + pool of tile t, double-buffered through LDS); WCT_HEAD_ROLES=0 (under WCT_DEBUG)
    selects enc_head_kernel<24>.  Same device functions per pixel -> the encoders' outputs must agree BIT FOR BIT: sizes with
    ragged right / bottom tiles and image-border (reflecting) tiles, SP16 and fp32 outputs behind the head (levels 5 and 2)."""
    import hashlib
    import os
    import subprocess
    import sys
    from tests.conftest import PKG, REPO
    code = (
        "import sys, types, hashlib, torch\n"
        "sys.path[:0] = [%r, %r]\n"
        "from wct_hip import WCT, model_zoo\n"
        "import os\n"
        "wct = WCT(types.SimpleNamespace(mode='16x', alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(%r, 'weights', '16x.npz')))\n"
        "g = torch.Generator(device='cuda').manual_seed(3)\n"
        "for (H, W) in ((1100, 1950), (1030, 4100), (2160, 3840)):\n"
        "    c = torch.rand((1, 3, H, W), device='cuda', generator=g)\n"
        "    for L in (5, 2):\n"
        "        y = wct.encode(L, c)\n"
        "        assert bool(torch.isfinite(y).all())\n"
        "        print(H, W, L, hashlib.sha256(y.cpu().numpy().tobytes()).hexdigest())\n" % (REPO, PKG, PKG))
    outs = []
    for env in ({}, {"WCT_DEBUG": "1", "WCT_HEAD_ROLES": "0"}):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-3000:]
        outs.append([ln for ln in r.stdout.splitlines() if ln and ln[0].isdigit()])
    assert len(outs[0]) == 6 and outs[0] == outs[1]


def test_sp16_dma_path_bitwise_equals_fp32_activations(torch_cuda, weights16x):
    """SP16 intermediates + DMA-staged kernels (default) vs fp32 NHWC intermediates + register-staged kernels
    (debug_set("sp", 0)): the split hi/lo values and every accumulation order are the same, so whole encoders / decoders agree bit
    for bit -- on odd sizes (partial tiles, odd pooling, image-border reflection) and in original mode (cout groups)."""
    from wct_hip import WCT
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(33)
    c = torch.rand((1, 3, 211, 173), device="cuda", generator=g)
    f5 = torch.rand((1, 128, 13, 11), device="cuda", generator=g)
    f3 = torch.rand((1, 64, 37, 45), device="cuda", generator=g)
    co = torch.rand((1, 3, 70, 90), device="cuda", generator=g)
    wo = model_zoo.synth_weights("original", 7)
    res = {}
    for sp in ("1", "0"):
        w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        w.debug_set("sp", int(sp))
        w.debug_set("upconv", 0)     # the nine-tap form behind the upsamples on both sides (the 2x2 form exists only for SP16 input)
        r = [w.e5(c).clone(), w.e4(c).clone(), w.e3(c).clone(), w.e2(c).clone(), w.d5(f5).clone(), w.d3(f3).clone()]
        w2 = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=wo)
        w2.debug_set("sp", int(sp))
        w2.debug_set("upconv", 0)
        r += [w2.e4(co).clone(), w2.d4(w2.e4(co)).clone()]
        res[sp] = r
    for a, b in zip(res["1"], res["0"]):
        assert a.shape == b.shape and torch.equal(a, b)


def test_upsample_layers_on_the_low_resolution_grid(torch_cuda, weights16x, oracle):
    """Decoder layers behind a nearest-x2 upsample run as per-parity 2x2 convolutions of the low-resolution map with summed taps
    (dec_tail_up_kernel, conv3x3_sp_up_kernel; debug_set("upconv", 0) = the nine-tap form on the gathered upsampled halo): the same
    operator, 4/9 of the products -- fp32 round-off agreement on odd sizes (clamped borders = reflect padding of the upsampled map,
    partial tiles, leftover columns of the parity groups) in both modes, and against the CPU checker."""
    from wct_hip import WCT
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(41)
    feats = {5: [(128, 2, 2), (128, 7, 11), (128, 33, 50)], 4: [(128, 9, 5), (128, 40, 66)], 3: [(64, 17, 130), (64, 68, 49)], 2: [(32, 3, 3), (32, 133, 77)]}
    on = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
    off = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
    off.debug_set("upconv", 0)
    mods = oracle.Modules("16x", weights16x)
    for L, shapes in feats.items():
        for shp in shapes:
            f = torch.rand((1,) + shp, device="cuda", generator=g)
            a, b = getattr(on, "d%d" % L)(f).clone(), getattr(off, "d%d" % L)(f).clone()
            assert a.shape == b.shape and float((a - b).abs().max() / b.abs().max()) < 3e-6, (L, shp)
            if shp[1] * shp[2] <= 2000:
                ref = mods.decode(L, f[0].cpu().numpy())
                assert rel_err(a[0].cpu().numpy(), ref) < 2e-5, (L, shp)
    wo = model_zoo.synth_weights("original", 7)
    on2 = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=wo)
    off2 = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=wo)
    off2.debug_set("upconv", 0)
    f = torch.rand((1, 512, 9, 13), device="cuda", generator=g)
    a, b = on2.d4(f).clone(), off2.d4(f).clone()
    assert float((a - b).abs().max() / b.abs().max()) < 5e-6


def test_level1_fused_matches_layerwise(torch_cuda, weights16x):
    """Level 1 without relu1_1 in HBM (level1.hip: image -> conv11 -> moments, image -> conv11 -> folded conv -> image)
    vs the layer-by-layer path (fp32-MFMA conv11, moments kernel, c16 decoder conv).  conv11 is f16x3 in the fused
    kernels and exact-fp32 MFMA layer-wise: fp32-class agreement, tolerance relative to max|y| as in the golden tests.
    Odd sizes put image-border tiles, partial tiles and the reflected halo ring on the path."""
    from wct_hip import WCT
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(5)
    c = torch.rand((1, 3, 203, 333), device="cuda", generator=g)
    s = torch.rand((1, 3, 97, 131), device="cuda", generator=g)
    res = {}
    for fuse in ("1", "0"):
        w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        w.debug_set("l1fuse", int(fuse))
        h, wd, sm, sq = w.content_encode(1, c, 16, 240)          # windowed moments (the sharded path's call)
        res[fuse] = (w.e1(c).clone(), sm.clone(), sq.clone(), w.style_transfer_level(1, c, s).clone(),
                     w.style_transfer_level(1, c, s, 0.6).clone())
    a, b = res["1"], res["0"]
    assert float((a[0] - b[0]).abs().max() / b[0].abs().max()) < 3e-6          # relu1_1 (API)
    assert float((a[1] - b[1]).abs().max() / b[1].abs().max()) < 1e-6          # sum over the window
    assert float((a[2] - b[2]).abs().max() / b[2].abs().max()) < 1e-6          # sum of products over the window
    for k in (3, 4):
        assert float((a[k] - b[k]).abs().max() / b[k].abs().max()) < 2e-5, k  # through the whitening (cond ~1e3)


# --------------------------------------------------------------------------- image edge, style cache
def test_g9_image_edge_bit_exact(torch_cuda, wct16, oracle, golden):
    """ToTensor / save_image conversions on the device: bit-exact against the golden and, on odd sizes that exercise
    the 4-pixel tails and unaligned plane starts, against the oracle."""
    torch = torch_cuda
    g = golden("g9_image_edge.npz")
    assert np.array_equal(wct16.to_tensor_u8(torch.from_numpy(g["u8"])).cpu().numpy()[0], g["to_tensor"])
    assert np.array_equal(wct16.to_u8(torch.from_numpy(g["f32"]).cuda(), 0).cpu().numpy(), g["save_trunc"])
    assert np.array_equal(wct16.to_u8(torch.from_numpy(g["f32"]).cuda(), 1).cpu().numpy(), g["save_round"])
    rng = np.random.default_rng(3)
    for (h, w) in ((1, 1), (3, 5), (7, 13), (64, 67), (251, 509)):
        u8 = rng.integers(0, 256, size=(h, w, 3), dtype=np.uint8)
        assert np.array_equal(wct16.to_tensor_u8(torch.from_numpy(u8)).cpu().numpy()[0], oracle.to_tensor_u8(u8))
        f = (rng.random((3, h, w), dtype=np.float32) * 1.4 - 0.2).astype(np.float32)
        for mode in (0, 1):
            assert np.array_equal(wct16.to_u8(torch.from_numpy(f).cuda(), mode).cpu().numpy(), oracle.to_u8(f, mode))


def test_stylize_u8_and_prepared_style(torch_cuda, wct16, oracle):
    """stylize_u8 == to_u8(stylize(to_tensor(.))) bitwise; the prepared-style cascade and export -> import of the style
    statistics reproduce stylize() bitwise (same kernels, same statistics)."""
    torch = torch_cuda
    rng = np.random.default_rng(11)
    cu8 = torch.from_numpy(rng.integers(0, 256, size=(96, 112, 3), dtype=np.uint8)).cuda()
    su8 = torch.from_numpy(rng.integers(0, 256, size=(80, 72, 3), dtype=np.uint8)).cuda()
    c, s = wct16.to_tensor_u8(cu8), wct16.to_tensor_u8(su8)
    ref = wct16.stylize(c, s).clone()
    assert torch.equal(wct16.stylize_u8(cu8, su8), wct16.to_u8(ref))
    wct16.style_prepare(s)
    assert torch.equal(wct16.stylize_prepared(c), ref)
    stats = {L: wct16.style_export(L).clone() for L in (5, 4, 3, 2, 1)}
    wct16.style_prepare(torch.rand_like(s))          # overwrite the statistics with another style's ...
    assert not torch.equal(wct16.stylize_prepared(c), ref)
    for L, v in stats.items():                       # ... and bring the first style back through import
        wct16.style_import(L, v)
    assert torch.equal(wct16.stylize_prepared(c), ref)
    wct16.style_prepare(s, levels=(5, 3))            # partial prepare (what a rank of a sharded run does)
    assert torch.equal(wct16.stylize_prepared(c), ref)


def test_g10_numpy_variant(torch_cuda, weights16x, golden):
    """`--numpy` semantics on the device (+ I on the content covariance, util_wct.py:143) against the reference's output."""
    from wct_hip import WCT
    g = golden("g10_numpy_variant.npz")
    w = WCT(types.SimpleNamespace(mode="16x", alpha=1.0, numpy=True), weights=weights16x)
    for tag in ("c64", "c24"):
        cF, sF = cu(torch_cuda, g[tag + ".cF"]), cu(torch_cuda, g[tag + ".sF"])
        y = w.transform(cF, sF, None, float(g[tag + ".alpha"])).cpu().numpy()
        assert rel_err(y, g[tag + ".csF"]) < 2e-6, tag


def test_mode_16x_kd2sd_uses_16x_graphs(torch_cuda, wct16, weights16x):
    """--mode 16x_kd2sd (WCT.py:60-70, model/model_kd2sd.py): the 16x encoders with decoders of the same conv graph (their aux
    1x1 heads are never called in forward); its checkpoints are absent, so the plumbing is checked with the 16x tensors."""
    from wct_hip import WCT
    torch = torch_cuda
    w = WCT(types.SimpleNamespace(mode="16x_kd2sd", alpha=1.0), weights=weights16x)
    g = torch.Generator(device="cuda").manual_seed(2)
    c, s = torch.rand((1, 3, 64, 80), device="cuda", generator=g), torch.rand((1, 3, 48, 48), device="cuda", generator=g)
    assert torch.equal(w.stylize(c, s), wct16.stylize(c, s))
    with pytest.raises(FileNotFoundError):
        WCT(types.SimpleNamespace(mode="16x_kd2sd", alpha=1.0))


def test_wide_model_deferred_solves_fall_back(torch_cuda, tmp_path):
    """--mode original: the C > 128 matrix functions no longer synchronise the host once per solve; the cascade-type calls enqueue
    everything, synchronise once, look at the iterations' outcomes and repeat the call the synchronous way if one failed
    (wct_api.hip with_deferred_solves).  Forced here: with an iteration budget of 3 (WCT_NS_MAXIT, honoured under WCT_DEBUG) no
    deflated iteration can converge, so every call must come back through the synchronous path and its global-memory Jacobi net --
    and agree with the normal run to the solvers' agreement (1e-5 of the output range; level 2, C = 128, takes the deflated path
    in a wide model too).  A fresh process per setting: the budget is read once."""
    import os
    import subprocess
    import sys
    from tests.conftest import PKG, REPO
    code = (
        "import sys, types, numpy as np, torch\n"
        "sys.path[:0] = [%r, %r]\n"
        "from wct_hip import WCT, model_zoo\n"
        "w = WCT(types.SimpleNamespace(mode='original', alpha=1.0), weights=model_zoo.synth_weights('original', 2024))\n"
        "g = torch.Generator(device='cuda').manual_seed(5)\n"
        "c, s = torch.rand((1, 3, 64, 80), device='cuda', generator=g), torch.rand((1, 3, 48, 64), device='cuda', generator=g)\n"
        "outs = [w.style_transfer_level(k, c, s).cpu().numpy() for k in (5, 3, 2)] + [w.stylize(c, s).cpu().numpy()]\n"
        "w.style_prepare(s); outs.append(w.stylize_prepared(c).cpu().numpy())\n"
        "img = c\n"
        "for k in (5, 4, 3, 2, 1): img = w.style_transfer_level(k, img, s)\n"
        "outs.append(img.cpu().numpy())\n"
        "assert all(np.isfinite(o).all() for o in outs) and w.saturation_count() == 0\n"
        "np.savez(sys.argv[1], *outs)\n" % (REPO, PKG))
    res = {}
    for tag, env in (("normal", {}), ("forced", {"WCT_DEBUG": "1", "WCT_NS_MAXIT": "3"})):
        out = str(tmp_path / (tag + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, out], env=dict(os.environ, **env), capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        with np.load(out) as z:
            res[tag] = [z[k] for k in z.files]
    for a, b in zip(res["forced"][:3], res["normal"][:3]):
        assert a.shape == b.shape and rel_err(a, b) < 1e-5
    for tag in ("normal", "forced"):     # the cascade in one call == against prepared style statistics == its levels chained, bit for bit,
        assert np.array_equal(res[tag][3], res[tag][4]) and np.array_equal(res[tag][3], res[tag][5]), tag   # on either solver path
    # (five chained levels of random 512-channel stacks on 4x5 .. 64x80-pixel maps are chaotic: the two solver paths' 1e-5 per level
    # does not survive the chain, so the cross-path comparison stays level-isolated)


def test_replica_stylizer_single_rank(torch_cuda, wct16):
    """wct_hip/replicas.py with one rank is stylize(); the multi-rank exchange is covered on CPU (gloo, test_sharded_gloo.py)."""
    from wct_hip.replicas import ReplicaStylizer
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(4)
    c, s = torch.rand((1, 3, 80, 96), device="cuda", generator=g), torch.rand((1, 3, 64, 64), device="cuda", generator=g)
    ref = wct16.stylize(c, s).clone()
    assert torch.equal(ReplicaStylizer(wct16, None).stylize(c, s), ref)


# --------------------------------------------------------------------------- G6 original arch, G7 config 1
def test_g6_original_arch(torch_cuda, golden):
    """--mode original graph (C = 512/512/256/128/64: multi-group conv launches, global-memory Jacobi)."""
    from wct_hip import WCT
    g = golden("g6_original.npz")
    w = model_zoo.synth_weights("original", int(g["seed"]))
    wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
    style = cu(torch_cuda, g["style"])[None]
    for k in (5, 4, 3, 2, 1):
        c = cu(torch_cuda, g["L%d.content" % k])[None]
        assert rel_err(getattr(wct, "e%d" % k)(c).cpu().numpy(), g["e%d.y" % k]) < 2e-5
        y = wct.style_transfer_level(k, c, style).cpu().numpy()[0]
        assert rel_err(y, g["L%d.out" % k]) < 5e-4, k


def test_original_first_conv_forms(torch_cuda):
    """The 3 -> 64 first conv of the un-pruned encoders in its three forms (debug key in3wide): 2 = exact-fp32 MFMA with four cout tiles per
    operand read (level1.hip in3_wide_f32_kernel, the default), 0 = the generic exact-fp32 kernel -- the same products in the same order,
    so bit-identical, fp32 NHWC out (level 1) and SP16 out (level 2), odd sizes and reflected borders included -- and 1 = f16x3 with K = 27,
    which agrees to the split's accuracy (DESIGN 2 explains why it is no longer the default)."""
    from wct_hip import WCT
    torch = torch_cuda
    w = model_zoo.synth_weights_conditioned("original", 15)
    wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
    g = torch.Generator(device="cuda").manual_seed(21)
    for (H, W) in ((200, 264), (45, 37), (2, 3), (301, 1000)):
        c = torch.rand((1, 3, H, W), device="cuda", generator=g)
        outs = {}
        for key in (2, 0, 1):
            wct.debug_set("in3wide", key)
            outs[key] = [wct.encode(1, c).clone()] + ([wct.encode(2, c).clone()] if H >= 4 and W >= 4 else [])
        for a, b in zip(outs[2], outs[0]):
            assert torch.equal(a, b), (H, W)
        for a, b in zip(outs[2], outs[1]):
            assert float((a - b).abs().max() / a.abs().max()) < 2e-5, (H, W)
    wct.debug_set("in3wide", 2)
    wct.sync()


def test_g8_constant_content(torch_cuda, wct16, golden):
    g = golden("g8_constant.npz")
    c, s = cu(torch_cuda, g["content"])[None], cu(torch_cuda, g["style"])[None]
    for k in (3, 1):
        y = wct16.style_transfer_level(k, c, s).cpu().numpy()[0]
        assert np.isfinite(y).all()
        assert rel_err(y, g["L%d.out" % k]) < 1e-5, k


def test_g7_config1(torch_cuda, wct16, golden):
    g = golden("g7_config1.npz")
    r0 = np.random.default_rng(0)
    c = r0.random((1, 3, 512, 512), dtype=np.float32)
    s = r0.random((1, 3, 512, 512), dtype=np.float32)
    out = wct16.style_transfer_level(1, cu(torch_cuda, c), cu(torch_cuda, s)).cpu().numpy()[0]
    assert rel_err(out[:, 200:264, 300:364], g["crop"]) < 1e-4
    assert abs(out.mean(dtype=np.float64) - float(g["mean"])) < 1e-5
    assert abs(float(out.max()) - float(g["max"])) < 1e-3


# --------------------------------------------------------------------------- oracle at moderate sizes
def smooth(rng, shape, it=3):
    x = rng.random(shape, dtype=np.float32)
    for _ in range(it):
        x = (x + np.roll(x, 1, 1) + np.roll(x, 1, 2) + np.roll(x, -1, 1) + np.roll(x, -1, 2)) / 5
    return np.ascontiguousarray((x - x.min()) / (x.max() - x.min()))


@pytest.mark.parametrize("H,W,Hs,Ws", [(250, 333, 200, 160), (512, 768, 384, 384)])
def test_levels_vs_oracle(torch_cuda, wct16, oracle, weights16x, H, W, Hs, Ws):
    """Tile-boundary shapes (not multiples of 16 or of the 2^4 pooling pyramid), uniform-noise style,
    smooth content (dead channels, ill-conditioned covariance)."""
    rng = np.random.default_rng(H * 7 + W)
    c, s = smooth(rng, (3, H, W)), rng.random((3, Hs, Ws), dtype=np.float32)
    mods = oracle.Modules("16x", weights16x)
    img = c
    for k in (5, 4, 3, 2, 1):
        ref = oracle.style_transfer(mods, k, img, s, 1.0)
        got = wct16.style_transfer_level(k, cu(torch_cuda, img)[None], cu(torch_cuda, s)[None]).cpu().numpy()[0]
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 2e-4, k
        img = ref


@pytest.mark.parametrize("mode,H,W,Hs,Ws,alpha", [("16x", 42, 410, 604, 229, 1.0), ("16x", 343, 748, 195, 56, 0.6), ("16x", 166, 581, 257, 430, 0.6),
                                                   ("original", 97, 47, 34, 217, 1.0), ("original", 118, 158, 146, 111, 0.6)])
def test_odd_shapes_vs_oracle(torch_cuda, oracle, weights16x, mode, H, W, Hs, Ws, alpha):
    """Shapes drawn by tools/experiments/fuzz_sizes.py (thin strips, fewer tiles than XCDs, feature maps of a few pixels under
    512 channels -> singular covariances through the deflated iteration), every level on the checker's own level input."""
    from wct_hip import WCT
    w = weights16x if mode == "16x" else model_zoo.synth_weights("original", 7)
    wct = WCT(types.SimpleNamespace(mode=mode, alpha=alpha), weights=w)
    mods = oracle.Modules(mode, w)
    rng = np.random.default_rng(H * 1000 + W)
    img, s = rng.random((3, H, W), dtype=np.float32), rng.random((3, Hs, Ws), dtype=np.float32)
    for k in (5, 4, 3, 2, 1):
        ref = oracle.style_transfer(mods, k, img, s, alpha)
        got = wct.style_transfer_level(k, cu(torch_cuda, img)[None], cu(torch_cuda, s)[None]).cpu().numpy()[0]
        assert got.shape == ref.shape
        assert rel_err(got, ref) < 2e-4, k
        img = ref


# --------------------------------------------------------------------------- properties at full size
def test_full_size_properties_config2(torch_cuda, wct16):
    """BASELINE config 2 (3840x2160 content, 2048x2048 style): the oracle would take minutes, so check
    size-independent properties of the HIP path itself:
      * after the transform the content feature has the style's mean and covariance (alpha = 1);
      * alpha = 0 is the identity on features;
      * the cascade is deterministic (bitwise) and finite;
      * a 16-aligned crop-free translation: stylising is equivariant to nothing global, so instead check
        that moments over column windows add up (what the sharded path relies on)."""
    torch = torch_cuda
    g = torch.Generator(device="cuda").manual_seed(1)
    c = torch.rand((1, 3, 2160, 3840), device="cuda", generator=g)
    s = torch.rand((1, 3, 2048, 2048), device="cuda", generator=g)
    for k in (4, 1):
        cF = wct16.encode(k, c, layout="nhwc")
        sF = wct16.encode(k, s, layout="nhwc")
        nc, sc, ssc = wct16.moments(cF)
        ns, ss_, sss = wct16.moments(sF)
        w = cF.shape[2]
        wins = ((0, w // 3), (w // 3, w - 5), (w - 5, w))
        parts = [wct16.moments(cF, a, b) for a, b in wins]
        assert abs(sum(p[0] for p in parts) - nc) == 0
        # maps of this size take fp32 block products (moments.hip F32 variant): windows cut the 64-pixel blocks differently, so the
        # parts add up to the whole to the blocks' fp32 rounding averaged over >= 1000 blocks, not to fp64 round-off ...
        assert rel_err(sum(p[1] for p in parts).cpu().numpy(), sc.cpu().numpy()) < 1e-7
        assert rel_err(sum(p[2] for p in parts).cpu().numpy(), ssc.cpu().numpy()) < 1e-7
        # ... and in the fp64 form (debug key mom32 = 0) they add up exactly as before
        wct16.debug_set("mom32", 0)
        n64, s64, q64 = wct16.moments(cF)
        parts64 = [wct16.moments(cF, a, b) for a, b in wins]
        wct16.debug_set("mom32", 1)
        assert rel_err(sum(p[1] for p in parts64).cpu().numpy(), s64.cpu().numpy()) < 1e-13
        assert rel_err(sum(p[2] for p in parts64).cpu().numpy(), q64.cpu().numpy()) < 1e-13
        assert rel_err(sc.cpu().numpy(), s64.cpu().numpy()) < 1e-7 and rel_err(ssc.cpu().numpy(), q64.cpu().numpy()) < 1e-7
        M, b = wct16.solve(nc, sc, ssc, ns, ss_, sss, alpha=1.0)
        C = cF.shape[3]
        out = torch.empty_like(cF)
        from wct_hip import lib
        wct16._stream()
        wct16._chk(wct16._lib.wct_apply(wct16._ctx, cF.data_ptr(), C, cF.shape[1], cF.shape[2], lib.LAYOUT_NHWC,
                                        M.data_ptr(), b.data_ptr(), out.data_ptr()))
        no, so, sso = wct16.moments(out)
        mu_o, mu_s = (so / no).cpu().numpy(), (ss_ / ns).cpu().numpy()
        cov_o = ((sso - no * torch.outer(so / no, so / no)) / (no - 1)).cpu().numpy()
        cov_s = ((sss - ns * torch.outer(ss_ / ns, ss_ / ns)) / (ns - 1)).cpu().numpy()
        cov_c = ((ssc - nc * torch.outer(sc / nc, sc / nc)) / (nc - 1)).cpu().numpy()
        assert rel_err(mu_o, mu_s) < 1e-5
        # colouring reproduces the style covariance on the content's live subspace; with a full-rank
        # content covariance that is all of it
        if np.linalg.matrix_rank(cov_c, tol=1e-8 * np.abs(cov_c).max()) == C:
            assert rel_err(cov_o, cov_s) < 1e-4
        M0, b0 = wct16.solve(nc, sc, ssc, ns, ss_, sss, alpha=0.0)
        assert rel_err(M0.cpu().numpy(), np.eye(C)) < 1e-12 and float(b0.abs().max()) < 1e-9
    a = wct16.stylize(c, s).clone()
    b_ = wct16.stylize(c, s)
    assert tuple(a.shape) == (1, 3, 2160, 3840)
    assert bool(torch.isfinite(a).all()) and torch.equal(a, b_)
    assert float(a.min()) >= 0.0  # every decoder ends in a ReLU (model_cd.py:293)


# --------------------------------------------------------------------------- error behaviour
def test_errors(torch_cuda, wct16):
    torch = torch_cuda
    with pytest.raises(ValueError):
        wct16.e5(torch.zeros(2, 3, 64, 64, device="cuda"))          # batch size 1 only (WCT.py:92)
    with pytest.raises(ValueError):
        wct16.e5(torch.zeros(1, 3, 16, 16, device="cuda"))          # 1x1 at relu5_1: reflect pad impossible
    with pytest.raises(ValueError):
        wct16.d5(torch.zeros(1, 64, 8, 8, device="cuda"))           # wrong channel count
    with pytest.raises(ValueError):
        wct16.transform(torch.zeros(24, 4, 4), torch.zeros(32, 4, 4))


# --------------------------------------------------------------------------- sharding logic on the real kernels (ranks: tests/test_sharded_gpu.py)
def test_strip_halos_exact_per_level(torch_cuda, wct16):
    """Sharding logic on the real kernels, level-isolated: with the SAME (M, b), a strip computed from
    [own - A_L, own + A_L) is BITWISE equal to the untiled level on own +- (A_L - halo_L), for edge strips and
    an interior strip, including a width that floor-pooling shrinks (2005 -> 2000)."""
    from wct_hip.sharded import CUM_HALO, LEVEL_HALO, ext_bounds, strip_bounds
    torch = torch_cuda
    H, W, world = 176, 2005, 3
    g = torch.Generator(device="cuda").manual_seed(3)
    content = torch.rand((1, 3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 200, 180), device="cuda", generator=g)
    for L in (5, 4, 3, 2, 1):
        sh = L - 1
        sF = wct16.encode(L, style, layout="nhwc")
        cF = wct16.encode(L, content, layout="nhwc")
        nc, sc, ssc = wct16.moments(cF)
        M, b = wct16.solve(nc, sc, ssc, *wct16.moments(sF))
        full = wct16.decode_affine(L, cF, M, b)
        Wn = (W >> sh) << sh
        assert full.shape[-1] == Wn
        tot = 0
        for own in strip_bounds(W, world):
            lo, hi = ext_bounds(own, W, CUM_HALO[L])
            f = wct16.encode(L, content[..., lo:hi].contiguous(), layout="nhwc")
            f0 = (own[0] - lo) >> sh
            f1 = f.shape[2] if own[1] >= W else (own[1] - lo) >> sh
            assert torch.equal(f[0, :, f0:f1], cF[0, :, own[0] >> sh:(own[0] >> sh) + f1 - f0])
            n, s1, s2 = wct16.moments(f, f0, f1)
            tot += n
            o = wct16.decode_affine(L, f, M, b)
            v = CUM_HALO[L] - LEVEL_HALO[L]
            a, e = max(0, own[0] - v), min(Wn, own[1] + v)
            assert torch.equal(o[..., a - lo:e - lo], full[..., a:e]), (L, own)
        assert tot == nc


def test_original_mode_from_t7_checkpoints(torch_cuda, tmp_path):
    """`--mode original` as the reference starts it: ten torch7 files in args.e1..d5 (WCT.py:36-46) -> the same device state as
    handing the arrays over directly."""
    from tests.test_t7 import write_module
    from wct_hip import WCT
    w = model_zoo.synth_weights("original", 5)
    args = types.SimpleNamespace(mode="original", alpha=0.7)
    for k in range(1, 6):
        for kind, key in (("enc", "e%d" % k), ("dec", "d%d" % k)):
            path = str(tmp_path / (key + ".t7"))
            write_module(path, kind, k, w, key, mm=(k == 2))
            setattr(args, key, path)
    a, b = WCT(args), WCT(types.SimpleNamespace(mode="original", alpha=0.7), weights=w)
    g = torch_cuda.Generator(device="cuda").manual_seed(1)
    c, s = torch_cuda.rand((1, 3, 96, 80), device="cuda", generator=g), torch_cuda.rand((1, 3, 64, 112), device="cuda", generator=g)
    ya, yb = a.stylize(c, s), b.stylize(c, s)
    assert torch_cuda.isfinite(ya).all() and torch_cuda.equal(ya, yb)


def test_frame_pipeline_equals_single_engine(torch_cuda, wct16, weights16x):
    """wct_hip/pipeline.py: three frames of different sizes in flight on two engines = the single-engine results, bit for bit."""
    from wct_hip import WCT
    from wct_hip.pipeline import FramePipeline
    g = torch_cuda.Generator(device="cuda").manual_seed(9)
    style = torch_cuda.rand((3, 200, 312), device="cuda", generator=g)
    frames = [torch_cuda.rand((3, h, w), device="cuda", generator=g) for h, w in ((160, 240), (96, 400), (333, 250), (160, 240))]
    pipe = FramePipeline(lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x), slots=2)
    got = pipe.stylize_many(frames, style=style)
    torch_cuda.cuda.synchronize()
    wct16.style_prepare(style)
    for f, y in zip(frames, got):
        ref = wct16.stylize_prepared(f)
        assert y.shape == ref.shape and torch_cuda.equal(y, ref)
    with pytest.raises(RuntimeError):
        FramePipeline(lambda: wct16, slots=1).stylize_many(frames[:1])


def test_stylize_16x_is_capturable_into_a_hip_graph(torch_cuda, weights16x):
    """include/wct_hip.h: "the 16x path never synchronises" -- so the whole 5-level cascade (both lanes: the side stream forks from and
    rejoins the caller's stream through events) must be capturable into ONE HIP graph.  Captured once
    after a warm-up call (workspaces sized: no allocation in the capture), replayed on the SAME buffers with a different content: bitwise
    the direct call's result both times.  In a fresh process: a failed capture can leave the runtime in capture mode."""
    import os
    import subprocess
    import sys
    from tests.conftest import PKG, REPO
    code = r"""
import sys, types
sys.path[:0] = [%r, %r]
import torch
from wct_hip import WCT, model_zoo
import os
w = model_zoo.load_npz_weights(os.path.join(%r, "weights", "16x.npz"))
wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
g = torch.Generator(device="cuda").manual_seed(5)
c1 = torch.rand((3, 272, 400), device="cuda", generator=g)
c2 = torch.rand((3, 272, 400), device="cuda", generator=g)
s = torch.rand((3, 200, 240), device="cuda", generator=g)
want1 = wct.stylize(c1, s).clone()
want2 = wct.stylize(c2, s).clone()
c = c1.clone()
out = torch.empty((3, 272, 400), device="cuda")
wct.stylize(c, s, out=out)                       # warm-up on the buffers the graph will use
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    wct.stylize(c, s, out=out)
out.zero_()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(out.view(1, 3, 272, 400), want1), "replay 1 differs"
c.copy_(c2)
graph.replay()
torch.cuda.synchronize()
assert torch.equal(out.view(1, 3, 272, 400), want2), "replay 2 (new content, same graph) differs"
assert wct.saturation_count() == 0
print("GRAPH_OK")
""" % (REPO, PKG, PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "GRAPH_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
