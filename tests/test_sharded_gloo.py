"""The N > 1 path (wct_hip/sharded.py) under torch.distributed `gloo` on CPU, world_size 2 and 3.

The orchestration under test is the product's; the arithmetic behind it is supplied by the CPU checker
(oracle/) through the same engine interface the HIP library implements (encode / moments / solve /
decode_affine), so what is verified here is the sharding logic: strip bounds, cumulative halos, windowed
moments + all-reduce, broadcast of (M, b), crops, floor-mode shrinking.  Expected result: the gathered strips
equal the UNTILED cascade."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests.conftest import PKG, REPO


class OracleEngine:
    """CPU stand-in with the engine interface wct_hip.sharded expects of wct_hip.WCT (test infrastructure only)."""

    def __init__(self, weights):
        from oracle import wct_oracle
        self.o = wct_oracle
        self.m = wct_oracle.Modules("16x", weights)
        self.style_stats = {}
        self.cF = None

    @staticmethod
    def _raw(f, x0=0, x1=None):
        f = f.astype(np.float64)
        x1 = f.shape[2] if x1 is None or x1 < 0 else x1
        X = f[:, :, x0:x1].reshape(f.shape[0], -1)
        return float(X.shape[1]), X.sum(1), X @ X.T

    def style_prepare(self, style, levels=(5, 4, 3, 2, 1)):
        s = (style[0] if style.dim() == 4 else style).numpy()
        for L in levels:
            self.style_stats[L] = self._raw(self.m.encode(L, s))

    # style_mode "strips": raw moments of a STRIP of the style over its owned feature columns, then the all-reduced totals come back
    def style_moments(self, level, strip, x0=0, x1=-1):
        s = (strip[0] if strip.dim() == 4 else strip).numpy()
        _, sm, ss = self._raw(self.m.encode(level, np.ascontiguousarray(s)), x0, x1)
        return torch.from_numpy(sm), torch.from_numpy(ss)

    def style_solve(self, level, n_s, sum_s, sumsq_s):
        self.style_stats[level] = (float(n_s), sum_s.numpy().copy(), sumsq_s.numpy().copy())

    # the level's style statistics as one fp64 vector (this checker ships raw moments: n | sum[C] | sumsq[C*C])
    def style_stats_count(self, level):
        from wct_hip import model_zoo
        C = model_zoo.feature_channels("16x", level)
        return 1 + C + C * C

    def style_export(self, level):
        n, s, ss = self.style_stats[level]
        return torch.from_numpy(np.concatenate([[n], s, ss.reshape(-1)]).astype(np.float64))

    def style_import(self, level, stats):
        v = stats.numpy()
        C = int(round((-1 + (1 + 4 * (v.size - 1)) ** 0.5) / 2))
        self.style_stats[level] = (float(v[0]), v[1:1 + C].copy(), v[1 + C:].reshape(C, C).copy())

    def content_encode(self, level, img, x0=0, x1=-1):
        x = (img[0] if img.dim() == 4 else img).numpy()
        self.cF = self.m.encode(level, x)
        n, s, ss = self._raw(self.cF, x0, x1)
        return self.cF.shape[1], self.cF.shape[2], torch.from_numpy(s), torch.from_numpy(ss)

    def content_solve(self, level, n_c, sum_c, sumsq_c, alpha=1.0):
        def mc(n, s, ss):
            mu = s / n
            return mu, (ss - n * np.outer(mu, mu)) / (n - 1)
        mu_c, cov_c = mc(n_c, sum_c.numpy(), sumsq_c.numpy())
        n_s, s_s, ss_s = self.style_stats[level]
        mu_s, cov_s = mc(n_s, s_s, ss_s)
        M, b = self.o.affine_from_moments(mu_c, cov_c, mu_s, cov_s, alpha)
        return torch.from_numpy(M), torch.from_numpy(b)

    def stylize_prepared(self, content, alpha=1.0, num_run=1):
        """The cascade against the style moments held by the engine (what wct_hip.WCT.stylize_prepared does on the GPU)."""
        img = (content[0] if content.dim() == 4 else content)
        for _ in range(num_run):
            for L in (5, 4, 3, 2, 1):
                h, w, s, ss = self.content_encode(L, img)
                M, b = self.content_solve(L, float(h * w), s, ss, alpha)
                img = self.content_decode(L, M, b, 0, 0)[0]
        return img[None]

    def content_decode(self, level, M, b, H, W):
        f = self.cF.astype(np.float64)
        y = (np.einsum("ab,bhw->ahw", M.numpy(), f) + b.numpy()[:, None, None]).astype(np.float32)
        return torch.from_numpy(self.m.decode(level, y))[None]


def _worker(rank, world, port, H, W, out_path, broadcast_map=False, halo_mode="recompute", style_mode="owner", style_w=96):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wct_hip import model_zoo
        from wct_hip.sharded import ShardedStylizer
        w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
        eng = OracleEngine(w)
        eng.o.set_num_threads(2)
        rng = np.random.default_rng(5)
        content = rng.random((3, H, W), dtype=np.float32)
        style = rng.random((3, 80, style_w), dtype=np.float32)
        sh = ShardedStylizer(eng, dist, H, W, 80, style_w, alpha=1.0, broadcast_map=broadcast_map, halo_mode=halo_mode, style_mode=style_mode)
        assert sh.halo_mode == halo_mode and sh.style_mode == style_mode
        x0, x1 = sh.input_columns()
        strip = sh.stylize_strip(torch.from_numpy(np.ascontiguousarray(content[:, :, x0:x1])), torch.from_numpy(style))
        parts = [None] * world
        dist.all_gather_object(parts, (sh.own, strip.numpy()))
        if rank == 0:
            full = np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0][0])], axis=3)
            np.save(out_path, full)
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,H,W,bmap,halo,smode,sw", [
    (2, 64, 1280, False, "recompute", "owner", 96), (3, 48, 1925, False, "recompute", "owner", 96),
    (6, 32, 1152, False, "recompute", "owner", 96),   # 6 ranks: every level's style side on another rank, rank 5 none
    (3, 48, 1925, True, "recompute", "owner", 96),    # rank 0 solves and broadcasts (M, b)
    (3, 48, 1925, False, "exchange", "owner", 96),    # exact per-level margins + neighbour exchange of the decoded edge columns;
    (6, 32, 1157, True, "exchange", "owner", 96),     # 1157 -> 1152 columns after level 5: the last strip shrinks before it is sent
    (8, 32, 1290, False, "exchange", "owner", 96),    # BASELINE configs[3]'s topology: 8 strips, neighbour exchange (1290 -> 1280 columns)
    # the STYLE cut into column strips like the content (every rank: its strip + the encoder's receptive field at every level, sums in the
    # level-5 all-reduce): 2 ranks; 3 ranks over a width that floor pooling shrinks (333 -> 320 feature-aligned columns at level 5) with
    # rank 0 solving and broadcasting (M, b); 8 ranks x 40 columns (narrower than the level-5 margin of 80: strips reach across
    # several neighbours); every rank repeating the whole style side
    (2, 64, 1280, False, "recompute", "strips", 96), (3, 48, 1925, True, "exchange", "strips", 333),
    (8, 32, 1290, False, "exchange", "strips", 320), (3, 48, 1925, False, "recompute", "replicate", 96)])
def test_sharded_equals_untiled(tmp_path, oracle, weights16x, world, H, W, bmap, halo, smode, sw):
    out = str(tmp_path / "sharded.npy")
    mp.spawn(_worker, args=(world, _free_port(), H, W, out, bmap, halo, smode, sw), nprocs=world, join=True)
    got = np.load(out)
    rng = np.random.default_rng(5)
    content = rng.random((3, H, W), dtype=np.float32)
    style = rng.random((3, 80, sw), dtype=np.float32)
    # untiled run of the SAME engine (world = 1): isolates the sharding logic from algorithmic differences
    from wct_hip.sharded import ShardedStylizer
    eng = OracleEngine(weights16x)
    one = ShardedStylizer(eng, None, H, W, 80, sw, rank=0, world=1)
    ref = one.stylize_strip(torch.from_numpy(content), torch.from_numpy(style)).numpy()
    assert got.shape == ref.shape              # W = 1925 shrinks to 1920 at level 5
    assert got.shape[3] == (W // 16) * 16 and got.shape[2] == (H // 16) * 16
    err = np.abs(got - ref).max() / np.abs(ref).max()
    assert err < 1e-6, err                     # identical arithmetic; only the moment summation order differs
    # and the whole thing is the reference cascade (SVD path) up to the cascade's own error amplification
    full = oracle.stylize(oracle.Modules("16x", weights16x), content, style, 1.0)
    assert np.abs(got[0] - full).max() / np.abs(full).max() < 1e-3


def test_strip_bounds():
    from wct_hip.sharded import CUM_HALO, LEVEL_HALO, ext_bounds, strip_bounds
    b = strip_bounds(3840 * 8, 8)
    assert b[0] == (0, 3840) and b[-1] == (3840 * 7, 3840 * 8)
    assert all(x0 % 16 == 0 for x0, _ in b)
    assert strip_bounds(10240, 8)[3] == (3840, 5120)     # BASELINE config 4: 8 strips of 1280
    for L in (5, 4, 3, 2):
        assert CUM_HALO[L] % (1 << (L - 1)) == 0
        assert CUM_HALO[L] - LEVEL_HALO[L] >= CUM_HALO[L - 1]   # what stays exact covers the next level's need
    assert CUM_HALO[1] >= LEVEL_HALO[1]
    assert ext_bounds((0, 1280), 10240, 272) == (0, 1552) and ext_bounds((8960, 10240), 10240, 272) == (8688, 10240)
    with pytest.raises(ValueError):
        strip_bounds(40, 4)
    # FLOP overhead of the two halo modes: config 4 on 8 GPUs = 1280-wide strips
    from wct_hip.sharded import ShardedStylizer, halo_flop_overhead
    assert 0.25 < halo_flop_overhead(1280, "recompute") < 0.27 and 0.15 < halo_flop_overhead(1280, "exchange") < 0.16
    assert ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=3, world=8).halo_mode == "exchange"
    assert ShardedStylizer(None, None, 2160, 3840 * 8, 2048, 2048, rank=3, world=8).halo_mode == "recompute"
    assert ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=3, world=8).input_columns() == (3840 - 160, 5120 + 160)
    # style side: "auto" deals the levels out whole (measured best at 2, 4 and 8 ranks); strips on request; one rank has nothing to share
    from wct_hip.sharded import STYLE_HALO
    assert ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=3, world=8).style_mode == "owner"
    assert ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=3, world=8, style_mode="strips").style_bounds[3] == (768, 1024)
    assert ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=0, world=1).style_mode == "replicate"
    assert all(STYLE_HALO[L] % (1 << (L - 1)) == 0 and STYLE_HALO[L] <= LEVEL_HALO[L] for L in (5, 4, 3, 2, 1))
    with pytest.raises(ValueError):
        ShardedStylizer(None, None, 64, 400, 64, 64, rank=0, world=4, halo_mode="exchange")


def _replica_worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wct_hip import model_zoo
        from wct_hip.replicas import ReplicaStylizer
        eng = OracleEngine(model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
        eng.o.set_num_threads(2)
        style = np.random.default_rng(2).random((3, 48, 64), dtype=np.float32)
        content = np.random.default_rng(10 + rank).random((3, 32 + 16 * rank, 48), dtype=np.float32)   # a different image per rank
        out = ReplicaStylizer(eng, dist).stylize(torch.from_numpy(content), torch.from_numpy(style))
        assert sorted(eng.style_stats) == [1, 2, 3, 4, 5]
        np.save(os.path.join(out_dir, "r%d.npy" % rank), out.numpy())
    finally:
        dist.destroy_process_group()


def test_replicas_share_the_style_side(tmp_path, weights16x):
    """BASELINE config 5: independent contents, one style.  Each rank computes some levels of the style statistics, all
    ranks end up with all of them, and every rank's result equals a single-process run on its own content."""
    world = 3
    mp.spawn(_replica_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    style = np.random.default_rng(2).random((3, 48, 64), dtype=np.float32)
    for rank in range(world):
        content = np.random.default_rng(10 + rank).random((3, 32 + 16 * rank, 48), dtype=np.float32)
        eng = OracleEngine(weights16x)
        eng.style_prepare(torch.from_numpy(style))
        ref = eng.stylize_prepared(torch.from_numpy(content)).numpy()
        got = np.load(str(tmp_path / ("r%d.npy" % rank)))
        assert got.shape == ref.shape and np.array_equal(got, ref)    # same engine, same statistics: identical
