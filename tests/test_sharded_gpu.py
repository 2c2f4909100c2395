"""The N > 1 path on the real engine.  The GPU test box has ONE MI355X, so the ranks share it and talk over gloo
(WCT_DIST_BACKEND=gloo; RCCL needs one device per rank): what runs is the product's sharding code (wct_hip/sharded.py),
libwct_hip's split-level C ABI and bench.py's multi-rank branches -- everything but the RCCL transport itself."""
import json
import os
import socket
import subprocess
import sys
import types

import numpy as np
import pytest

from tests.conftest import PKG, REPO, rel_err

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _shard_worker(rank, world, port, H, W, halo_mode, bmap, out_path, style_hw=(300, 260), frames="rand", backend="gloo", c_coll=False,
                  style_mode="auto", c_cascade=False):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    if backend == "nccl":            # one device per rank: RCCL over xGMI (only on a box with >= `world` devices)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)   # the ranks share the one GPU of the test box
    try:
        from wct_hip import WCT, model_zoo
        from wct_hip.sharded import ShardedStylizer
        w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
        wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
        if frames == "g16":          # the frame the reference itself was run on (tools/make_goldens.py gen_g16)
            from tests.fixture_compare import cfg4_geometry_frames
            content, style = (torch.from_numpy(a).cuda() for a in cfg4_geometry_frames())
            assert tuple(content.shape) == (3, H, W) and tuple(style.shape[1:]) == tuple(style_hw)
        else:
            g = torch.Generator(device="cuda").manual_seed(11)
            content = torch.rand((3, H, W), device="cuda", generator=g)
            style = torch.rand((3,) + tuple(style_hw), device="cuda", generator=g)
        if c_coll or c_cascade:      # the collectives inside the library (wct_level_sharded / wct_stylize_sharded): on its own RCCL communicator, or --
            if backend == "nccl":    # ranks sharing one GPU over gloo -- through the torch.distributed transport adapter (test infrastructure)
                wct.comm_init(dist)
            else:
                from tools.sharded_standins import dist_transport
                wct.comm_attach_collectives(*dist_transport(dist), world, rank)
            wct.comm_selftest()
        sh = ShardedStylizer(wct, dist, H, W, style_hw[0], style_hw[1], halo_mode=halo_mode, broadcast_map=bmap, style_mode=style_mode,
                             c_cascade=c_cascade, c_collectives=(bool(c_coll) if (c_coll or c_cascade) else None))
        assert sh.c_collectives == bool(c_coll) and sh.c_cascade == bool(c_cascade)
        x0, x1 = sh.input_columns()
        strip = sh.stylize_strip(content[:, :, x0:x1].contiguous(), style)
        wct.sync()
        parts = [None] * world
        dist.all_gather_object(parts, (sh.own, sh.halo_mode, strip.cpu().numpy()))
        if rank == 0:
            full = np.concatenate([p[2] for p in sorted(parts, key=lambda t: t[0][0])], axis=3)
            ref = wct.stylize(content, style).cpu().numpy()
            np.savez(out_path, got=full, ref=ref, modes=np.array([p[1] for p in parts]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,H,W,halo_mode,bmap,smode", [(2, 272, 1525, "recompute", False, "owner"), (2, 272, 1525, "exchange", False, "strips"),
                                                            (3, 144, 1168, "exchange", True, "owner"), (2, 272, 1525, "auto", True, "auto"),
                                                            (3, 144, 1168, "recompute", False, "strips")])
def test_sharded_ranks_match_untiled(tmp_path, world, H, W, halo_mode, bmap, smode):
    """wct_hip/sharded.py driving libwct_hip on the GPU: column strips (cumulative halos, or exact per-level margins + the
    neighbour exchange of decoded edge columns), all-reduced fp64 moments, replicated solve or broadcast (M, b)  ==  the
    untiled HIP cascade.  Not bitwise: the runs' (M, b) differ by ~1e-13 (moment summation order), which flips fp32 roundings
    of the folded decoder weights at level 5, and the cascade amplifies that level by level (tools/debug/shard_diag.py); the
    strips themselves are bit-exact given the same (M, b) (test_strip_halos_exact_per_level)."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "sh.npz")
    mp.spawn(_shard_worker, args=(world, _free_port(), H, W, halo_mode, bmap, out, (300, 260), "rand", "gloo", False, smode), nprocs=world, join=True)
    z = np.load(out)
    assert z["got"].shape == z["ref"].shape == (1, 3, H // 16 * 16, W // 16 * 16)
    assert all(m == ("exchange" if halo_mode == "auto" else halo_mode) for m in z["modes"])
    assert rel_err(z["got"], z["ref"]) < 5e-4


@pytest.mark.parametrize("world,H,W,halo_mode,bmap,smode", [(2, 272, 1525, "exchange", False, "owner"), (2, 272, 1525, "recompute", True, "owner"),
                                                            (3, 144, 1168, "exchange", True, "owner"), (4, 208, 2560, "exchange", False, "owner"),
                                                            (2, 272, 1525, "exchange", False, "strips"), (3, 144, 1168, "recompute", True, "strips")])
def test_device_buffer_collectives_are_checked(tmp_path, world, H, W, halo_mode, bmap, smode):
    """The DEVICE-BUFFER branch of the sharded path, checked for correctness (VERDICT r3 task 4): all ranks of the job as threads
    of this process, one engine and one stream each, collectives that move device buffers ordered by events only
    (tools/sharded_standins.py InProcessWorld: `get_backend() == "nccl"`, so sharded._p2p takes its no-staging branch) -- real asynchrony
    between the ranks' streams and the library's two lanes, which gloo's host staging hides.  Checked:
      * against the untiled cascade (the tolerance of the gloo test);
      * bitwise against the SAME job with a device-wide synchronisation around every collective (a missing event dependency
        between style_export -> broadcast -> style_import, moments -> all_reduce or decoded columns -> send would differ);
      * bitwise against itself run again (no dependence on thread timing);
      * world = 2: bitwise against the multi-process gloo job on the same inputs (a two-term sum is commutative: the host-staged
        and the device-buffer transports must agree to the last bit);
      * the collectives seen: 5 all-reduces, the level's broadcasts, 4 neighbour exchanges in exchange mode."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools import sharded_standins as standins
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(11)
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 300, 260), device="cuda", generator=g)
    got, groups = standins.run_in_process(world, make, content, style, halo_mode=halo_mode, broadcast_map=bmap, style_mode=smode)
    ref = make().stylize(content, style)
    assert tuple(got.shape) == tuple(ref.shape) == (1, 3, H // 16 * 16, W // 16 * 16)
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 5e-4
    synced, _ = standins.run_in_process(world, make, content, style, sync_every=True, halo_mode=halo_mode, broadcast_map=bmap, style_mode=smode)
    assert torch.equal(got, synced)
    again, _ = standins.run_in_process(world, make, content, style, halo_mode=halo_mode, broadcast_map=bmap, style_mode=smode)
    assert torch.equal(got, again)
    for grp in groups:
        assert grp.calls["all_reduce"] == 5       # strips: the style sums ride in level 5's
        assert grp.calls["p2p"] == (4 if halo_mode == "exchange" else 0)
        # owner mode: style statistics travel to every solver that does not own the level; with broadcast_map only rank 0 solves, and (M, b) follows
        if smode == "owner":
            assert grp.calls["broadcast"] == ((5 + sum(1 for L in (5, 4, 3, 2, 1) if (5 - L) % world != 0)) if bmap else 5)
        else:
            assert grp.calls["broadcast"] == (5 if bmap else 0)
    if world == 2:
        import torch.multiprocessing as mp
        out = str(tmp_path / "sh.npz")
        mp.spawn(_shard_worker, args=(world, _free_port(), H, W, halo_mode, bmap, out, (300, 260), "rand", "gloo", False, smode), nprocs=world, join=True)
        assert np.array_equal(np.load(out)["got"], got.cpu().numpy())


@pytest.mark.parametrize("world,H,W,halo_mode,bmap,smode,style_hw", [
    (2, 272, 1525, "exchange", False, "strips", (300, 260)), (3, 144, 1168, "exchange", True, "owner", (300, 260)),
    (4, 208, 2560, "recompute", False, "strips", (200, 333)), (3, 144, 1168, "recompute", False, "replicate", (300, 260)),
    (2, 272, 1525, "recompute", True, "strips", (300, 260)), (5, 80, 2000, "exchange", False, "owner", (120, 200))])
def test_c_cascade_bitwise_equals_python_orchestration(world, H, W, halo_mode, bmap, smode, style_hw):
    """VERDICT r5 task 1: the WHOLE column-sharded cascade behind the C ABI (include/wct_hip.h wct_stylize_sharded -- strip geometry, per-level
    crops, the style side in all three arrangements, the all-reduces, the style-statistics / (M, b) broadcasts, the grouped neighbour
    exchange, all issued by the library on the context's streams) against wct_hip/sharded.py's Python orchestration of the split-level
    entries: every rank of the job is a thread of this process with its own context and stream, and BOTH paths talk through the same
    device-buffer collectives (tools/sharded_standins.py InProcessWorld; the C cascade through a wct_collectives table of callbacks into
    them, wct_comm_attach_collectives).  BITWISE equal, the same collectives counted, bitwise equal again with a device-wide
    synchronisation around every collective (no missing stream dependency between the library's lanes and the transport), and within
    the sharded tests' tolerance of the untiled frame."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools import sharded_standins as standins
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(11)
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3,) + style_hw, device="cuda", generator=g)
    kw = dict(halo_mode=halo_mode, broadcast_map=bmap, style_mode=smode)
    want, groups_py = standins.run_in_process(world, make, content, style, **kw)
    got, groups_c = standins.run_in_process(world, make, content, style, c_cascade=True, **kw)
    assert tuple(got.shape) == tuple(want.shape) == (1, 3, H // 16 * 16, W // 16 * 16)
    assert torch.equal(got, want), float((got - want).abs().max())
    assert [g_.calls for g_ in groups_c] == [g_.calls for g_ in groups_py]
    synced, _ = standins.run_in_process(world, make, content, style, c_cascade=True, sync_every=True, **kw)
    assert torch.equal(got, synced)
    ref = make().stylize(content, style)
    assert rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 5e-4


@pytest.mark.parametrize("world,H,W,halo_mode,bmap,smode", [(3, 144, 1168, "exchange", True, "owner"), (4, 208, 2560, "recompute", False, "strips"),
                                                            (8, 96, 2560, "exchange", False, "owner")])
def test_c_cascade_multi_process_equals_python_orchestration(tmp_path, world, H, W, halo_mode, bmap, smode):
    """The library's cascade in a MULTI-PROCESS job (one process per rank, as deployed; here the ranks share the one GPU and the library's transport
    table is the torch.distributed adapter over gloo, host-staged): bitwise the Python orchestration of the same job over the same process group
    (gloo sums the ranks' terms in one order for both), selftest included -- 3, 4 and 8 ranks, neighbour exchange, (M, b) broadcast, style strips."""
    import torch.multiprocessing as mp
    outs = []
    for c_cascade in (False, True):
        out = str(tmp_path / ("c%d.npz" % c_cascade))
        mp.spawn(_shard_worker, args=(world, _free_port(), H, W, halo_mode, bmap, out, (200, 333), "rand", "gloo", False, smode, c_cascade), nprocs=world, join=True)
        outs.append(np.load(out))
    assert np.array_equal(outs[0]["got"], outs[1]["got"])
    assert rel_err(outs[1]["got"], outs[1]["ref"]) < 5e-4


@pytest.mark.parametrize("world,halo_mode,smode", [(2, "recompute", "owner"), (3, "exchange", "strips")])
def test_c_cascade_mode_original(world, halo_mode, smode):
    """--mode original (feature maps up to 512 channels wide: the deflated multi-launch solves, which read one flag back per solve) through the library's
    cascade -- round 5's wct_level_sharded refused that mode -- bitwise the Python orchestration of the split-level entries, and the untiled frame within
    the tolerance the conditioned generated weights allow (model_zoo.synth_weights_conditioned, the set G15's strict gate uses)."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools import sharded_standins as standins
    w = model_zoo.synth_weights_conditioned("original", 15)
    make = lambda: WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(21)
    H, W = 112, 160 * world + 304
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 128, 176), device="cuda", generator=g)
    kw = dict(halo_mode=halo_mode, style_mode=smode)
    want, gp = standins.run_in_process(world, make, content, style, **kw)
    got, gc = standins.run_in_process(world, make, content, style, c_cascade=True, **kw)
    assert torch.equal(got, want), float((got - want).abs().max())
    assert [x.calls for x in gc] == [x.calls for x in gp]
    ref = make().stylize(content, style)
    assert tuple(got.shape) == tuple(ref.shape) and rel_err(got.cpu().numpy(), ref.cpu().numpy()) < 2e-3


@pytest.mark.parametrize("world,H,W,halo_mode,smode", [(3, 144, 1168, "exchange", "owner"), (4, 208, 2560, "recompute", "strips")])
def test_c_cascade_fast_fold(world, H, W, halo_mode, smode):
    """WCT_SHARD_FAST_FOLD: the library's cascade folding the way the single-GPU cascade does ((W cov_s^1/2) cov_c^-1/2 straight into the decoder's first
    conv, no (M, b) on the critical path) -- what bench.py times for N > 1.  fp32 round-off from the default (M, b) form (<= 1e-4 after five levels, the
    bound of test_fast_fold_matches_the_map_based_fold), reproducible bit for bit, and at least as close to the untiled frame (which folds the same way)."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools import sharded_standins as standins
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(11)
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 300, 260), device="cuda", generator=g)
    kw = dict(halo_mode=halo_mode, style_mode=smode, c_cascade=True)
    base, _ = standins.run_in_process(world, make, content, style, **kw)
    fast, gf = standins.run_in_process(world, make, content, style, fast_fold=True, **kw)
    again, _ = standins.run_in_process(world, make, content, style, fast_fold=True, **kw)
    assert torch.equal(fast, again)          # deterministic (the two folds differ in fp64 association BEFORE the weights are rounded to fp32: usually the same bits)
    ref = make().stylize(content, style).cpu().numpy()
    e_fast, e_base = rel_err(fast.cpu().numpy(), ref), rel_err(base.cpu().numpy(), ref)
    print("\n[fast fold, %d ranks] fast vs (M, b) form %.2e; vs untiled: fast %.2e, (M, b) form %.2e" % (world, rel_err(fast.cpu().numpy(), base.cpu().numpy()), e_fast, e_base))
    assert rel_err(fast.cpu().numpy(), base.cpu().numpy()) < 1e-4 and e_fast < 5e-4


def test_c_cascade_random_geometries():
    """tools/debug/cascade_fuzz.py: sixteen random jobs (2..8 ranks, odd frame sizes and widths that floor pooling shrinks, random style sizes, every
    halo / style / map arrangement, alpha 1 and 0.6) through the library's cascade and through the Python orchestration: bitwise equal, the same
    collectives, and the untiled frame within the sharded tests' tolerance (40 more such cases ran in round 6: none differed)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "debug", "cascade_fuzz.py"), "16", "7"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "failures: 0" in r.stdout, (r.stdout[-3000:], r.stderr[-2000:])
    assert r.stdout.count("bitwise True") >= 10        # (a drawn geometry that cannot be cut as asked is skipped)


def test_c_cascade_refuses_what_it_cannot_run():
    """No silent fallback: without a communicator or transport wct_stylize_sharded returns WCT_ERR_STATE; content columns other than
    wct_shard_geometry's are refused; the geometry function agrees with sharded.py for config 4."""
    import torch
    from wct_hip import WCT, model_zoo
    from wct_hip.lib import WctError
    from wct_hip.sharded import ShardedStylizer
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
    c, s = torch.rand((3, 64, 640), device="cuda"), torch.rand((3, 64, 64), device="cuda")
    assert wct.comm_info() == (0, 0)
    with pytest.raises((WctError, RuntimeError), match="communicator"):
        wct.stylize_sharded(c, s, 640, 0, 640)
    with pytest.raises(ValueError):
        ShardedStylizer(wct, None, 64, 640, 64, 64, rank=0, world=1, c_cascade=True)
    for r in range(8):
        sh = ShardedStylizer(None, None, 4096, 10240, 2048, 2048, rank=r, world=8)
        assert wct.shard_geometry(10240, 8, r, "auto") == (sh.own[0], sh.own[1]) + sh.input_columns() + ("exchange",)
        sh = ShardedStylizer(None, None, 2160, 3840 * 8, 2048, 2048, rank=r, world=8)
        assert wct.shard_geometry(3840 * 8, 8, r, "auto") == (sh.own[0], sh.own[1]) + sh.input_columns() + ("recompute",)
    with pytest.raises(ValueError):
        wct.shard_geometry(400, 4, 0, "exchange")
    ok = lambda *a: 0     # noqa: E731
    wct.comm_attach_collectives(ok, ok, ok, 2, 1)
    assert wct.comm_info() == (2, 1)
    with pytest.raises(ValueError, match="columns"):
        wct.stylize_sharded(c, s, 1280, 0, 640)       # rank 1 of 2 over 1280 columns owns [640, 1280): these are not its columns
    wct.comm_destroy()
    assert wct.comm_info() == (0, 0)


def test_config4_eight_strips_match_untiled(tmp_path):
    """BASELINE configs[3] as specified: ONE 10240x4096 content (2048x2048 style) in EIGHT column strips of 1280 -- eight
    ranks (here sharing the one GPU over gloo; on the 8-GPU node: RCCL), halo mode "auto" = neighbour exchange at this strip
    width, style statistics computed once per node and broadcast, all-reduced moments -- against the untiled single-GPU
    cascade of the same frame."""
    import torch.multiprocessing as mp
    out = str(tmp_path / "sh8.npz")
    mp.spawn(_shard_worker, args=(8, _free_port(), 4096, 10240, "auto", False, out, (2048, 2048)), nprocs=8, join=True)
    z = np.load(out)
    assert z["got"].shape == z["ref"].shape == (1, 3, 4096, 10240)
    assert all(m == "exchange" for m in z["modes"])
    e = rel_err(z["got"], z["ref"])
    print("\n[cfg4 8 strips vs untiled] rel_err=%.3e" % e)
    assert np.isfinite(z["got"]).all() and e < 1e-3


def test_config4_eight_strips_device_buffers_in_process():
    """The same job -- 10240x4096 in eight 1280-column strips, exchange halos, 2048x2048 style -- with every collective on DEVICE buffers:
    eight rank threads, eight engines, eight streams in this process (tools/sharded_standins.py InProcessWorld), i.e. the geometry of the 8-GPU
    configuration (edge strips with one halo, interior strips with two, the style levels dealt round robin, 3.5 MB edge-column messages)
    through the branch of sharded.py that RCCL takes.  Against the untiled cascade (the gloo test's bound), and bitwise against the same
    job with a device-wide synchronisation around every collective."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools import sharded_standins as standins
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(11)
    content = torch.rand((3, 4096, 10240), device="cuda", generator=g)
    style = torch.rand((3, 2048, 2048), device="cuda", generator=g)
    got, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="owner")
    ref = make().stylize(content, style)
    e = float((got - ref).abs().max() / ref.abs().max())
    print("\n[cfg4 8 strips, device buffers, in process] rel_err=%.3e" % e)
    assert tuple(got.shape) == (1, 3, 4096, 10240) and bool(torch.isfinite(got).all()) and e < 1e-3
    assert all(grp.calls == {"all_reduce": 5, "broadcast": 5, "p2p": 4} for grp in groups)
    synced, _ = standins.run_in_process(8, make, content, style, sync_every=True, halo_mode="auto", style_mode="owner")
    assert torch.equal(got, synced)
    del synced
    # round 6: the STYLE cut into eight 256-column strips as well (style_mode "strips"), and the frame as ONE library call per rank
    strips, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="strips")
    es = float((strips - ref).abs().max() / ref.abs().max())
    print("[cfg4 8 strips, style in strips too] rel_err=%.3e  vs owner-mode job %.3e" % (es, float((strips - got).abs().max() / ref.abs().max())))
    assert es < 1e-3 and all(grp.calls == {"all_reduce": 5, "broadcast": 0, "p2p": 4} for grp in groups)
    del got, ref
    ccas, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="strips", c_cascade=True)
    assert torch.equal(ccas, strips) and all(grp.calls == {"all_reduce": 5, "broadcast": 0, "p2p": 4} for grp in groups)


def test_config4_geometry_vs_reference(tmp_path, oracle):
    """BASELINE configs[3]'s geometry against THE REFERENCE'S OWN PIXELS (G16, VERDICT r4 task 4): a 10240-wide x 512-tall content
    (seed 5) + the 2048x2048 style (seed 2) went through util_wct.WCT itself (tools/make_goldens.py gen_g16; 1/16 lattice + eight
    96x96 crops, six of them straddling strip boundaries x = 1280 k).  Under THE GATE (the rule of tests/test_hip_scale.py
    and bench.py: hip_vs_reference <= max(1e-3, 1.25 x oracle_vs_reference), the oracle arm recomputed HERE on the box's host cores --
    on this frame it sits at 1.10e-3 with ONE lattice pixel of 983 040 beyond 1e-3, p99.99 2.7e-4: uniform noise through five whitenings):
      (a) the untiled HIP frame;
      (b) the 8 x 1280 exchange-halo job, eight ranks over gloo (multi-process, the ranks share this GPU);
      (c) the same job with every collective on DEVICE buffers (tools/sharded_standins.py InProcessWorld -- the branch RCCL takes);
    (b) and (c) must also agree with each other to fp32-rounding level (different summation orders of eight moment terms)."""
    import torch
    import torch.multiprocessing as mp
    from tests.conftest import load_golden
    from tests.fixture_compare import GATE, cfg4_geometry_frames, compare_to_fixture
    from tools import sharded_standins as standins
    from wct_hip import WCT, model_zoo
    g16 = load_golden("g16_cfg4_geometry.npz")
    c_np, s_np = cfg4_geometry_frames()
    assert abs(float(c_np.sum(dtype=np.float64)) - float(g16["content.checksum"])) < 1e-6
    assert abs(float(s_np.sum(dtype=np.float64)) - float(g16["style.checksum"])) < 1e-6
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    content, style = torch.from_numpy(c_np).cuda(), torch.from_numpy(s_np).cuda()
    eng = make()
    untiled = eng.stylize(content, style).cpu().numpy()[0]
    assert eng.saturation_count() == 0
    ru = compare_to_fixture(untiled, g16)
    out = str(tmp_path / "g16.npz")
    mp.spawn(_shard_worker, args=(8, _free_port(), 512, 10240, "auto", False, out, (2048, 2048), "g16", "gloo", False, "owner"), nprocs=8, join=True)
    z = np.load(out)
    assert all(m == "exchange" for m in z["modes"])
    rg = compare_to_fixture(z["got"][0], g16)
    inproc, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="owner")
    assert all(grp.calls == {"all_reduce": 5, "broadcast": 5, "p2p": 4} for grp in groups)
    ri = compare_to_fixture(inproc.cpu().numpy()[0], g16)
    # (d) the same job as ONE library call per rank (wct_stylize_sharded, style levels dealt out whole: the round-6 default path) ...
    ccas, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", c_cascade=True)
    assert all(grp.calls == {"all_reduce": 5, "broadcast": 5, "p2p": 4} for grp in groups) and torch.equal(ccas, inproc)
    # ... and with the style cut into strips too
    ccas, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="strips", c_cascade=True)
    assert all(grp.calls == {"all_reduce": 5, "broadcast": 0, "p2p": 4} for grp in groups)
    rc = compare_to_fixture(ccas.cpu().numpy()[0], g16)
    print("\n[G16 cfg4 geometry vs REFERENCE] C cascade, style strips: %.3e (p99.99 %.3e)" % (rc["max"], rc["lattice_p9999"]))
    assert rc["max"] <= GATE and rc["lattice_p9999"] <= GATE / 2
    oracle.set_num_threads(min(os.cpu_count() or 1, 32))
    ro = compare_to_fixture(oracle.stylize(oracle.Modules("16x", w), c_np, s_np, 1.0), g16)
    limit = max(GATE, 1.25 * ro["max"])
    print("\n[G16 cfg4 geometry] oracle_vs_reference %.3e (p99.99 %.3e)  limit %.3e" % (ro["max"], ro["lattice_p9999"], limit))
    print("\n[G16 cfg4 geometry vs REFERENCE] untiled %.3e (p99.99 %.3e) | 8x1280 gloo %.3e (p99.99 %.3e) | 8x1280 device buffers %.3e (p99.99 %.3e)"
          " | gloo vs device buffers %.3e | sharded vs untiled %.3e"
          % (ru["max"], ru["lattice_p9999"], rg["max"], rg["lattice_p9999"], ri["max"], ri["lattice_p9999"],
             rel_err(z["got"], inproc.cpu().numpy()), rel_err(z["got"][0], untiled)))
    assert ro["max"] <= 1.5e-3                                                         # the oracle stays where it was measured (1.10e-3)
    assert ru["max"] <= limit and rg["max"] <= limit and ri["max"] <= limit            # THE GATE, all three forms of the job
    assert max(ru["max"], rg["max"], ri["max"]) <= GATE                                # ... and the LITERAL 1e-3 (measured 5.7e-4 / 6.0e-4): a 2x drift fails
    assert max(ru["lattice_p9999"], rg["lattice_p9999"], ri["lattice_p9999"]) <= GATE / 2
    assert rel_err(z["got"], inproc.cpu().numpy()) < 5e-4


def test_config4_geometry_natural_image_vs_reference(oracle):
    """BASELINE configs[3]'s geometry on a NATURAL image against the reference's own pixels (G17, VERDICT r5 task 7): the reference's UHD
    sample tiled to 10240 x 512 + its style/in1.jpg went through util_wct.WCT itself (tools/make_goldens.py gen_g17).  Uniform noise
    (G16) leaves the reference's arithmetic ~1e-3 of room -- the oracle itself sits at 1.1e-3 there; a natural image an order of magnitude
    more, so here the bounds have real headroom: the oracle, the untiled HIP frame, the 8 x 1280 job through the Python orchestration and
    through the library's own cascade (wct_stylize_sharded, style in strips) each within 1e-4 of the reference's pixels (literal gate: 1e-3)."""
    import torch
    from tests.conftest import GOLD, load_golden
    from tests.fixture_compare import cfg4_natural_frames, compare_to_fixture
    from tools import sharded_standins as standins
    from wct_hip import WCT, model_zoo
    g17 = load_golden("g17_cfg4_geometry_natural.npz")
    c_np, s_np = cfg4_natural_frames(GOLD)
    assert abs(float(c_np.sum(dtype=np.float64)) - float(g17["content.checksum"])) < 1e-6       # the JPEG decoder of this box = the fixture's
    assert abs(float(s_np.sum(dtype=np.float64)) - float(g17["style.checksum"])) < 1e-6
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    content, style = torch.from_numpy(c_np).cuda(), torch.from_numpy(s_np).cuda()
    eng = make()
    ru = compare_to_fixture(eng.stylize(content, style).cpu().numpy()[0], g17)
    assert eng.saturation_count() == 0
    py, _ = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="owner")
    rp = compare_to_fixture(py.cpu().numpy()[0], g17)
    cc, groups = standins.run_in_process(8, make, content, style, halo_mode="auto", style_mode="strips", c_cascade=True)
    assert all(grp.calls == {"all_reduce": 5, "broadcast": 0, "p2p": 4} for grp in groups)
    rc = compare_to_fixture(cc.cpu().numpy()[0], g17)
    oracle.set_num_threads(min(os.cpu_count() or 1, 32))
    ro = compare_to_fixture(oracle.stylize(oracle.Modules("16x", w), c_np, s_np, 1.0), g17)
    print("\n[G17 cfg4 geometry, natural image, vs REFERENCE] oracle %.3e | untiled %.3e | 8x1280 python %.3e | 8x1280 C cascade %.3e (p99.99 %.3e)"
          % (ro["max"], ru["max"], rp["max"], rc["max"], rc["lattice_p9999"]))
    assert max(ro["max"], ru["max"], rp["max"], rc["max"]) <= 1e-4


def _replica_worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wct_hip import WCT, model_zoo
        from wct_hip.replicas import ReplicaStylizer
        wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
        style = torch.rand((3, 320, 256), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
        content = torch.rand((3, 272 + 32 * rank, 400), device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 + rank))
        out = ReplicaStylizer(wct, dist).stylize(content, style)
        wct.sync()
        np.save(os.path.join(out_dir, "r%d.npy" % rank), out.cpu().numpy())
    finally:
        dist.destroy_process_group()


def test_replicas_three_ranks_on_the_hip_engine(tmp_path):
    """BASELINE configs[4] ("batch of distinct contents x 1 style, one per GPU"): wct_hip/replicas.py on the HIP engine, three
    ranks (sharing the one GPU over gloo), a different content per rank, each level's style statistics computed by ONE rank
    and broadcast -- every rank's result equals a single-engine stylisation of its content bit for bit."""
    import torch
    import torch.multiprocessing as mp
    from wct_hip import WCT, model_zoo
    world = 3
    mp.spawn(_replica_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
    style = torch.rand((3, 320, 256), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    for rank in range(world):
        content = torch.rand((3, 272 + 32 * rank, 400), device="cuda", generator=torch.Generator(device="cuda").manual_seed(10 + rank))
        ref = wct.stylize(content, style).cpu().numpy()
        got = np.load(str(tmp_path / ("r%d.npy" % rank)))
        assert got.shape == ref.shape and np.array_equal(got, ref), rank


def _range_worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wct_hip import WCT, model_zoo
        from wct_hip.sharded import ShardedStylizer
        wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
        g = torch.Generator(device="cuda").manual_seed(17)
        H, W = 160, 2048
        content = torch.rand((3, H, W), device="cuda", generator=g)
        style = torch.rand((3, 200, 180), device="cuda", generator=g)
        content[1, 80, 1900] = 3.0e6            # far inside rank 1's strip (and beyond rank 0's 272-column halo): only rank 1 clamps
        sh = ShardedStylizer(wct, dist, H, W, 200, 180, halo_mode="recompute")
        x0, x1 = sh.input_columns()
        out = sh.stylize_strip(content[:, :, x0:x1].contiguous(), style)       # must complete on BOTH ranks (no rank raises mid-frame)
        own_clamped = wct.saturation_count() > 0
        try:
            sh.check_range()
            raised = False
        except OverflowError:
            raised = True
        # a pipelined loop WITHOUT per-frame synchronisation: the flag of frame k is read at the start of frame k + 2 on
        # every rank -- a fixed lag, not "whichever read-back has landed on this rank" -- so both ranks complete frames 1 and 2 and
        # both raise at the start of frame 3, before any of its collectives
        wct.saturation_count(reset=True)
        seq = []
        for _ in range(3):
            try:
                sh.stylize_strip(content[:, :, x0:x1].contiguous(), style)
                seq.append(0)
            except OverflowError:
                seq.append(1)
        dist.barrier()                             # nobody is stuck in a collective its peer skipped
        with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
            f.write("%d %d %d %s" % (int(own_clamped), int(raised), int(bool(torch.isfinite(out).all())), " ".join(str(v) for v in seq)))
    finally:
        dist.destroy_process_group()


def test_range_flag_travels_with_the_moments(tmp_path):
    """A clamp of the f16x3 arithmetic on ONE rank (a 3e6 pixel in rank 1's strip) is seen by EVERY rank: each rank's saturation
    counter rides as one more double in the all-reduce of the moments, ShardedStylizer.check_range() raises OverflowError on both,
    and nobody raises in the middle of the frame (the engine's per-call check is suspended inside stylize_strip: a rank that
    stopped there would leave its peers waiting in the next collective).  In a loop without check_range() the report comes at a
    FIXED lag (frame k's flag at the start of frame k + 2), identically on every rank."""
    import torch.multiprocessing as mp
    mp.spawn(_range_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0 = [int(v) for v in open(str(tmp_path / "r0.txt")).read().split()]
    r1 = [int(v) for v in open(str(tmp_path / "r1.txt")).read().split()]
    assert r0 == [0, 1, 1, 0, 0, 1], r0      # rank 0 did not clamp itself, raised all the same, finished its strip; pipelined: frame 3
    assert r1 == [1, 1, 1, 0, 0, 1], r1


def _cfg5_worker(rank, world, port, out_dir):
    for p in (REPO, PKG):
        if p not in sys.path:
            sys.path.insert(0, p)
    import hashlib
    import torch
    import torch.distributed as dist
    from tests.fixture_compare import noise_frame
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from wct_hip import WCT, model_zoo
        from wct_hip.replicas import ReplicaStylizer
        wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
        style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
        content = torch.from_numpy(noise_frame(10 + rank, 2160, 3840)).cuda()
        out = ReplicaStylizer(wct, dist).stylize(content, style)
        wct.sync()
        assert tuple(out.shape) == (1, 3, 2160, 3840) and wct.saturation_count() == 0
        with open(os.path.join(out_dir, "r%d.sha" % rank), "w") as f:
            f.write(hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest())
        with open(os.path.join(out_dir, "r%d.aborts" % rank), "w") as f:
            f.write("%d %d" % (int(wct.debug_get("nscoop_aborts")), int(wct.debug_get("nscoop_solves"))))
        np.save(os.path.join(out_dir, "r%d.npy" % rank), out.cpu().numpy()[:, :, ::16, ::16])
    finally:
        dist.destroy_process_group()


def test_config5_eight_4k_contents_one_style(tmp_path):
    """BASELINE configs[4] AT ITS SIZE: eight distinct 3840x2160 contents (SURVEY 8d seeds 10..17) x one 2048x2048 style (seed 2),
    one content per rank -- eight ranks (sharing the one GPU over gloo; on the 8-GPU node: one GPU each, xGMI idle), each level's
    style statistics computed by ONE rank and broadcast (wct_hip/replicas.py; PytorchWCT/data_loader.py:32-36 builds the content x
    style pairs).  Every rank's image equals the single-engine `stylize_prepared` of its content BIT FOR BIT (sha256 of the
    result), and so does wct_hip/pipeline.py's FramePipeline with three of these frames in flight on one GPU."""
    import hashlib
    import torch
    import torch.multiprocessing as mp
    from tests.fixture_compare import noise_frame
    from wct_hip import WCT, model_zoo
    from wct_hip.pipeline import FramePipeline
    world = 8
    mp.spawn(_cfg5_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    weights = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights)   # noqa: E731
    wct = make()
    style = torch.from_numpy(noise_frame(2, 2048, 2048)).cuda()
    contents = [torch.from_numpy(noise_frame(10 + r, 2160, 3840)).cuda() for r in range(world)]
    wct.style_prepare(style)
    sha = lambda t: hashlib.sha256(t.cpu().numpy().tobytes()).hexdigest()   # noqa: E731
    want = [sha(wct.stylize_prepared(c)) for c in contents]
    assert len(set(want)) == world                                    # distinct contents, distinct results
    aborts = [open(str(tmp_path / ("r%d.aborts" % r))).read() for r in range(world)]
    print("\n[cfg5] single-launch solves aborted / enqueued per rank:", aborts)
    # Bitwise -- unless a rank's single-launch matrix-function solve ABORTED into the Jacobi net (include/wct_hip.h wct_debug_get "nscoop_aborts"):
    # eight PROCESSES share this one GPU here, a solve's 32 workgroups must all be resident on one XCD while seven other processes' persistent
    # kernels hold CUs, and now and then (round 6: 4 of ~260 process runs, tools/debug/cfg5_stress.py) the 5 ms watchdog fires first.  The repaired
    # solve is correct but not bit-identical (measured 5.6e-6 / 6.8e-6 of the image's maximum after the cascade).  So: no abort -> equal hashes;
    # an abort on that rank -> within 5e-5 on the 1/256 lattice the worker kept.  One process per GPU (the product's deployment) never aborts.
    for r in range(world):
        if open(str(tmp_path / ("r%d.sha" % r))).read() == want[r]:
            continue
        n_abort = int(aborts[r].split()[0])
        ref = wct.stylize_prepared(contents[r]).cpu().numpy()[:, :, ::16, ::16]
        got = np.load(str(tmp_path / ("r%d.npy" % r)))
        mag = float(np.abs(got - ref).max() / np.abs(ref).max())
        print("[cfg5] rank %d: %d aborted solve(s), image differs from the single engine by %.3e of its maximum (1/256 lattice)" % (r, n_abort, mag))
        assert n_abort > 0 and mag < 5e-5, (r, aborts, mag)
    free0, _ = torch.cuda.mem_get_info()
    pipe = FramePipeline(make, slots=3)
    outs = pipe.stylize_many(contents, style=style)
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert [sha(o) for o in outs] == want
    print("\n[cfg5 at size] 8 ranks + 3-slot pipeline bitwise equal; pipeline device memory %.1f GiB" % ((free0 - free1) / 2**30))
    assert (free0 - free1) / 2**30 < 24.0


def test_rank_simulation_runs_the_sharded_path(tmp_path):
    """bench.py's passes.cfg4_rank_sim relies on tools/sharded_standins.py LoopbackGroup: one rank of an 8-rank job with its peers
    emulated on the device.  Shapes and call order must be those of the real job: rank 3's strip of a 2560-wide frame (8 x 320
    columns, exchange mode) comes back with the owned width, finite, and the group saw 5 broadcasts and 5 all-reduces."""
    import torch
    from wct_hip import WCT, model_zoo
    from tools.sharded_standins import LoopbackGroup
    from wct_hip.sharded import ShardedStylizer
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
    g = torch.Generator(device="cuda").manual_seed(4)
    H, W = 208, 2560
    content, style = torch.rand((3, H, W), device="cuda", generator=g), torch.rand((3, 256, 240), device="cuda", generator=g)
    wct.style_prepare(style)
    stats = {L: wct.style_export(L).clone() for L in (5, 4, 3, 2, 1)}
    calls = {"ar": 0}

    class Counting(LoopbackGroup):
        def all_reduce(self, t, op=None):
            calls["ar"] += 1
    for r in (0, 3, 7):
        grp = Counting(r, 8)
        grp.style_stats = stats
        sh = ShardedStylizer(wct, grp, H, W, 256, 240, halo_mode="exchange", style_mode="owner")
        x0, x1 = sh.input_columns()
        out = sh.stylize_strip(content[:, :, x0:x1].contiguous(), style)
        wct.sync()
        assert tuple(out.shape) == (1, 3, H, 320) and bool(torch.isfinite(out).all()) and grp._bcast == 5
    assert calls["ar"] == 15


def test_rccl_first_contact_single_rank():
    """The GPU test box has one device, so RCCL cannot carry a 2-rank job here; this at least executes the calls bench.py and
    wct_hip/sharded.py make -- init_process_group("nccl", device_id=...), all_reduce(SUM) of fp64 moments, broadcast, a
    MAX all-reduce, barrier -- on a 1-rank RCCL communicator, in a fresh process."""
    code = (
        "import os, torch, torch.distributed as dist\n"
        "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1')\n"
        "torch.cuda.set_device(0)\n"
        "dist.init_process_group('nccl', device_id=torch.device('cuda', 0))\n"
        "x = torch.arange(128 * 128 + 128, dtype=torch.float64, device='cuda'); y = x.clone()\n"
        "dist.all_reduce(x); dist.broadcast(x, src=0)\n"
        "t = torch.tensor([1.5], device='cuda', dtype=torch.float64); dist.all_reduce(t, op=dist.ReduceOp.MAX)\n"
        "dist.barrier(); torch.cuda.synchronize()\n"
        "assert torch.equal(x, y) and float(t) == 1.5 and dist.get_backend() == 'nccl'\n"
        "dist.destroy_process_group(); print('RCCL_OK')\n" % _free_port())
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "RCCL_OK" in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])


@pytest.mark.parametrize("extra,name", [([], "cfg2x2"), (["--config", "cfg4", "--halo-mode", "exchange"], "cfg4")])
def test_bench_two_ranks_on_one_gpu(extra, name):
    """bench.py's N = 2 code path exactly as the driver launches it (torch.distributed.run, one process per rank), ranks
    sharing the one GPU over gloo: the JSON contract of the multi-rank line, both workloads."""
    env = dict(os.environ, WCT_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"] + extra
    r = subprocess.run(cmd, env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 prints ONE line
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["unit"] == "MP/s" and line["config"]["name"] == name
    # a multi-rank line carries a correctness signal (VERDICT r5 missing #4): the library's cascade bitwise equal to the torch.distributed
    # orchestration on every rank, every strip against the untiled frame, configs[3]'s geometry against the reference's pixels (G16)
    assert line["value"] > 0 and line["parity_ok"] is True and line["config"]["dist_backend"] == "gloo"
    par = line["parity"]
    assert all(v["c_cascade_equals_torch_distributed"] and v["rank0_max_rel_deviation"] == 0.0 for v in par["first_contact"].values()) and "g16" in par["first_contact"]
    assert par["timed_frame_strips_vs_untiled_same_gpu"] <= par["limit"] and par["g16_cfg4_geometry"]["ok"] and par["g16_cfg4_geometry"]["hip_vs_reference"] <= 1e-3
    assert "wct_stylize_sharded" in line["config"]["collectives"] and "fast fold" in line["config"]["collectives"] and "style side: owner" in line["config"]["workload"]
    if name == "cfg4":
        assert line["scaling"] == "strong" and line["config"]["content_total"] == "10240x4096"
        assert "halo: exchange" in line["config"]["workload"]
        assert abs(line["value"] - 41.94304 / line["ms_per_step"] * 1e3) < 0.02 * line["value"]
    else:
        assert line["scaling"] == "weak" and line["config"]["content_total"] == "7680x2160"
        s4 = line["passes"]["cfg4_strong"]                       # ONE 10240x4096 frame in 2 strips beside the weak-scaling number
        assert s4["scaling"] == "strong" and s4["MPs"] > 0 and "10240x4096" in s4["workload"] and s4["strips_vs_untiled_same_gpu"] <= 1e-3
    assert line["roofline"]["frac"] > 0


def _n_devices():
    try:
        import torch
        return torch.cuda.device_count()
    except Exception:      # noqa: BLE001
        return 0


needs_two_devices = pytest.mark.skipif(_n_devices() < 2, reason="needs >= 2 MI355X (RCCL carries nothing between ranks sharing one device); "
                                                                "collected everywhere, runs the day a multi-GPU node runs this suite")


@needs_two_devices
@pytest.mark.parametrize("halo_mode,bmap", [("exchange", False), ("recompute", False), ("exchange", True), ("recompute", True)])
def test_two_devices_rccl_matches_untiled_and_in_process(tmp_path, halo_mode, bmap):
    """FIRST CONTACT WITH xGMI (VERDICT r4 task 2): the 2-rank job with one DEVICE per rank over torch.distributed "nccl" (= RCCL) --
    moments all-reduce, style-statistics / (M, b) broadcasts, neighbour exchange or recomputed halos, both map arrangements -- against
    the untiled cascade (the gloo test's tolerance) and BITWISE against the same job on InProcessWorld's device-buffer collectives (a
    two-term sum is commutative: every transport must agree to the last bit)."""
    import torch
    import torch.multiprocessing as mp
    from tools import sharded_standins as standins
    from wct_hip import WCT, model_zoo
    world, H, W = 2, 272, 1525
    out = str(tmp_path / "sh.npz")
    mp.spawn(_shard_worker, args=(world, _free_port(), H, W, halo_mode, bmap, out, (300, 260), "rand", "nccl"), nprocs=world, join=True)
    z = np.load(out)
    assert z["got"].shape == z["ref"].shape == (1, 3, H // 16 * 16, W // 16 * 16)
    assert rel_err(z["got"], z["ref"]) < 5e-4
    w = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    make = lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)     # noqa: E731
    g = torch.Generator(device="cuda").manual_seed(11)
    content = torch.rand((3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 300, 260), device="cuda", generator=g)
    inproc, _ = standins.run_in_process(world, make, content, style, halo_mode=halo_mode, broadcast_map=bmap)
    assert np.array_equal(z["got"], inproc.cpu().numpy())


def test_level_behind_one_c_call_with_rccl_inside_single_rank():
    """SURVEY 8b's "multi-GPU variant takes an ncclComm_t": include/wct_hip.h wct_comm_* + wct_level_sharded -- one level's encoder -> owned-column
    moments -> ncclAllReduce (issued BY THE LIBRARY on the context's stream, RCCL resolved at run time from torch's librccl.so) -> solve ->
    fold -> decoder as ONE call.  This box has one device, so the communicator has one rank (RCCL refuses two ranks on one device): the job is
    the sharded path at world 1 with every call, buffer and collective of the N-rank job.  In a fresh process:
      * the C-collective path is BITWISE the torch.distributed path (wct_content_encode / all_reduce / wct_content_solve / wct_content_decode);
      * it stays within the sharded test's tolerance of the untiled cascade; the range flag arrives (0 clamps);
      * a context without a communicator refuses wct_level_sharded with WCT_ERR_STATE (no silent fallback)."""
    code = r"""
import os, sys, types
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from wct_hip import WCT, model_zoo
from wct_hip.lib import WctError
from wct_hip.sharded import ShardedStylizer
w = model_zoo.load_npz_weights(os.path.join(%r, 'weights', '16x.npz'))
make = lambda: WCT(types.SimpleNamespace(mode='16x', alpha=1.0), weights=w)
g = torch.Generator(device='cuda').manual_seed(11)
H, W = 272, 1525
content = torch.rand((3, H, W), device='cuda', generator=g)
style = torch.rand((3, 300, 260), device='cuda', generator=g)
plain = make()
try:
    plain.level_sharded(5, content, 0, -1, 1.0)
    raise SystemExit('level_sharded without a communicator did not fail')
except (WctError, RuntimeError) as e:
    assert 'RCCL' in str(e) or 'communicator' in str(e), e
ref_sh = ShardedStylizer(plain, dist, H, W, 300, 260, halo_mode='exchange')
assert not ref_sh.c_collectives
want = ref_sh.stylize_strip(content, style)
ref_sh.check_range()
eng = make()
eng.comm_init(dist)
sh = ShardedStylizer(eng, dist, H, W, 300, 260, halo_mode='exchange')
assert sh.c_collectives
got = sh.stylize_strip(content, style)
sh.check_range()
eng.sync()
assert torch.equal(got, want), float((got - want).abs().max())
untiled = make().stylize(content, style)
err = float((got - untiled).abs().max() / untiled.abs().max())
assert tuple(got.shape) == tuple(untiled.shape) and err < 5e-4, err
got2 = sh.stylize_strip(content, style)          # again on the same communicator
assert torch.equal(got2, got)
eng.comm_destroy()
dist.destroy_process_group()
print('CCOLL_OK %%.2e' %% err)
""" % (REPO, PKG, _free_port(), PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0 and "CCOLL_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


@needs_two_devices
@pytest.mark.parametrize("halo_mode", ["exchange", "recompute"])
def test_two_devices_c_collectives_bitwise_equal_torch_distributed(tmp_path, halo_mode):
    """Two devices: the library's own all-reduce (wct_level_sharded over xGMI) against torch.distributed's, same job, bit for bit."""
    import torch.multiprocessing as mp
    outs = []
    for c_coll in (False, True):
        out = str(tmp_path / ("c%d.npz" % c_coll))
        mp.spawn(_shard_worker, args=(2, _free_port(), 272, 1525, halo_mode, False, out, (300, 260), "rand", "nccl", c_coll), nprocs=2, join=True)
        outs.append(np.load(out)["got"])
    assert np.array_equal(outs[0], outs[1])


@needs_two_devices
@pytest.mark.parametrize("halo_mode,bmap,smode", [("exchange", False, "strips"), ("recompute", False, "strips"), ("exchange", True, "owner"), ("recompute", False, "owner")])
def test_two_devices_c_cascade_bitwise_equal_torch_distributed(tmp_path, halo_mode, bmap, smode):
    """Two devices: the whole cascade as ONE library call per rank (wct_stylize_sharded: ncclAllReduce / ncclBroadcast / grouped ncclSend +
    ncclRecv issued by the library over xGMI, after wct_comm_selftest) against wct_hip/sharded.py over torch.distributed, bit for bit."""
    import torch.multiprocessing as mp
    outs = []
    for c_cascade in (False, True):
        out = str(tmp_path / ("c%d.npz" % c_cascade))
        mp.spawn(_shard_worker, args=(2, _free_port(), 272, 1525, halo_mode, bmap, out, (300, 260), "rand", "nccl", False, smode, c_cascade), nprocs=2, join=True)
        outs.append(np.load(out)["got"])
    assert np.array_equal(outs[0], outs[1])


def test_c_cascade_with_rccl_inside_single_rank():
    """wct_stylize_sharded on the library's OWN RCCL table (wct_comm_init: ncclAllReduce / ncclBroadcast / ncclSend / ncclRecv / group calls looked up
    in torch's librccl.so).  One device here, so the communicator has one rank -- in a fresh process:
      * wct_comm_selftest moves known data through all three table functions on it (send-to-self + receive-from-self in one group);
      * the C cascade at world 1 is BITWISE the Python orchestration of the split-level entries and within tolerance of the untiled frame;
      * with WCT_DEBUG set, debug key "shard_emulate" gives the context the GEOMETRY of rank 3 (interior) and rank 0 (edge) of an 8-rank job
        (bench.py passes.cfg4_rank_sim): crops, style strip, five all-reduces and four grouped ncclSend / ncclRecv exchanges really run on
        RCCL with the rank as its own neighbour; the owned strip comes back finite with the owned width; without WCT_DEBUG the key is refused;
      * that emulated frame -- ncclAllReduce, ncclBroadcast, grouped ncclSend / ncclRecv and both lanes' kernels -- is captured into ONE HIP graph
        and replayed on the same buffers with another strip: bitwise the direct call (the call never synchronises the host or allocates)."""
    code = r"""
import os, sys, types
sys.path[:0] = [%r, %r]
import torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
from wct_hip import WCT, model_zoo
from wct_hip.sharded import ShardedStylizer
w = model_zoo.load_npz_weights(os.path.join(%r, 'weights', '16x.npz'))
make = lambda: WCT(types.SimpleNamespace(mode='16x', alpha=1.0), weights=w)
g = torch.Generator(device='cuda').manual_seed(11)
H, W = 272, 1525
content = torch.rand((3, H, W), device='cuda', generator=g)
style = torch.rand((3, 300, 260), device='cuda', generator=g)
want = ShardedStylizer(make(), dist, H, W, 300, 260, halo_mode='exchange').stylize_strip(content, style)
eng = make()
eng.comm_init(dist)
assert eng.comm_info() == (1, 0)
eng.comm_selftest()
sh = ShardedStylizer(eng, dist, H, W, 300, 260, halo_mode='exchange', c_cascade=True)
assert sh.c_cascade and not sh.c_collectives
got = sh.stylize_strip(content, style)
sh.check_range()
eng.sync()
assert torch.equal(got, want), float((got - want).abs().max())
untiled = make().stylize(content, style)
err = float((got - untiled).abs().max() / untiled.abs().max())
assert tuple(got.shape) == tuple(untiled.shape) and err < 5e-4, err
assert torch.equal(sh.stylize_strip(content, style), got)
# one rank of an 8-rank job, its peers itself (measurement geometry): refused without WCT_DEBUG, runs with it
try:
    eng.debug_set('shard_emulate', 803)
    raise SystemExit('shard_emulate accepted without WCT_DEBUG')
except ValueError:
    pass
os.environ['WCT_DEBUG'] = '1'
Hf, Wf = 272, 8 * 320
frame = torch.rand((3, Hf, Wf), device='cuda', generator=g)
style8 = torch.rand((3, 128, 8 * 64), device='cuda', generator=g)
for r in (3, 0, 7):
    eng.debug_set('shard_emulate', 800 + r)
    assert eng.comm_info() == (8, r)
    own0, own1, in0, in1, mode = eng.shard_geometry(Wf, 8, r, 'auto')
    assert mode == 'exchange' and own1 - own0 == 320
    out = eng.stylize_sharded(frame[:, :, in0:in1].contiguous(), style8, Wf, in0, in1, halo_mode='auto', style_mode='strips')
    eng.sync()
    assert tuple(out.shape) == (1, 3, Hf, 320) and bool(torch.isfinite(out).all()), (r, tuple(out.shape))
# ... and the call is capturable into ONE HIP graph, RCCL launches included (no host synchronisation, no allocation after the first call of a size):
# rank 3's frame captured once, replayed on the same buffers with another strip -- bitwise the direct call both times
eng.debug_set('shard_emulate', 803)
own0, own1, in0, in1, mode = eng.shard_geometry(Wf, 8, 3, 'auto')
s1, s2 = frame[:, :, in0:in1].contiguous(), torch.rand((3, Hf, in1 - in0), device='cuda', generator=g)
buf = torch.empty(3 * Hf * 320, device='cuda')
want1 = eng.stylize_sharded(s1, style8, Wf, in0, in1, out=buf).clone()
want2 = eng.stylize_sharded(s2, style8, Wf, in0, in1, out=buf).clone()
strip = s1.clone()
eng.stylize_sharded(strip, style8, Wf, in0, in1, out=buf)
torch.cuda.synchronize()
cap = torch.cuda.Stream()
with torch.cuda.stream(cap):
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph, stream=cap):
        eng.stylize_sharded(strip, style8, Wf, in0, in1, out=buf)
buf.zero_()
graph.replay()
torch.cuda.synchronize()
assert torch.equal(buf.view(-1)[:want1.numel()], want1.view(-1)), 'graph replay 1 differs'
strip.copy_(s2)
graph.replay()
torch.cuda.synchronize()
assert torch.equal(buf.view(-1)[:want2.numel()], want2.view(-1)), 'graph replay 2 (new strip, same graph) differs'
del graph
eng.debug_set('shard_emulate', 0)
assert eng.comm_info() == (1, 0)
eng.comm_destroy()
dist.destroy_process_group()
print('CCASCADE_OK %%.2e' %% err)
""" % (REPO, PKG, _free_port(), PKG)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0").items() if k != "WCT_DEBUG"})
    assert r.returncode == 0 and "CCASCADE_OK" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])


def test_style_strip_margins_reproduce_the_untiled_features():
    """style_mode "strips" relies on: the ENCODER of a style strip fed own +- STYLE_HALO[L] columns (80 / 32 / 12 / 4 / 1: its receptive
    field alone, rounded up to the level's pooling stride) reproduces the untiled relu{L}_1 map BITWISE on the strip's owned feature
    columns -- edge strips, interior strips, a width floor pooling shrinks (525) -- so the strips' raw moments add up to the style's."""
    import torch
    from wct_hip import WCT, model_zoo
    from wct_hip.sharded import STYLE_HALO, ext_bounds, strip_bounds
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
    Hs, Ws, world = 150, 525, 4
    style = torch.rand((1, 3, Hs, Ws), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5))
    for L in (5, 4, 3, 2, 1):
        sh = L - 1
        full = wct.encode(L, style, layout="nhwc")                       # [1, h, w, C]
        n_full, s_full, ss_full = wct.moments(full)
        acc_s, acc_ss, acc_n = 0, 0, 0
        for own in strip_bounds(Ws, world):
            lo, hi = ext_bounds(own, Ws, STYLE_HALO[L])
            f = wct.encode(L, style[..., lo:hi].contiguous(), layout="nhwc")
            f0 = (own[0] - lo) >> sh
            f1 = f.shape[2] if own[1] >= Ws else (own[1] - lo) >> sh
            a = own[0] >> sh
            assert torch.equal(f[:, :, f0:f1], full[:, :, a:a + (f1 - f0)]), (L, own)
            sm, ssm = wct.style_moments(L, style[0, :, :, lo:hi].contiguous(), f0, -1 if own[1] >= Ws else f1)
            acc_s, acc_ss, acc_n = acc_s + sm, acc_ss + ssm, acc_n + f.shape[1] * (f1 - f0)
        assert acc_n == n_full
        assert float((acc_s - s_full).abs().max() / s_full.abs().max()) < 1e-7 and float((acc_ss - ss_full).abs().max() / ss_full.abs().max()) < 1e-7


@needs_two_devices
def test_bench_two_devices_rccl_self_launched():
    """`python bench.py --gpus 2` as the driver types it, on two real devices: self-launched under torch.distributed.run, RCCL."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"],
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "WCT_DIST_BACKEND")},
                       cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["dist_backend"] == "nccl" and line["value"] > 0 and line["scaling"] == "weak"
    assert line["passes"]["cfg4_strong"]["MPs"] > 0


def test_bench_launches_itself_for_n_ranks():
    """`python bench.py --gpus 2` with NO launcher around it and WORLD_SIZE unset (the form the driver uses for --gpus 1; VERDICT r4
    missing #1: it used to die on a usage message): the script re-executes itself under torch.distributed.run --nproc-per-node 2 on
    127.0.0.1 and rank 0 prints the ONE line.  On this box's single device the ranks share it over gloo, and the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "WCT_DIST_BACKEND")}
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--steps-only"],
                       env=env, cwd=REPO, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["config"]["name"] == "cfg2x2" and line["value"] > 0
    assert line["config"]["dist_backend"] == ("nccl" if _n_devices() >= 2 else "gloo")
    assert line["latency_ms_median"] > 0 and "gpu_telemetry" in line


@pytest.mark.parametrize("cfg,name,total", [("cfg3", "cfg3", "1920x1080"), ("cfg2", "cfg2", "3840x2160")])
def test_bench_single_gpu_line_contract(cfg, name, total):
    """bench.py on one GPU, short (--steps-only leaves out the CPU oracle, the parity leg and the extra passes): exactly ONE line on
    stdout -- whatever RCCL or the runtime print goes to stderr -- with the fields the driver reads, for the default configuration
    and for --config cfg3 (--mode original, generated weights)."""
    cmd = [sys.executable, os.path.join(REPO, "bench.py"), "--steps", "2", "--warmup", "1", "--steps-only", "--no-cpu-baseline", "--config", cfg]
    r = subprocess.run(cmd, cwd=REPO, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), r.stdout[-2000:]
    line = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
                "roofline", "cpu_baseline", "latency_ms_median", "latency_ms_min_max", "gpu_telemetry"):
        assert key in line, key
    assert line["n_gpus"] == 1 and line["steps"] == 2 and line["unit"] == "MP/s" and line["value"] > 0 and line["vs_baseline"] is None
    assert line["config"]["name"] == name and line["config"]["content_total"] == total and "workload" in line["config"]
    roof = line["roofline"]
    # per-frame-synchronised latency beside the throughput mean: a lone call is never faster than the pipelined mean by much
    assert line["latency_ms_median"] >= 0.9 * line["ms_per_step"] and line["latency_ms_min_max"][0] <= line["latency_ms_median"] <= line["latency_ms_min_max"][1]
    # the dominant family is the largest time share among ALL families with algorithmic FLOPs (not only conv3x3 names)
    top = max((k for k in line["kernels"] if k["tflops"]), key=lambda k: k["ms_per_step"])
    assert line["roofline"]["kernel"] == top["kernel"]
    assert roof["bound"] in ("mfma", "hbm") and 0 < roof["frac"] < 1 and roof["achieved"] > 0 and roof["peak"] > 0 and "traffic" in roof
    assert abs(line["value"] - float(total.split("x")[0]) * float(total.split("x")[1]) / 1e6 / line["ms_per_step"] * 1e3) < 0.02 * line["value"]


def test_strip_halos_exact_with_level_margins(tmp_path):
    """halo_mode "exchange" relies on: a strip fed own +- LEVEL_HALO[L] columns reproduces the untiled level BITWISE on its
    owned columns (same (M, b)) -- edge strips, an interior strip, and a width that floor pooling shrinks (2005 -> 2000)."""
    import torch
    from wct_hip import WCT, model_zoo
    from wct_hip.sharded import LEVEL_HALO, ext_bounds, strip_bounds
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz")))
    H, W, world = 176, 2005, 3
    g = torch.Generator(device="cuda").manual_seed(3)
    content = torch.rand((1, 3, H, W), device="cuda", generator=g)
    style = torch.rand((3, 200, 180), device="cuda", generator=g)
    for L in (5, 4, 3, 2, 1):
        sh = L - 1
        sF = wct.encode(L, style, layout="nhwc")
        cF = wct.encode(L, content, layout="nhwc")
        nc, sc, ssc = wct.moments(cF)
        M, b = wct.solve(nc, sc, ssc, *wct.moments(sF))
        full = wct.decode_affine(L, cF, M, b)
        Wn = (W >> sh) << sh
        for own in strip_bounds(W, world):
            lo, hi = ext_bounds(own, W, LEVEL_HALO[L])
            f = wct.encode(L, content[..., lo:hi].contiguous(), layout="nhwc")
            o = wct.decode_affine(L, f, M, b)
            a, e = own[0], min(Wn, own[1])
            assert torch.equal(o[..., a - lo:e - lo], full[..., a:e]), (L, own)


def test_bench_collects_hbm_traffic_live():
    """roofline.traffic of the default bench run: two rocprofv3 --pmc child runs (FETCH_SIZE, WRITE_SIZE) of the bench command,
    summarised per launch -- from this build's kernels (source ids equal), and close to the dominant kernel's algorithmic bytes
    (SP16 activations are read once apart from tile halos, written once)."""
    sys.path.insert(0, REPO)
    import bench
    live = bench.live_pmc("cfg2")
    if not live:      # no rocprofv3 here, or counters not collectable in this environment (e.g. already under a profiler):
        pytest.skip("rocprofv3 PMC passes not available here; bench.py then reports the committed profile and says so")
    assert os.path.exists(live)
    traffic, src = bench.pmc_traffic("conv3x3_f16x3<co=64,dma>", live)
    assert src["stale"] is False and src["file"].startswith("live")
    # 19 launches per step: 64 -> 64 at 960x540 / 1024x1024 and 128 -> 64, 64 -> 64 behind an upsample ...: 100 .. 400 MB each
    assert 5e7 < traffic < 6e8, traffic
