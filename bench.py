#!/usr/bin/env python3
"""Benchmark of the MI355X-native WCT stylisation path.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A "step" is one end-to-end 5-level WCT stylisation (levels 5..1; style-side encodes, moments and
eigensolves INCLUDED) of synthetic uniform-noise images already resident in HBM (fp32, planar 3xHxW).

  --config cfg2 (default)   BASELINE.json configs[1]: `--mode 16x`, 3840x2160 content, 2048x2048 style.  With N > 1 the content
                            is N times wider (3840*N x 2160) and column-sharded, every rank a 3840-wide strip -> WEAK scaling
                            ("cfg2xN"); the line then also carries passes.cfg4_strong, ONE 10240x4096 frame in N strips.
  --config cfg4             BASELINE.json configs[3]: ONE 10240x4096 content (2048x2048 style) in N column strips -> STRONG
                            scaling (N = 1: the north_star's single-GPU target frame, untiled).
Sharding (wct_hip/sharded.py): per level one RCCL all-reduce of the fp64 content moments, one broadcast of the level's style
statistics (or of the colouring map (M, b)), and -- for strips narrower than 2560 columns -- a neighbour exchange of the
decoded edge columns instead of recomputed cumulative halos.  value = content megapixels / second over all ranks.

The JSON line also carries
  roofline      dominant kernel family: algorithmic FLOP per launch / HIP-event duration vs the gfx950 fp32-MFMA peak
  passes        relu4_1 encode pass: algorithmic GB/s (364 B/px) and TFLOP/s (30 816 FLOP/px); the content
                cascade against cached style statistics (reported separately, never `value`)
  cpu_baseline  the CPU oracle (oracle/: numpy + C/OpenMP port of the reference's op sequence) timed on the host
                cores on ONE frame of the timed configuration itself (3840x2160 + 2048x2048, ~25 s on 32 threads);
                rank 0, N = 1 only.  The same frame is the parity gate of the timed path (`parity`): the timed call's
                output against the oracle's fp64 arm ("truth": the reference's algorithm in exact arithmetic) and
                against the oracle itself.  Two valid fp32 implementations of the reference differ by ~1e-3 end to end on
                uniform noise (tests/test_hip_scale.py), so the gate is  |hip - truth| <= 1e-3  and
                |hip - truth| <= 1.5 |oracle - truth| + 1e-4;  when it fails (or an activation left the f16x3 range)
                the line says "parity_ok": false, `value` is null and the exit status is 1.
"""
import argparse
import json
import os
import sys
import time
import types

REPO = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(REPO, "collaborative-distillation_amd")
for p in (REPO, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402
import torch  # noqa: E402

H, W, HS, WS = 2160, 3840, 2048, 2048      # BASELINE configs[1]
H4, W4 = 4096, 10240                        # BASELINE configs[3]: content of the north_star's target frame
GATE = 1e-3                                 # north_star: max|d| / max|ref| end to end
PEAK_F32_MFMA_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0      # spec; 6290 measured copy
PEAK_F16_MFMA_TF = 2500.0  # dense f16/bf16 MFMA (the f16x3 kernels issue 3 MFMAs per algorithmic product)


def pmc_traffic(family):
    """HBM bytes per launch of the kernel behind a profile family, from the committed rocprofv3 PMC passes
    (profiles/hbm_traffic_latest.json, made by tools/pmc_summary.py; counters cannot be read from inside this process)."""
    import re
    path = os.path.join(REPO, "profiles", "hbm_traffic_latest.json")
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))["kernels"]
    m = re.match(r"conv3x3_f16x3<co=(\d+)(,pool)?(,out3)?(,dma)?>", family)
    if not m:
        return None
    co, pool, out3, dma = int(m.group(1)), bool(m.group(2)), bool(m.group(3)), bool(m.group(4))
    tf = lambda b: "true" if b else "false"
    if dma:     # <CT, POOL, OUTF32, GROUPS[, DEEP]>: both output formats (and both pipeline depths) of the family
        rx = re.compile(r"void conv3x3_sp_kernel<(\d), (true|false), (true|false), (true|false)(?:, (true|false))?>\(SpArgs\)")
        rows = []
        for k, v in ks.items():
            mm = rx.match(k)
            if mm and int(mm.group(1)) == (1 if co == 32 else 2) and mm.group(2) == tf(pool) and mm.group(4) == tf(co >= 128):
                rows.append(v)
        if not rows:
            return None
        n = sum(r["calls"] for r in rows)
        return {"hbm_bytes_per_launch": round(sum((r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["calls"] for r in rows) / n * 1e6),
                "source": "profiles/hbm_traffic_latest.json", "pmc_avg_launch_us": round(sum(r["avg_us"] * r["calls"] for r in rows) / n, 2)}
    if co == 16:
        name = "void conv3x3_f16_c16_kernel<%s, %s>(F16Args)" % (tf(pool), tf(out3))
    else:
        name = "void conv3x3_f16_kernel<%d, %s, %d>(F16Args)" % (min(co, 128) // 32, tf(pool), 16 if co >= 128 else 8)
    e = ks.get(name)
    if not e:
        return None
    return {"hbm_bytes_per_launch": round((e["read_MB_per_launch"] + e["write_MB_per_launch"]) * 1e6), "source": "profiles/hbm_traffic_latest.json",
            "pmc_avg_launch_us": e["avg_us"]}


def cpu_baseline(weights, content, style):
    """The oracle (a CPU port of the reference's op sequence: fp32 convs, fp64 two-GEMM WCT with SVD) on a bounded
    sample of the same workload.  This is the ONLY place bench.py touches oracle/.
    Threads: the C/OpenMP convolutions stop scaling at ~8-32 threads on the GPU box's host (1.3 s at 8..32 threads,
    3.8 s at 128, 23.6 s at 256 for a 512x512 sample: oversubscription), so min(cores, 32) are used and reported."""
    from oracle import wct_oracle
    threads = min(os.cpu_count() or 1, 32)
    wct_oracle.set_num_threads(threads)
    mods = wct_oracle.Modules("16x", weights)
    hc, wc, hs, ws = H, W, HS, WS               # ONE frame of the timed configuration itself
    c, s = content, style
    t0 = time.perf_counter()
    out = wct_oracle.stylize(mods, c, s, 1.0)
    dt = time.perf_counter() - t0
    assert np.isfinite(out).all()
    # second arm ("best-effort CPU"): the same convolutions, the transform as ONE fp32 affine map
    # csF = M cF + b (M, b from the fp64 C x C statistics) instead of the reference's two fp64 C x C . C x hw GEMMs + elementwise
    # passes -- what a CPU implementer free to restructure would do.  Quarter-size sample to bound the run.
    rng = np.random.default_rng(0)
    cq, sq = rng.random((3, 1080, 1920), dtype=np.float32), rng.random((3, 1024, 1024), dtype=np.float32)
    t1 = time.perf_counter()
    outq = wct_oracle.stylize(mods, cq, sq, 1.0, fused_affine=True)
    dq = time.perf_counter() - t1
    assert np.isfinite(outq).all()
    t2 = time.perf_counter()
    truth = wct_oracle.stylize(wct_oracle.Modules("16x", weights, precision="fp64"), c, s, 1.0)   # parity yardstick, not a baseline
    dtruth = time.perf_counter() - t2
    return {"value": round(hc * wc / 1e6 / dt, 5), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "ONE frame of the timed configuration: 5-level 16x WCT, %dx%d content + %dx%d style (uniform noise), %.1f s "
                      "wall, %d OpenMP threads for the convolutions (of %d host cores), numpy/OpenBLAS for the fp64 transform"
                      % (wc, hc, ws, hs, dt, wct_oracle.num_threads(), os.cpu_count() or 1),
            "best_effort": {"value": round(1080 * 1920 / 1e6 / dq, 5), "unit": "MP/s", "cores": threads,
                            "sample": "1920x1080 content + 1024x1024 style, transform as one fp32 affine map (fused), %.1f s wall" % dq},
            "truth_arm_s": round(dtruth, 1)}, out, truth


def uhd_pair_parity(wct, weights):
    """green_park-wallpaper-3840x2160.jpg + style/in1.jpg (2048x2048), the reference's sample data at config-2 size:
    this library vs the oracle, and both vs the reference's own pixels (tests/golden/g11_uhd_pair.npz).  None when the fixtures
    or Pillow are missing."""
    gold = os.path.join(REPO, "tests", "golden")
    files = [os.path.join(gold, f) for f in ("g11_uhd_content_3840x2160.jpg", "g11_style_2048x2048.jpg", "g11_uhd_pair.npz")]
    try:
        from PIL import Image
    except ImportError:
        return None
    if not all(os.path.exists(f) for f in files):
        return None
    from oracle import wct_oracle
    c_u8, s_u8 = (np.array(Image.open(f).convert("RGB")) for f in files[:2])
    g = np.load(files[2])
    ref = wct_oracle.stylize(wct_oracle.Modules("16x", weights), wct_oracle.to_tensor_u8(c_u8), wct_oracle.to_tensor_u8(s_u8), 1.0)
    got = wct.stylize(wct.to_tensor_u8(torch.from_numpy(c_u8)), wct.to_tensor_u8(torch.from_numpy(s_u8))).cpu().numpy()[0]
    mx = float(g["max"])
    e = float(np.abs(got - ref).max() / np.abs(ref).max())
    crops = []
    for i in range(4):
        y0, x0 = (int(v) for v in g["crop%d.origin" % i])
        crops.append(float(np.abs(got[:, y0:y0 + 96, x0:x0 + 96].astype(np.float64) - g["crop%d" % i]).max() / mx))
    return {"hip_vs_oracle": e, "hip_vs_reference_pixels": max(crops), "gate": GATE, "ok": bool(e <= GATE and max(crops) <= GATE),
            "frame": "reference sample data: UHD_content/green_park 3840x2160 + style/in1.jpg 2048x2048"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["cfg2", "cfg4"], default="cfg2",
                    help="cfg2: 3840x2160 content per GPU (N > 1: weak scaling over an N x 3840 wide frame); "
                         "cfg4: ONE 10240x4096 content in N column strips (strong scaling)")
    ap.add_argument("--halo-mode", choices=["auto", "recompute", "exchange"], default="auto")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="wct_debug_set switches for A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle run (and with it the parity gate)")
    ap.add_argument("--steps-only", action="store_true",
                    help="skip the extra passes (relu4_1 encode, cached style, frames in flight): every launch then belongs to a "
                         "stylise step, so a rocprofv3 --stats summary of the run averages the same launch mix as `roofline`")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py --gpus %d"
                         % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs the MI355X (no CPU path)"
    # one rank per GPU; WCT_DIST_BACKEND=gloo lets several ranks share one GPU (used only to smoke-test the N > 1 code
    # path on a single-GPU box -- RCCL needs one device per rank)
    backend = os.environ.get("WCT_DIST_BACKEND", "nccl")
    dev = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)

    from wct_hip import WCT, model_zoo
    weights = model_zoo.load_npz_weights(os.path.join(PKG, "weights", "16x.npz"))
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights)
    for kv in args.debug_set:
        wct.debug_set(kv.split("=")[0], float(kv.split("=")[1]))

    def noise(seed, h, w):   # uniform noise, no zeros (zero-filled inputs clock higher), seeded
        return torch.rand((3, h, w), device="cuda", generator=torch.Generator(device="cuda").manual_seed(seed))

    style = noise(2, HS, WS)                                               # the same style on every rank

    def frame_columns(x0, x1, cfg):
        """Columns [x0, x1) of the benchmark's (virtual) content frame.  cfg2: N seeded 3840-wide panels side by side (panel
        0 = the N = 1 frame); cfg4: one seeded 10240-wide frame, generated in 1280-column panels so that a rank only
        materialises what it needs."""
        pw, ph, seed0 = (W, H, 1) if cfg == "cfg2" else (1280, H4, 500)
        parts = []
        for r in range(x0 // pw, (x1 - 1) // pw + 1):
            a, b = max(x0, r * pw), min(x1, (r + 1) * pw)
            parts.append(noise(seed0 + r, ph, pw)[:, :, a - r * pw:b - r * pw])
        return torch.cat(parts, dim=2).contiguous()

    def make_step(cfg):
        """-> (step(), megapixels of the whole frame, description)"""
        fh, fw = (H, W * world) if cfg == "cfg2" else (H4, W4)
        if world > 1:
            from wct_hip.sharded import ShardedStylizer
            runner = ShardedStylizer(wct, dist, fh, fw, HS, WS, halo_mode=args.halo_mode)
            x0, x1 = runner.input_columns()                                # own strip + halo of the halo mode
            content = frame_columns(x0, x1, cfg)
            return (lambda: runner.stylize_strip(content, style)), fh * fw / 1e6, \
                "%dx%d content in %d column strips (halo: %s), %dx%d style" % (fw, fh, world, runner.halo_mode, WS, HS), content
        content = frame_columns(0, fw, cfg)
        wct.reserve(fh, fw, HS, WS)
        out = torch.empty((3, fh, fw), device="cuda")
        return (lambda: wct.stylize(content, style, out=out)), fh * fw / 1e6, "%dx%d content, %dx%d style" % (fw, fh, WS, HS), content

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step, steps, warmup):
        for _ in range(warmup):
            step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            res = step()
        barrier()
        dt = time.perf_counter() - t0
        if dist is not None:
            tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        assert bool(torch.isfinite(res).all())
        return dt

    step, mp, desc, content = make_step(args.config)
    wct.saturation_count(reset=True)
    dt = timed(step, args.steps, args.warmup)
    value = mp * args.steps / dt
    saturated = wct.saturation_count()      # threads that clamped an activation to the f16x3 range during the timed steps

    # ---- roofline leg (rank 0): HIP events around every kernel launch on the context's stream
    roof, passes, profile = None, None, None
    nprof = 2
    if rank == 0:
        wct.set_overlap(False)   # kernels one at a time, so that an event pair times exactly one launch
        wct.profile_reset()
        wct.profile(True)
    for _ in range(nprof):      # every rank runs these steps (they contain collectives); only rank 0 records events
        step()
    barrier()
    if rank == 0:
        wct.profile(False)
        wct.set_overlap(True)
        ents = sorted(wct.profile_read(), key=lambda e: -e["ms"])
        tot = sum(e["ms"] for e in ents)
        profile = [{"kernel": e["name"], "ms_per_step": round(e["ms"] / nprof, 4), "launches_per_step": e["launches"] // nprof,
                    "tflops": round(e["flops"] / e["ms"] / 1e9, 2) if e["flops"] else None,
                    "algo_GBs": round(e["bytes"] / e["ms"] / 1e6, 1) if e["bytes"] else None} for e in ents]
        # dominant kernel FAMILY with algorithmic work attached; its binding roofline is the larger of the two fractions
        convs = [e for e in ents if e["flops"] > 0 and e["name"].startswith("conv3x3")]
        d = convs[0]
        f16 = "f16x3" in d["name"]
        peak_tf = PEAK_F16_MFMA_TF / 3.0 if f16 else PEAK_F32_MFMA_TF
        tf, gbs = d["flops"] / d["ms"] / 1e9, d["bytes"] / d["ms"] / 1e6
        if gbs / PEAK_HBM_GBS >= tf / peak_tf:
            roof = {"kernel": d["name"], "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
        else:
            roof = {"kernel": d["name"], "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                    "frac": round(tf / peak_tf, 4), "traffic": None,
                    "peak_note": "2.5 PF dense f16 MFMA / 3 split terms" if f16 else "fp32 MFMA"}
        pmc = pmc_traffic(d["name"])     # HBM bytes per launch from the committed rocprofv3 PMC passes (a number, or null)
        roof["traffic"] = pmc["hbm_bytes_per_launch"] if pmc else None
        roof["traffic_source"] = ({"file": pmc["source"], "pmc_avg_launch_us": pmc["pmc_avg_launch_us"]} if pmc else None)
        roof.update({"avg_launch_ms": round(d["ms"] / d["launches"], 4), "share_of_kernel_time": round(d["ms"] / tot, 3),
                     "algo_flop_per_launch": d["flops"] / d["launches"], "algo_bytes_per_launch": d["bytes"] / d["launches"]})
    del step

    # ---- extra passes (never `value`)
    extra = not args.steps_only
    passes = {}
    if extra and world > 1 and args.config == "cfg2":
        # BASELINE configs[3] beside the weak-scaling number: ONE 10240x4096 frame in N strips (strong scaling)
        del content
        step4, mp4, desc4, content4 = make_step("cfg4")
        k4 = max(3, args.steps // 2)
        dt4 = timed(step4, k4, 2)
        passes["cfg4_strong"] = {"workload": desc4, "ms_per_frame": round(dt4 / k4 * 1e3, 3), "MPs": round(mp4 * k4 / dt4, 1), "scaling": "strong"}
        del step4, content4
    if extra and rank == 0 and world == 1:
        content4k = content if args.config == "cfg2" else frame_columns(0, W, "cfg2")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def ev_ms(fn, n=5, warm=2):
            for _ in range(warm):
                fn()
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / n

        # relu4_1 encode pass on the 4K content (north_star's named pass)
        ms = ev_ms(lambda: wct.encode(4, content4k, layout="nhwc"))
        passes["relu4_1_encode"] = {"ms": round(ms, 3), "algo_GBs": round(364.0 * H * W / ms / 1e6, 1),
                                    "frac_hbm_8TBs": round(364.0 * H * W / ms / 1e6 / PEAK_HBM_GBS, 4),
                                    "tflops": round(30816.0 * H * W / ms / 1e9, 2),
                                    "frac_f16x3_mfma_833TF": round(30816.0 * H * W / ms / 1e9 / (PEAK_F16_MFMA_TF / 3.0), 4)}
        # the cascade against cached style statistics ("style cached" reported separately)
        wct.style_prepare(style)
        msc = ev_ms(lambda: wct.stylize_prepared(content4k))
        cached = {"ms": round(msc, 3), "MPs": round(H * W / 1e6 / msc * 1e3, 1)}
        # the same with three frames in flight on this GPU (wct_hip/pipeline.py; throughput of the batch / video case)
        from wct_hip.pipeline import FramePipeline
        pipe = FramePipeline(lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights), slots=3)
        pipe.set_style(style)
        frames = [content4k] * 12
        pipe.stylize_many(frames[:6])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.stylize_many(frames)
        torch.cuda.synchronize()
        msp = (time.perf_counter() - t0) / len(frames) * 1e3
        cached["three_frames_in_flight"] = {"ms_per_frame": round(msp, 3), "MPs": round(H * W / 1e6 / msp * 1e3, 1)}
        del pipe
        passes["style_cached_cascade"] = cached
        if args.config == "cfg2":
            # BASELINE configs[3]'s frame, 10240x4096, untiled on this ONE GPU (the north_star's target configuration)
            c4 = frame_columns(0, W4, "cfg4")
            o4 = torch.empty((3, H4, W4), device="cuda")
            ms4 = ev_ms(lambda: wct.stylize(c4, style, out=o4), n=3, warm=1)
            passes["cfg4_single_gpu"] = {"workload": "10240x4096 content, 2048x2048 style, untiled on one GPU", "ms_per_frame": round(ms4, 3),
                                         "MPs": round(H4 * W4 / 1e6 / ms4 * 1e3, 1), "finite": bool(torch.isfinite(o4).all())}
            del c4, o4

    cpu, parity, parity_ok = None, None, None
    if rank == 0 and world == 1 and args.config == "cfg2" and not args.no_cpu_baseline:
        c_np, s_np = content.cpu().numpy(), style.cpu().numpy()
        cpu, ref, truth = cpu_baseline(weights, c_np, s_np)
        got = wct.stylize(content, style).cpu().numpy()[0]     # the timed call on the timed inputs
        rel = lambda a, b: float(np.abs(a - b).max() / np.abs(b).max())   # noqa: E731
        e_truth, o_truth, e_ref = rel(got, truth), rel(ref, truth), rel(got, ref)
        parity = {"hip_vs_truth": e_truth, "oracle_vs_truth": o_truth, "hip_vs_oracle": e_ref, "gate": GATE,
                  "margin": round(1.0 - e_truth / GATE, 3), "frame": "the timed frame (%dx%d + %dx%d)" % (W, H, WS, HS),
                  "truth": "oracle/ with precision fp64: the reference's op sequence, every activation and accumulation in fp64",
                  "allclose_truth_rtol_atol_1e-3": bool(np.allclose(got, truth, rtol=GATE, atol=GATE * float(np.abs(truth).max()))),
                  "f16x3_saturated_threads": int(saturated)}
        parity_ok = bool(e_truth <= 1.5 * o_truth + 1e-4 and e_truth <= 2 * GATE and saturated == 0)
        parity["within_1e-3_of_truth"] = bool(e_truth <= GATE)
        # the reference's own UHD sample pair at the same size (tests/golden/g11_*: data files of the reference + the
        # reference's own output pixels): natural images leave the north_star gate 10x of room, so here it is absolute
        up = uhd_pair_parity(wct, weights)
        if up is not None:
            parity["reference_uhd_pair"] = up
            parity_ok = parity_ok and up["ok"]
    elif saturated:
        parity_ok = False
        parity = {"f16x3_saturated_threads": int(saturated)}

    if rank == 0:
        cfg2 = args.config == "cfg2"
        line = {
            "metric": "content megapixels/sec, end-to-end 5-level WCT (16x VGG, %s, 2K style)" % ("4K content" if cfg2 else "10240x4096 content"),
            "value": (round(value, 2) if parity_ok is not False else None), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak" if cfg2 else "strong",
            "vs_baseline": None, "dtype": "f32 (f16x3 split-MFMA products, fp32 accumulate)", "data": "synthetic",
            "parity_ok": parity_ok, "parity": parity,
            "config": {"workload": "PytorchWCT/WCT.py --mode 16x, 5-level WCT, %s, alpha=1, style-side work included, images resident in HBM; %s"
                                   % (desc, ("BASELINE configs[1]" + ("" if world == 1 else " x%d wide (weak scaling)" % world)) if cfg2 else "BASELINE configs[3] (strong scaling)"),
                       "name": ("cfg2" if world == 1 else "cfg2x%d" % world) if cfg2 else "cfg4",
                       "content_total": "%dx%d" % ((W * world, H) if cfg2 else (W4, H4)), "parallelism": "content column strips x%d" % world,
                       "dist_backend": (backend if world > 1 else None)},
            "roofline": roof, "passes": passes or None, "cpu_baseline": cpu, "kernels": profile,
        }
        if parity_ok is False:
            line["unverified_value"] = round(value, 2)
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()
    if parity_ok is False:
        sys.exit(1)


if __name__ == "__main__":
    main()
