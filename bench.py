#!/usr/bin/env python3
"""Benchmark of the MI355X-native WCT stylisation path.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: launched by torch.distributed.run)

A "step" is one end-to-end 5-level WCT stylisation (levels 5..1; style-side encodes, moments and
eigensolves INCLUDED) of BASELINE.json configs[1]: `--mode 16x`, 3840x2160 content, 2048x2048 style, synthetic
uniform-noise images already resident in HBM (fp32, planar 3xHxW).  With N > 1 the content is N times wider
(3840*N x 2160) and column-sharded: every rank stylises its own 3840-wide strip (+ a cumulative halo, so no
neighbour exchange), the only exchanges being one RCCL all-reduce of the fp64 content moments and one broadcast
of the colouring map (M, b) per level (wct_hip/sharded.py) -> weak scaling.  value = content megapixels / second over all ranks.

The JSON line also carries
  roofline      dominant kernel family: algorithmic FLOP per launch / HIP-event duration vs the gfx950 fp32-MFMA peak
  passes        relu4_1 encode pass: algorithmic GB/s (364 B/px) and TFLOP/s (30 816 FLOP/px); the content
                cascade against cached style statistics (reported separately, never `value`)
  cpu_baseline  the CPU oracle (oracle/: numpy + C/OpenMP port of the reference's op sequence) timed on the host
                cores on a bounded sample (1920x1080 content + 1024x1024 style, 5 levels); rank 0, N = 1 only.
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

H, W, HS, WS = 2160, 3840, 2048, 2048
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
    if dma:     # <CT, POOL, OUTF32, GROUPS>: both output formats of the family
        pre = "void conv3x3_sp_kernel<%d, %s, " % (1 if co == 32 else 2, tf(pool))
        rows = [v for k, v in ks.items() if k.startswith(pre) and k.endswith(", %s>(SpArgs)" % tf(co >= 128))]
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


def cpu_baseline(weights):
    """The oracle (a CPU port of the reference's op sequence: fp32 convs, fp64 two-GEMM WCT with SVD) on a bounded
    sample of the same workload.  This is the ONLY place bench.py touches oracle/.
    Threads: the C/OpenMP convolutions stop scaling at ~8-32 threads on the GPU box's host (1.3 s at 8..32 threads,
    3.8 s at 128, 23.6 s at 256 for a 512x512 sample: oversubscription), so min(cores, 32) are used and reported."""
    from oracle import wct_oracle
    threads = min(os.cpu_count() or 1, 32)
    wct_oracle.set_num_threads(threads)
    mods = wct_oracle.Modules("16x", weights)
    rng = np.random.default_rng(0)
    hc, wc, hs, ws = 1080, 1920, 1024, 1024     # a quarter of the benchmark's content / style pixels
    c = rng.random((3, hc, wc), dtype=np.float32)
    s = rng.random((3, hs, ws), dtype=np.float32)
    t0 = time.perf_counter()
    out = wct_oracle.stylize(mods, c, s, 1.0)
    dt = time.perf_counter() - t0
    assert np.isfinite(out).all()
    return {"value": round(hc * wc / 1e6 / dt, 5), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "5-level 16x WCT, %dx%d content + %dx%d style (uniform noise), %.1f s wall, %d OpenMP threads for the "
                      "convolutions (of %d host cores), numpy/OpenBLAS for the fp64 transform" % (wc, hc, ws, hs, dt, wct_oracle.num_threads(), os.cpu_count() or 1)}, out, (c, s)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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

    def strip_image(r):   # strip r of the virtual (W*world) x H content: uniform noise, no zeros, seeded per strip
        g = torch.Generator(device="cuda").manual_seed(1 + r)
        return torch.rand((3, H, W), device="cuda", generator=g)

    g2 = torch.Generator(device="cuda").manual_seed(2)
    style = torch.rand((3, HS, WS), device="cuda", generator=g2)          # same style on every rank

    if world > 1:
        from wct_hip.sharded import ShardedStylizer
        runner = ShardedStylizer(wct, dist, H, W * world, HS, WS)
        x0, x1 = runner.input_columns()                                    # own strip + cumulative halo
        parts = []
        for r in range(world):
            a, b = max(x0, r * W), min(x1, (r + 1) * W)
            if a < b:
                parts.append(strip_image(r)[:, :, a - r * W:b - r * W])
        content = torch.cat(parts, dim=2).contiguous()
        del parts
        step = lambda: runner.stylize_strip(content, style)   # noqa: E731
    else:
        content = strip_image(0)
        wct.reserve(H, W, HS, WS)
        out = torch.empty((3, H, W), device="cuda")
        step = lambda: wct.stylize(content, style, out=out)   # noqa: E731

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        res = step()
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device="cuda", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert bool(torch.isfinite(res).all())
    mp = H * W * world / 1e6
    value = mp * args.steps / dt

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
        roof["traffic"] = pmc_traffic(d["name"])
        roof.update({"avg_launch_ms": round(d["ms"] / d["launches"], 4), "share_of_kernel_time": round(d["ms"] / tot, 3),
                     "algo_flop_per_launch": d["flops"] / d["launches"], "algo_bytes_per_launch": d["bytes"] / d["launches"]})
    if rank == 0 and not args.steps_only:
        # relu4_1 encode pass on the 4K content (north_star's named pass)
        content4k = content[:, :, :W].contiguous()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            wct.encode(4, content4k, layout="nhwc")
        e0.record()
        for _ in range(5):
            wct.encode(4, content4k, layout="nhwc")
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        # the cascade against cached style statistics ("style cached" reported separately; N = 1 only)
        cached = None
        if world == 1:
            wct.style_prepare(style)
            for _ in range(2):
                wct.stylize_prepared(content4k)
            e0.record()
            for _ in range(5):
                wct.stylize_prepared(content4k)
            e1.record()
            torch.cuda.synchronize()
            msc = e0.elapsed_time(e1) / 5
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
        passes = {"style_cached_cascade": cached, "relu4_1_encode": {"ms": round(ms, 3), "algo_GBs": round(364.0 * H * W / ms / 1e6, 1),
                                     "frac_hbm_8TBs": round(364.0 * H * W / ms / 1e6 / PEAK_HBM_GBS, 4),
                                     "tflops": round(30816.0 * H * W / ms / 1e9, 2),
                                     "frac_f32_mfma": round(30816.0 * H * W / ms / 1e9 / PEAK_F32_MFMA_TF, 4)}}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu, ref, (c_np, s_np) = cpu_baseline(weights)
        got = wct.stylize(torch.from_numpy(c_np).cuda(), torch.from_numpy(s_np).cuda()).cpu().numpy()[0]
        cpu["gpu_vs_oracle_rel_err"] = float(np.abs(got - ref).max() / np.abs(ref).max())   # parity gate of the timed path

    if rank == 0:
        line = {
            "metric": "content megapixels/sec, end-to-end 5-level WCT (16x VGG, 4K content, 2K style)",
            "value": round(value, 2), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PytorchWCT/WCT.py --mode 16x, 5-level WCT, %dx%d content per GPU / %dx%d style, alpha=1, "
                                   "style-side work included, images resident in HBM" % (W, H, WS, HS),
                       "content_total": "%dx%d" % (W * world, H), "parallelism": "content column strips x%d" % world, "dist_backend": (backend if world > 1 else None)},
            "roofline": roof, "passes": passes, "cpu_baseline": cpu, "kernels": profile,
        }
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
