#!/usr/bin/env python3
"""Benchmark of the MI355X-native WCT stylisation path.

    python bench.py [--gpus N] [--steps K] [--warmup W]          (N > 1: under torch.distributed.run, or plainly -- it then
                                                                  launches itself that way, one process per GPU)

A "step" is one end-to-end 5-level WCT stylisation (levels 5..1; style-side encodes, moments and matrix functions INCLUDED) of
synthetic images already resident in HBM (fp32, planar 3xHxW).  The frames are the SURVEY 8(d) ones: uniform noise from
numpy.random.default_rng (content seed 1, style seed 2; config 3: seeds 3 / 4; config 4: seed 5), generated on the host and
uploaded before the timed region -- the build container can regenerate them bit for bit, which is how the reference itself was
run on the timed frame (tests/golden/g13_cfg2_noise.npz, tools/make_goldens.py gen_g13).

  --config cfg2 (default)   BASELINE.json configs[1]: `--mode 16x`, 3840x2160 content, 2048x2048 style.  With N > 1 the content
                            is N times wider (3840*N x 2160) and column-sharded, every rank a 3840-wide strip -> WEAK scaling
                            ("cfg2xN"); the line then also carries passes.cfg4_strong, ONE 10240x4096 frame in N strips.
  --config cfg4             BASELINE.json configs[3]: ONE 10240x4096 content (2048x2048 style) in N column strips -> STRONG
                            scaling (N = 1: the north_star's single-GPU target frame, untiled).
  --config cfg3             BASELINE.json configs[2]: `--mode original` (un-pruned VGG-19 graph, GENERATED weights: the torch7
                            checkpoints are absent, real-weight parity unpinned), 1920x1080 content and style; N = 1 only.
                            The default run reports the same thing as passes.cfg3_original; this switch makes it the timed
                            step so that tools/profile_round.sh can profile it.
Sharding (wct_hip/sharded.py): per level one RCCL all-reduce of the fp64 content moments, one broadcast of the level's style
statistics (or of the colouring map (M, b)), and -- for strips narrower than 2560 columns -- a neighbour exchange of the
decoded edge columns instead of recomputed cumulative halos.  value = content megapixels / second over all ranks.

The JSON line also carries
  roofline      dominant kernel family: algorithmic FLOP per launch / HIP-event duration vs the f16x3 MFMA roofline
  passes        relu4_1 encode pass (364 B/px, 30 816 FLOP/px); the cascade against cached style statistics and
                with three frames in flight (cfg5_per_gpu); configs[3]'s frame untiled on one GPU; configs[2] (cfg3_original);
                the uint8 image edge end to end over PCIe (u8_end_to_end: the reference's timer, WCT.py:118-131, includes
                save_image); one rank's share of the 8-GPU config-4 job timed on this GPU (cfg4_rank_sim) -- never `value`
  cpu_baseline  the CPU oracle (oracle/: numpy + C/OpenMP port of the reference's op sequence) timed on the host cores on ONE
                frame of the timed configuration itself; rank 0, N = 1 only
  parity        THE GATE (the same rule in tests/test_hip_scale.py and DESIGN.md 2): the timed call's output on
                the timed frame against the REFERENCE'S OWN PIXELS for that frame (G13: util_wct.WCT on torch CPU, every pixel
                the fixture holds = 1/16 lattice + 8 crops):
                    hip_vs_reference <= max(1e-3, 1.25 * oracle_vs_reference)
                i.e. the north_star's 1e-3 wherever the reference's arithmetic is itself reproducible to 1e-3 by a second valid
                fp32 implementation of it (the oracle: the same op sequence in C loops), and otherwise no further from the
                reference than that implementation is (+25 %).  Plus: the reference's UHD sample pair (G11) <= 1e-3 outright,
                and no f16x3 saturation during the timed steps.  A failing gate prints "value": null and exits 1.
"""
import argparse
import hashlib
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

from tests.fixture_compare import GATE, compare_to_fixture, noise_frame  # noqa: E402  (plain numpy helpers + the seeds)

H, W, HS, WS = 2160, 3840, 2048, 2048      # BASELINE configs[1]
H3, W3 = 1080, 1920                         # BASELINE configs[2]
H4, W4 = 4096, 10240                        # BASELINE configs[3]: content of the north_star's target frame
PEAK_F32_MFMA_TF = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2_f32, dense
PEAK_HBM_GBS = 8000.0      # spec; 6290 measured copy
PEAK_F16_MFMA_TF = 2500.0  # dense f16/bf16 MFMA (the f16x3 kernels issue 3 MFMAs per algorithmic product)
MEASURED_F16X3_CEILING_TF = 420.0   # profiles/r02_mfma_ceiling_skeleton_random_pmc.txt: bare loop + DMA + stores, random operands
GOLD = os.path.join(REPO, "tests", "golden")


def source_id():
    """sha256 over the sources of the convolution / moments kernels (the ones with HBM traffic worth counting): ties a committed
    PMC summary to the build it was taken from.  Same function in tools/pmc_summary.py."""
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.startswith(("conv", "level1", "moments", "wct_common")) and f.endswith((".hip", ".h")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def live_pmc(config, timeout=90):
    """The two PMC passes of MI355X_MICROARCH.md's HBM recipe, taken NOW: counters cannot be read from inside this process, so a
    bounded copy of this very command (`--steps-only`, 1 warm-up + 2 steps, kernels one at a time) runs twice as a child under
    `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `... WRITE_SIZE` (separate passes, nothing else traced) while this process idles,
    and tools/pmc_summary.py turns the two counter tables into bytes per launch.  -> path of the summary json, or None (no
    rocprofv3, a failed or overlong pass: the caller then falls back to the committed profile and says so)."""
    import shutil
    import subprocess
    import tempfile
    rocprof = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rocprof):
        return None
    tmp = tempfile.mkdtemp(prefix="wct_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp", WCT_OVERLAP="0")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    csvs = []
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(tmp, counter)
            cmd = [rocprof, "--kernel-trace", "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__), "--config", config, "--steps", "2", "--warmup", "1",
                   "--no-cpu-baseline", "--steps-only"]
            r = subprocess.run(cmd, cwd="/tmp", env=env, timeout=timeout, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
            found = [os.path.join(d, f) for d, _, fs in os.walk(out) for f in fs if f.endswith("counter_collection.csv")]
            if r.returncode != 0 or not found:
                sys.stderr.write("live PMC pass %s failed (rc %d): %s\n" % (counter, r.returncode, r.stderr.decode(errors="replace")[-400:]))
                return None
            csvs.append(found[0])
        sys.path.insert(0, os.path.join(REPO, "tools"))
        import pmc_summary
        argv, sys.argv = sys.argv, ["pmc_summary.py", csvs[0], csvs[1], os.path.join(tmp, "hbm_traffic.txt")]
        try:
            pmc_summary.main()
        finally:
            sys.argv = argv
        return os.path.join(tmp, "hbm_traffic.json")
    except Exception as e:   # noqa: BLE001 -- measurement garnish: never take the bench line down with it
        sys.stderr.write("live PMC collection failed: %r\n" % (e,))
        return None
    finally:
        for c in csvs:       # the raw per-dispatch tables are tens of MB
            try:
                os.remove(c)
            except OSError:
                pass


def pmc_traffic(family, live=None):
    """HBM bytes per launch of the kernel behind a profile family from rocprofv3 PMC passes: `live` (live_pmc() of this run) or
    the committed profiles/hbm_traffic_latest.json (tools/profile_round.sh), whose `stale` says whether the kernel sources
    changed since it was taken."""
    import re
    path = live or os.path.join(REPO, "profiles", "hbm_traffic_latest.json")
    if not os.path.exists(path):
        return None
    doc = json.load(open(path))
    ks = doc["kernels"]
    src = {"file": "profiles/hbm_traffic_latest.json", "profiled_source_id": doc.get("source_id"), "source_id": source_id()}
    if live:
        src["file"] = "live: rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE around `bench.py --steps-only --steps 2` children of this run"
    src["stale"] = src["profiled_source_id"] != src["source_id"]
    # the fused full-resolution ends and the level-1 kernels: one device kernel per family (prefix match on the demangled name)
    prefix = {"enc_head_fused": ("void enc_head_roles_kernel<", "void enc_head_kernel<"), "dec_tail_fused": ("void dec_tail_up_kernel<", "void dec_tail_kernel<"),
              "l1_moments_fused": ("void l1_moments_kernel<",), "l1_decode_fused": ("void l1_decode_kernel<",)}.get(family.split("<")[0])
    if prefix:
        rows = [v for k, v in ks.items() if k.startswith(prefix)]
        if not rows:
            return None
        n = sum(r["calls"] for r in rows)
        src["pmc_avg_launch_us"] = round(sum(r["avg_us"] * r["calls"] for r in rows) / n, 2)
        return round(sum((r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["calls"] for r in rows) / n * 1e6), src
    m = re.match(r"conv3x3_f16x3<co=(\d+)(,pool)?(,out3)?(,dma)?>", family)
    if not m:
        return None
    co, pool, out3, dma = int(m.group(1)), bool(m.group(2)), bool(m.group(3)), bool(m.group(4))
    tf = lambda b: "true" if b else "false"   # noqa: E731
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
        src["pmc_avg_launch_us"] = round(sum(r["avg_us"] * r["calls"] for r in rows) / n, 2)
        return round(sum((r["read_MB_per_launch"] + r["write_MB_per_launch"]) * r["calls"] for r in rows) / n * 1e6), src
    if co == 16:
        name = "void conv3x3_f16_c16_kernel<%s, %s>(F16Args)" % (tf(pool), tf(out3))
    else:
        name = "void conv3x3_f16_kernel<%d, %s, %d>(F16Args)" % (min(co, 128) // 32, tf(pool), 16 if co >= 128 else 8)
    e = ks.get(name)
    if not e:
        return None
    src["pmc_avg_launch_us"] = e["avg_us"]
    return round((e["read_MB_per_launch"] + e["write_MB_per_launch"]) * 1e6), src


def family_peak(name):
    """(dense matrix-core peak in TFLOP/s, note) of a profile family (csrc/wct_api.hip ProfScope names) by the arithmetic its products
    run in: split-f16 families issue three f16 MFMAs per algorithmic product; `_f32` / `_fp32` families and the moments' block
    products run on the fp32 MFMA (the deep levels' small maps on the fp64 MFMA: the fp32 peak is then the generous one)."""
    if name.startswith(("conv3x3_f32", "conv3x3_fp32", "moments", "l1_encode")):
        return PEAK_F32_MFMA_TF, "fp32 MFMA"
    return PEAK_F16_MFMA_TF / 3.0, "f16x3: 2.5 PF dense f16 MFMA / 3 split terms"


class Telemetry:
    """GPU clock / power samples during a timed loop, taken by a CHILD process polling the amdgpu sysfs files every ~5 ms (no thread
    of this process: the step loop's enqueue rate must not change) -- so that the 7.3 .. 9.1 ms box-to-box spread of the same build
    (profiles/r04_box_spread_final_build.txt) can be attributed.  The child samples only between start() and stop() and appends to a
    temporary FILE (a pipe nobody drains fills after ~2 s and blocks the writer: ADVICE r5); it is killed at exit whatever happens.
    Nothing readable -> {"source": None}.  read_sclk(): the device's clock NOW, read by this process (one sysfs read, ~20 us)."""
    CHILD = r"""
import sys, time
fp, ff, fm, ft, out = sys.argv[1:6]
def rd(f):
    try:
        return int(open(f).read().split()[0])
    except Exception:
        return -1
sys.stdin.readline()                      # start()
with open(out, "a", buffering=1) as o:
    while True:
        o.write("%f %d %d %d %d\n" % (time.time(), rd(fp), rd(ff), rd(fm), rd(ft)))
        time.sleep(0.005)
"""

    def __init__(self, device_index=0):
        import atexit
        import glob
        import subprocess
        import tempfile
        self.proc, self.t0, self.out, self.f_sclk = None, None, None, None
        # the box may expose many amdgpu cards (partitions of a multi-GPU host): find THIS device's sysfs node through its PCI address
        dev = None
        try:
            pr = torch.cuda.get_device_properties(device_index)
            addr = "%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id)
            if glob.glob("/sys/bus/pci/devices/%s/hwmon/hwmon*" % addr):
                dev = "/sys/bus/pci/devices/%s" % addr
            self.pci = addr
        except Exception:     # noqa: BLE001
            self.pci = None
        if dev is None:
            cards = [c for c in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")) if glob.glob(c + "/hwmon/hwmon*")]
            if len(cards) != 1:          # several candidates and no PCI match: a wrong card's numbers are worse than none
                return

            dev = cards[0]

        def first(pats):
            for pat in pats:
                g = sorted(glob.glob(dev + pat))
                if g:
                    return g[0]
            return "-"
        files = [first(["/hwmon/hwmon*/power1_average", "/hwmon/hwmon*/power1_input"]), first(["/hwmon/hwmon*/freq1_input"]),
                 first(["/hwmon/hwmon*/freq2_input"]), first(["/hwmon/hwmon*/temp2_input", "/hwmon/hwmon*/temp1_input"])]
        self.f_sclk = files[1] if files[1] != "-" else None
        try:
            fd, self.out = tempfile.mkstemp(prefix="wct_telemetry_", suffix=".txt")
            os.close(fd)
            self.proc = subprocess.Popen([sys.executable, "-c", self.CHILD] + files + [self.out], stdin=subprocess.PIPE, stdout=subprocess.DEVNULL,
                                         stderr=subprocess.DEVNULL, text=True)
            atexit.register(self._kill)
        except Exception:     # noqa: BLE001 -- garnish
            self.proc = None

    def _kill(self):
        if self.proc is not None and self.proc.poll() is None:
            self.proc.kill()
        if self.out and os.path.exists(self.out):
            try:
                os.unlink(self.out)
            except OSError:
                pass

    def read_sclk(self):
        """MHz, or None."""
        try:
            return int(open(self.f_sclk).read().split()[0]) * 1e-6 if self.f_sclk else None
        except Exception:     # noqa: BLE001
            return None

    def start(self):
        self.t0 = time.time()
        if self.proc is not None:
            try:
                self.proc.stdin.write("go\n")
                self.proc.stdin.flush()
            except Exception:     # noqa: BLE001
                pass

    def stop(self):
        t1 = time.time()
        if self.proc is None:
            return {"source": None}
        time.sleep(0.01)
        try:
            self.proc.terminate()
            self.proc.wait(timeout=5)
            out = open(self.out).read()
        except Exception:     # noqa: BLE001
            return {"source": None}
        finally:
            self._kill()
        rows = [ln.split() for ln in out.splitlines() if ln and ln[0].isdigit()]
        rows = [[float(v) for v in r] for r in rows if len(r) == 5 and self.t0 <= float(r[0]) <= t1]
        if not rows:
            return {"source": None}
        col = lambda i, scale: [r[i] * scale for r in rows if r[i] >= 0]   # noqa: E731
        pw, sclk, mclk, temp = col(1, 1e-6), col(2, 1e-6), col(3, 1e-6), col(4, 1e-3)
        avg = lambda v: round(sum(v) / len(v), 1) if v else None   # noqa: E731
        return {"source": "amdgpu sysfs hwmon of PCI device %s (power1_average | power1_input, freq1_input, freq2_input, temp), %d samples at ~5 ms over %.1f s (the timed steps + the sustained block)" % (self.pci, len(rows), t1 - self.t0),
                "power_W_avg": avg(pw), "power_W_max": round(max(pw), 1) if pw else None, "sclk_MHz_avg": avg(sclk),
                "sclk_MHz_min": round(min(sclk), 1) if sclk else None, "mclk_MHz_avg": avg(mclk), "temp_C_avg": avg(temp)}


def load_fixture(name):
    path = os.path.join(GOLD, name)
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def cli_folder_pass(wct, depth=3, io_threads=8):
    """Folders of copies (distinct names) of the reference's UHD sample content (3840x2160 JPEG, committed fixture G11) x its 2048x2048
    sample style through wct_hip.cli's two loops with the engine `wct`; JPEG decode, H2D, 5-level cascade (style statistics cached per
    style), D2H and JPEG encode + file write are ALL inside the wall time.  Two folder sizes: 8 contents (VERDICT r3's case: the
    pipeline's fill -- the first decode, ~60 ms on one core -- and drain -- the last save, ~35 ms -- are a large part of it) and 32
    (closer to the steady rate)."""
    import shutil
    import tempfile
    from wct_hip import cli
    src_c, src_s = os.path.join(GOLD, "g11_uhd_content_3840x2160.jpg"), os.path.join(GOLD, "g11_style_2048x2048.jpg")
    root = tempfile.mkdtemp(prefix="wct_cli_bench_")
    log = lambda sth: None     # noqa: E731
    try:
        res = {"workload": "N x 3840x2160 JPEG contents x 1 2048x2048 JPEG style, --mode 16x: decode -> H2D -> cascade -> D2H -> JPEG save, "
                           "all inside the wall time; pipelined = --pipeline %d --io_threads %d" % (depth, io_threads)}
        for n_contents in (8, 32):
            cdir, sdir = os.path.join(root, "content%d" % n_contents), os.path.join(root, "style%d" % n_contents)
            os.makedirs(cdir); os.makedirs(sdir)
            for i in range(n_contents):
                shutil.copyfile(src_c, os.path.join(cdir, "c%02d.jpg" % i))
            shutil.copyfile(src_s, os.path.join(sdir, "s.jpg"))
            pairs = cli.list_pairs(cdir, sdir)
            out_bytes, r = {}, {}
            for tag, pipe in (("serial", 0), ("pipelined", depth)):
                if tag == "serial" and n_contents > 8:
                    continue                  # the serial loop has no fill / drain: its rate is the 8-content one
                outf = os.path.join(root, "out%d_%s" % (n_contents, tag))
                os.makedirs(outf)
                a = cli.build_parser().parse_args(["--mode", "16x", "--contentPath", cdir, "--stylePath", sdir, "--outf", outf, "--log_mark", "B",
                                                   "--pipeline", str(pipe), "--io_threads", str(io_threads)])
                (cli.run_pipelined if pipe else cli.run_serial)(a, wct, pairs[:2], cdir, sdir, log)      # warm-up: page cache, pinned pools
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                (cli.run_pipelined if pipe else cli.run_serial)(a, wct, pairs, cdir, sdir, log)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                r[tag] = {"pairs_per_s": round(len(pairs) / dt, 2), "ms_per_pair": round(dt / len(pairs) * 1e3, 2),
                          "MPs": round(len(pairs) * 3840 * 2160 / 1e6 / dt, 1)}
                out_bytes[tag] = [open(os.path.join(outf, f), "rb").read() for f in sorted(os.listdir(outf)) if f.endswith(".jpg")]
            if "serial" in r:
                r["outputs_byte_identical"] = out_bytes["serial"] == out_bytes["pipelined"]
                r["speedup"] = round(r["serial"]["ms_per_pair"] / r["pipelined"]["ms_per_pair"], 2)
            res["%d_contents" % n_contents] = r
        return res
    finally:
        shutil.rmtree(root, ignore_errors=True)


def oracle_vs_reference_table():
    path = os.path.join(GOLD, "oracle_vs_reference.json")
    return json.load(open(path)) if os.path.exists(path) else None


def cpu_baseline(weights, content, style):
    """The oracle (a CPU port of the reference's op sequence: fp32 convs, fp64 two-GEMM WCT with SVD) on ONE frame of the timed
    configuration.  This and the parity leg are the only places bench.py touches oracle/.
    Threads: the C/OpenMP convolutions stop scaling at ~8-32 threads on the GPU box's host (1.3 s at 8..32 threads,
    3.8 s at 128, 23.6 s at 256 for a 512x512 sample: oversubscription), so min(cores, 32) are used and reported."""
    from oracle import wct_oracle
    threads = min(os.cpu_count() or 1, 32)
    wct_oracle.set_num_threads(threads)
    mods = wct_oracle.Modules("16x", weights)
    t0 = time.perf_counter()
    out = wct_oracle.stylize(mods, content, style, 1.0)
    dt = time.perf_counter() - t0
    assert np.isfinite(out).all()
    # second arm ("best-effort CPU"): the same convolutions, the transform as ONE fp32 affine map
    # csF = M cF + b (M, b from the fp64 C x C statistics) instead of the reference's two fp64 C x C . C x hw GEMMs + elementwise
    # passes -- what a CPU implementer free to restructure would do.  Same full-size frame.
    t1 = time.perf_counter()
    outq = wct_oracle.stylize(mods, content, style, 1.0, fused_affine=True)
    dq = time.perf_counter() - t1
    assert np.isfinite(outq).all()
    mp = content.shape[1] * content.shape[2] / 1e6
    return {"value": round(mp / dt, 5), "unit": "MP/s", "cores": threads, "kind": "port",
            "sample": "ONE frame of the timed configuration: 5-level 16x WCT, %dx%d content + %dx%d style (uniform noise), %.1f s "
                      "wall, %d OpenMP threads for the convolutions (of %d host cores), numpy/OpenBLAS for the fp64 transform"
                      % (content.shape[2], content.shape[1], style.shape[2], style.shape[1], dt, wct_oracle.num_threads(), os.cpu_count() or 1),
            "best_effort": {"value": round(mp / dq, 5), "unit": "MP/s", "cores": threads,
                            "sample": "the same full-size frame, transform as one fp32 affine map (fused), %.1f s wall" % dq}}, out


def uhd_pair_parity(wct):
    """green_park-wallpaper-3840x2160.jpg + style/in1.jpg (2048x2048), the reference's sample data at config-2 size: this library
    against the reference's own pixels (tests/golden/g11_uhd_pair.npz).  None when the fixtures or Pillow are missing."""
    files = [os.path.join(GOLD, f) for f in ("g11_uhd_content_3840x2160.jpg", "g11_style_2048x2048.jpg")]
    g = load_fixture("g11_uhd_pair.npz")
    try:
        from PIL import Image
    except ImportError:
        return None
    if g is None or not all(os.path.exists(f) for f in files):
        return None
    c_u8, s_u8 = (np.array(Image.open(f).convert("RGB")) for f in files)
    got = wct.stylize(wct.to_tensor_u8(torch.from_numpy(c_u8)), wct.to_tensor_u8(torch.from_numpy(s_u8))).cpu().numpy()[0]
    r = compare_to_fixture(got, g)
    return {"hip_vs_reference": r["max"], "down16": r["down16_max"], "gate": GATE, "ok": bool(r["max"] <= GATE),
            "frame": "reference sample data: UHD_content/green_park 3840x2160 + style/in1.jpg 2048x2048 (G11: four 96x96 crops)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=["cfg2", "cfg3", "cfg4"], default="cfg2",
                    help="cfg2: 3840x2160 content per GPU (N > 1: weak scaling over an N x 3840 wide frame); "
                         "cfg4: ONE 10240x4096 content in N column strips (strong scaling); cfg3: --mode original at 1920x1080 (N = 1)")
    ap.add_argument("--halo-mode", choices=["auto", "recompute", "exchange"], default="auto")
    ap.add_argument("--debug-set", action="append", default=[], metavar="KEY=VALUE", help="wct_debug_set switches for A/B runs")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU oracle run (the parity gate then has no oracle arm)")
    ap.add_argument("--no-live-pmc", action="store_true", help="roofline.traffic from the committed profile instead of two rocprofv3 "
                                                               "--pmc child runs of this command (~1 min)")
    ap.add_argument("--steps-only", action="store_true",
                    help="skip the extra passes and the parity leg: every launch then belongs to a stylise step, so a rocprofv3 "
                         "--stats summary of the run averages the same launch mix as `roofline`")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` (the form the driver uses for N = 1) launches itself the way the driver launches N > 1: one
        # process per GPU under torch.distributed.run on 127.0.0.1; rank 0's line is the only thing on stdout.  With fewer devices
        # than ranks (a 1-GPU box) the ranks share the device over gloo -- a code-path check, `config.dist_backend` says so.
        import socket
        import subprocess
        env = dict(os.environ, MASTER_ADDR="127.0.0.1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if torch.cuda.device_count() < args.gpus and "WCT_DIST_BACKEND" not in env:
            sys.stderr.write("bench.py: %d rank(s) on %d device(s): the ranks share a GPU over gloo (WCT_DIST_BACKEND=gloo)\n"
                             % (args.gpus, torch.cuda.device_count()))
            env["WCT_DIST_BACKEND"] = "gloo"
        sk = socket.socket()
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
        sk.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.run(cmd, env=env).returncode)

    # ONE JSON line on stdout, nothing else: RCCL prints a version banner to stdout when a communicator is created (N > 1, and the
    # 1-rank communicator of passes.cfg4_rank_sim), so everything this process and its libraries print goes to stderr and the line
    # is written to the saved descriptor at the end
    sys.stdout.flush()
    line_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with python -m torch.distributed.run --nnodes=1 --nproc-per-node %d bench.py "
                         "--gpus %d, or plainly as python bench.py --gpus %d" % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if args.config == "cfg3" and world > 1:
        raise SystemExit("--config cfg3 is a single-GPU configuration")
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
    wct16 = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights)
    for kv in args.debug_set:
        wct16.debug_set(kv.split("=")[0], float(kv.split("=")[1]))

    def original_engine(weights=None):
        eng = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=weights if weights is not None else model_zoo.synth_weights("original", 3))
        for kv in args.debug_set:
            eng.debug_set(kv.split("=")[0], float(kv.split("=")[1]))
        return eng

    wct = original_engine() if args.config == "cfg3" else wct16
    cu = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()   # noqa: E731

    style_np = noise_frame(4, H3, W3) if args.config == "cfg3" else noise_frame(2, HS, WS)     # the same style on every rank
    style = cu(style_np)
    hs_, ws_ = style_np.shape[1:]

    def frame_columns(x0, x1, cfg):
        """Columns [x0, x1) of the benchmark's (virtual) content frame, host numpy.  cfg2: N seeded 3840-wide panels side by side
        (panel 0 = the N = 1 frame, seed 1; panel r > 0: seed 1000 + r); cfg4: the ONE 10240x4096 frame of seed 5; cfg3: seed 3."""
        if cfg == "cfg4":
            return np.ascontiguousarray(noise_frame(5, H4, W4)[:, :, x0:x1])
        if cfg == "cfg3":
            return noise_frame(3, H3, W3)
        parts = []
        for r in range(x0 // W, (x1 - 1) // W + 1):
            a, b = max(x0, r * W), min(x1, (r + 1) * W)
            parts.append(noise_frame(1 if r == 0 else 1000 + r, H, W)[:, :, a - r * W:b - r * W])
        return np.ascontiguousarray(np.concatenate(parts, axis=2))

    collectives_used = [None]
    sharded_checks = {}         # N > 1: first-contact verdicts of the library's own cascade, per configuration
    first_contact_dev = [0.0]

    def all_ranks_agree(ok):
        """True only if `ok` holds on EVERY rank (one MIN all-reduce; every rank calls it)."""
        t = torch.tensor([1.0 if ok else 0.0], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return bool(t.item() > 0.5)

    def c_transport_ready(eng):
        """The engine's transport for wct_stylize_sharded: its own RCCL communicator over the job's ranks (the product path), checked with
        known data (wct_comm_selftest) -- or, when the ranks share one GPU over gloo (1-GPU box: RCCL refuses two ranks on a device), the
        torch.distributed adapter of tools/sharded_standins.py, so that the same C cascade runs there.  -> description, or raises."""
        if not getattr(eng, "has_comm", False):
            if backend == "nccl":
                eng.comm_init(dist)
            else:
                from tools.sharded_standins import dist_transport
                eng.comm_attach_collectives(*dist_transport(dist), world, rank)
        eng.comm_selftest()
        return ("library: wct_stylize_sharded, ONE call per frame, RCCL on the context's own communicator" if backend == "nccl" else
                "library: wct_stylize_sharded, ONE call per frame, over a %s adapter (ranks share a GPU)" % backend)

    def make_step(cfg, eng, fh=None, fw=None, frame=None):
        """-> (step(), megapixels of the whole frame, description, content on the device).  N > 1: the frame through the library's own
        sharded cascade (wct_stylize_sharded) unless WCT_C_CASCADE=0 or its first contact fails -- transport self-test, then ONE frame
        through it and through wct_hip/sharded.py's orchestration over torch.distributed must agree on every rank (bit for bit where both sum in the
        same order; <= 1e-5 of the image's maximum where two RCCL communicators may order a >= 3-term sum differently); otherwise
        the torch.distributed path is timed and the line says why (config.collectives)."""
        if fh is None:
            fh, fw = {"cfg2": (H, W * world), "cfg3": (H3, W3), "cfg4": (H4, W4)}[cfg]
        if world > 1:
            from wct_hip.sharded import ShardedStylizer
            runner = ShardedStylizer(eng, dist, fh, fw, hs_, ws_, halo_mode=args.halo_mode, c_collectives=False)
            x0, x1 = runner.input_columns()                                # own strip + halo of the halo mode
            content = cu(frame(x0, x1) if frame is not None else frame_columns(x0, x1, cfg))
            used = "torch.distributed (wct_hip/sharded.py orchestration of the split-level entries)"
            if os.environ.get("WCT_C_CASCADE", "1") != "0":
                ok, why, runner_c = True, None, None
                try:
                    note = c_transport_ready(eng)
                    runner_c = ShardedStylizer(eng, dist, fh, fw, hs_, ws_, halo_mode=args.halo_mode, c_cascade=True)
                    for attempt in (0, 1):
                        # (a single-launch matrix-function solve that ABORTS into its Jacobi net -- ranks sharing one GPU, never one process per
                        #  GPU -- is repaired correctly but not bit-identically: such a frame is compared again, once)
                        aborts0 = eng.debug_get("nscoop_aborts")
                        a = runner_c.stylize_strip(content, style)
                        b = runner.stylize_strip(content, style)
                        runner_c.check_range()
                        runner.check_range()
                        # Bitwise when both paths sum the moments in the same order (two ranks: a two-term sum; the one-device worlds of the
                        # tests).  With three or more devices the library's RCCL communicator and torch's are two communicators: their rings may add
                        # the ranks' fp64 terms in different orders -- 1e-16 in the moments, ~1e-7 in the image after five whitenings.  The verdict is
                        # therefore: equal to 1e-5 of the image's maximum (50x below the sharded tests' bound); `bitwise` is reported beside it.
                        bitwise = bool(torch.equal(a, b))
                        dev = 0.0 if bitwise else float((a - b).abs().max() / b.abs().max())
                        ok = dev <= 1e-5
                        why = None if ok else "C cascade differs from the torch.distributed path by %.3e of the image's maximum" % dev
                        first_contact_dev[0] = max(first_contact_dev[0], dev)
                        del a, b
                        again = torch.tensor([1.0 if (not ok and eng.debug_get("nscoop_aborts") > aborts0) else 0.0], device="cuda")
                        dist.all_reduce(again, op=dist.ReduceOp.MAX)          # every rank repeats, or none (the frame contains collectives)
                        if again.item() < 0.5:
                            break
                except Exception as e:      # noqa: BLE001
                    ok, why = False, repr(e)
                    sys.stderr.write("bench.py rank %d: C cascade first contact failed: %r\n" % (rank, e))
                agreed = all_ranks_agree(ok)
                sharded_checks[cfg if frame is None else "g16"] = {"c_cascade_equals_torch_distributed": agreed, "rank0_max_rel_deviation": first_contact_dev[0],
                                                                   "limit": 1e-5, "rank0_note": why}
                first_contact_dev[0] = 0.0
                if agreed:
                    # the TIMED path: the same cascade with the single-GPU fold (WCT_SHARD_FAST_FOLD: no (M, b) and no assemble launch on the critical
                    # path; fp32 round-off from the form checked above) -- what the parity leg then holds against the untiled frame, which folds the same way
                    runner = ShardedStylizer(eng, dist, fh, fw, hs_, ws_, halo_mode=args.halo_mode, c_cascade=True, fast_fold=True)
                    used = note + "; fast fold"
                else:
                    used += " -- the library's cascade was NOT used: " + (why or "another rank's first-contact check failed")
            collectives_used[0] = used
            desc = "%dx%d content in %d column strips (halo: %s; style side: %s), %dx%d style" % (fw, fh, world, runner.halo_mode, runner.style_mode, ws_, hs_)
            step = lambda: runner.stylize_strip(content, style)     # noqa: E731
            step.runner = runner
            return step, fh * fw / 1e6, desc, content
        content = cu(frame_columns(0, fw, cfg))
        eng.reserve(fh, fw, hs_, ws_)
        out = torch.empty((3, fh, fw), device="cuda")
        return (lambda: eng.stylize(content, style, out=out)), fh * fw / 1e6, "%dx%d content, %dx%d style" % (fw, fh, ws_, hs_), content

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

    def kernel_profile(eng, step, nprof=2, record=True):
        """HIP events around every kernel launch on the context's stream, overlap off so one event pair times one launch.
        Every rank runs the steps (they contain collectives); only a recording rank gets
        (rows for the `kernels` list, dominant conv family as a roofline dict, kernel time per step)."""
        if record:
            eng.set_overlap(False)
            eng.profile_reset()
            eng.profile(True)
        for _ in range(nprof):
            step()
        barrier()
        if not record:
            return None, None, None
        eng.profile(False)
        eng.set_overlap(True)
        ents = sorted(eng.profile_read(), key=lambda e: -e["ms"])
        tot = sum(e["ms"] for e in ents)
        rows = [{"kernel": e["name"], "ms_per_step": round(e["ms"] / nprof, 4), "launches_per_step": e["launches"] // nprof,
                 "tflops": round(e["flops"] / e["ms"] / 1e9, 2) if e["flops"] else None,
                 "algo_GBs": round(e["bytes"] / e["ms"] / 1e6, 1) if e["bytes"] else None} for e in ents]
        # dominant kernel FAMILY = the largest time share among ALL families that carry algorithmic FLOPs (the fused full-resolution
        # ends included, VERDICT r4 #11 -- not only the conv3x3 names); its binding roofline is the larger of the two fractions
        d = [e for e in ents if e["flops"] > 0][0]
        peak_tf, peak_note = family_peak(d["name"])
        f16 = "f16x3" in peak_note
        tf, gbs = d["flops"] / d["ms"] / 1e9, d["bytes"] / d["ms"] / 1e6
        if gbs / PEAK_HBM_GBS >= tf / peak_tf:
            roof = {"kernel": d["name"], "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                    "frac": round(gbs / PEAK_HBM_GBS, 4), "traffic": None}
        else:
            roof = {"kernel": d["name"], "bound": "mfma", "achieved": round(tf, 2), "peak": round(peak_tf, 1), "unit": "TFLOP/s",
                    "frac": round(tf / peak_tf, 4), "traffic": None, "peak_note": peak_note}
            if f16:
                roof["frac_of_measured_ceiling_420TF"] = round(tf / MEASURED_F16X3_CEILING_TF, 4)
        roof.update({"avg_launch_ms": round(d["ms"] / d["launches"], 4), "share_of_kernel_time": round(d["ms"] / tot, 3),
                     "algo_flop_per_launch": d["flops"] / d["launches"], "algo_bytes_per_launch": d["bytes"] / d["launches"]})
        return rows, roof, round(tot / nprof, 3)

    def latency_median(step, n=10):
        """SURVEY 8(d)'s latency definition beside the throughput mean: every frame individually synchronised, median of n."""
        ts = []
        for _ in range(n):
            barrier()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        ts.sort()
        return round(ts[len(ts) // 2], 3), round(ts[0], 3), round(ts[-1], 3)

    step, mp, desc, content = make_step(args.config, wct)
    collectives_timed = collectives_used[0]
    wct.saturation_count(reset=True)
    tele = Telemetry(dev) if rank == 0 else None
    for _ in range(args.warmup):
        step()
    if tele:
        tele.start()
    dt = timed(step, args.steps, 0)
    value = mp * args.steps / dt

    def sustained_block(seconds=2.0, n_fit=60):
        """Comparable-across-devices figures (VERDICT r5 task 8; the same binary measured 7.27 .. 7.97 ms on four devices at 2.11 .. 2.25 GHz):
        (1) ms per step of K more steps timed AFTER `seconds` of back-to-back steps (clocks and temperature settled); (2) n_fit steps
        synchronised one by one with the device clock read beside each -> least squares ms = a + b * (2200 / sclk) and its prediction at
        2200 MHz (the step is matrix-core / issue bound: time ~ 1 / sclk; the HBM-side share is the intercept).  Every rank runs the
        steps (collectives); the clock is rank 0's device."""
        n = max(args.steps, int(seconds / max(dt / args.steps, 1e-4)))
        for _ in range(n):
            step()
        dts = timed(step, args.steps, 0)
        res = {"ms_per_step_sustained": round(dts / args.steps * 1e3, 3), "after_s_of_back_to_back_steps": round(n * dt / args.steps, 2)}
        pts = []
        for _ in range(n_fit):
            barrier()
            t0 = time.perf_counter()
            step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) * 1e3
            f = tele.read_sclk() if tele else None
            if f:
                pts.append((2200.0 / f, ms, f))
        if len(pts) >= 8:
            x, y = np.array([q[0] for q in pts]), np.array([q[1] for q in pts])
            res["sclk_MHz_during_fit"] = [round(float(min(q[2] for q in pts)), 0), round(float(max(q[2] for q in pts)), 0)]
            res["latency_ms_avg_during_fit"] = round(float(y.mean()), 3)
            # time ~ 1 / sclk for the whole step is the simple model; the fitted line says how much of it the data supports
            res["latency_ms_scaled_to_2200MHz"] = round(float((y / x).mean()), 3)
            # two-box calibration of round 6 (7.313 ms at 2259 MHz, 7.900 ms at 2021 MHz): 31 % of the step does not follow the shader clock
            res["latency_ms_at_2200MHz_two_box_model"] = round(float((y / (0.31 + 0.69 * x)).mean()), 3)
            if float(x.max() - x.min()) > 0.004:          # the clock moved by > 0.4 % over the samples: a slope can be estimated
                b, a = np.polyfit(x, y, 1)
                r = float(np.corrcoef(x, y)[0, 1])
                res["fit_ms_vs_2200_over_sclk"] = {"intercept_ms": round(float(a), 3), "slope_ms": round(float(b), 3), "r": round(r, 3),
                                                   "latency_ms_at_2200MHz": round(float(a + b), 3)}
        return res

    sustained = sustained_block() if not args.steps_only else None
    telemetry = tele.stop() if tele else None
    saturated = wct.saturation_count()      # threads that clamped an activation to the f16x3 range during the timed steps
    lat_med, lat_min, lat_max = latency_median(step, max(10, min(args.steps, 20)))

    # ---- roofline leg (rank 0 records; every rank runs the steps: they contain collectives)
    profile, roof, _ = kernel_profile(wct, step, record=(rank == 0))

    def sharded_vs_untiled(runner, strip_in, cfg, fw, frame=None):
        """N > 1 parity, part (b): this rank's owned strip of the sharded frame against the UNTILED frame computed on this same GPU by the
        single-GPU cascade (no further collective: every rank builds the whole frame itself).  -> max over ranks of max|d| / max|untiled|."""
        out = runner.stylize_strip(strip_in, style)
        runner.check_range()
        full = cu(frame(0, fw) if frame is not None else frame_columns(0, fw, cfg))
        ref = wct16.stylize(full, style)
        a = runner.own[0]
        err = float((out - ref[..., a:a + out.shape[-1]]).abs().max() / ref.abs().max())
        del full, ref, out
        t = torch.tensor([err], device="cuda", dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    sharded_parity = None
    if world > 1 and not args.steps_only:
        fw_ = {"cfg2": W * world, "cfg4": W4}[args.config]
        # bound: the sharded tests' -- 5e-4; 1e-3 on the 4096-row config-4 frame (tests/test_sharded_gpu.py: sharded and untiled (M, b) differ
        # by ~1e-8 in the moments, which the five whitenings of a 42 MP noise frame amplify further than those of an 8 MP one)
        sharded_parity = {"timed_frame_strips_vs_untiled_same_gpu": sharded_vs_untiled(step.runner, content, args.config, fw_),
                          "limit": 1e-3 if args.config == "cfg4" else 5e-4}
    del step

    # ---- extra passes (never `value`)
    extra = not args.steps_only
    passes = {}
    if extra and world > 1 and args.config == "cfg2":
        # BASELINE configs[3] beside the weak-scaling number: ONE 10240x4096 frame in N strips (strong scaling)
        del content
        step4, mp4, desc4, content4 = make_step("cfg4", wct)
        k4 = max(3, args.steps // 2)
        dt4 = timed(step4, k4, 2)
        passes["cfg4_strong"] = {"workload": desc4, "ms_per_frame": round(dt4 / k4 * 1e3, 3), "MPs": round(mp4 * k4 / dt4, 1), "scaling": "strong",
                                 "collectives": collectives_used[0], "strips_vs_untiled_same_gpu": sharded_vs_untiled(step4.runner, content4, "cfg4", W4)}
        del step4, content4
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

    if extra and rank == 0 and world == 1 and args.config != "cfg3":
        content4k = content if args.config == "cfg2" else cu(frame_columns(0, W, "cfg2"))
        # relu4_1 encode pass on the 4K content (north_star's named pass)
        ms = ev_ms(lambda: wct16.encode(4, content4k, layout="nhwc"))
        passes["relu4_1_encode"] = {"ms": round(ms, 3), "algo_GBs": round(364.0 * H * W / ms / 1e6, 1),
                                    "frac_hbm_8TBs": round(364.0 * H * W / ms / 1e6 / PEAK_HBM_GBS, 4),
                                    "tflops": round(30816.0 * H * W / ms / 1e9, 2),
                                    "frac_f16x3_mfma_833TF": round(30816.0 * H * W / ms / 1e9 / (PEAK_F16_MFMA_TF / 3.0), 4)}
        # the cascade against cached style statistics ("style cached" reported separately)
        wct16.style_prepare(style)
        msc = ev_ms(lambda: wct16.stylize_prepared(content4k))
        passes["style_cached_cascade"] = {"ms": round(msc, 3), "MPs": round(H * W / 1e6 / msc * 1e3, 1)}
        # BASELINE configs[4] seen from ONE GPU: distinct 4K contents against one style, three frames in flight
        # (wct_hip/pipeline.py; the per-GPU throughput of the batch / video case; 8 such GPUs share nothing but the style statistics)
        from wct_hip.pipeline import FramePipeline
        pipe = FramePipeline(lambda: WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights), slots=3)
        pipe.set_style(style)
        frames = [content4k] + [cu(noise_frame(10 + i, H, W)) for i in range(3)]     # SURVEY 8d cfg5 seeds 10..
        frames = (frames * 3)[:12]
        pipe.stylize_many(frames[:6])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        pipe.stylize_many(frames)
        torch.cuda.synchronize()
        msp = (time.perf_counter() - t0) / len(frames) * 1e3
        passes["cfg5_per_gpu"] = {"workload": "distinct 3840x2160 contents x 1 style (2048x2048), style statistics cached, three frames "
                                              "in flight on this GPU (wct_hip/pipeline.py)", "ms_per_frame": round(msp, 3),
                                  "MPs": round(H * W / 1e6 / msp * 1e3, 1)}
        del pipe, frames
        # the reference's timed region includes save_image (WCT.py:118-131): uint8 frame in pinned host memory -> H2D ->
        # ToTensor + cascade + save_image's conversion on the device (wct_stylize_u8) -> D2H, and the host JPEG encode beside it
        c_u8 = torch.from_numpy((frame_columns(0, W, "cfg2").transpose(1, 2, 0) * 255).astype(np.uint8)).pin_memory()
        s_u8 = torch.from_numpy((style_np.transpose(1, 2, 0) * 255).astype(np.uint8)).pin_memory()
        o_u8 = torch.empty((H, W, 3), dtype=torch.uint8).pin_memory()

        def u8_frame():
            o_u8.copy_(wct16.stylize_u8(c_u8.cuda(non_blocking=True), s_u8.cuda(non_blocking=True)), non_blocking=True)
        for _ in range(2):
            u8_frame()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            u8_frame()
        torch.cuda.synchronize()
        ms_u8 = (time.perf_counter() - t0) / 5 * 1e3
        u8 = {"workload": "3840x2160 uint8 content + 2048x2048 uint8 style in pinned host memory -> H2D -> wct_stylize_u8 -> D2H "
                          "(pinned)", "ms_per_frame": round(ms_u8, 3), "MPs": round(H * W / 1e6 / ms_u8 * 1e3, 1)}
        try:
            import io
            from PIL import Image
            t0 = time.perf_counter()
            Image.fromarray(o_u8.numpy()).save(io.BytesIO(), format="JPEG")
            u8["host_jpeg_encode_ms"] = round((time.perf_counter() - t0) * 1e3, 1)
            u8["note"] = "the JPEG codec stays on the host (Pillow, one core) and is NOT in ms_per_frame"
        except ImportError:
            pass
        passes["u8_end_to_end"] = u8
        del c_u8, s_u8, o_u8
        # the reference's timed region as a FOLDER run (WCT.py:112-131 + data_loader.py:46-59): 8 x 4K JPEG contents x 1 style through the CLI's
        # loop, serial (the reference's structure) and pipelined (decode-ahead pool, async copies, writer pool) -- same engine, same bytes out
        try:
            passes["cli_pairs_per_s"] = cli_folder_pass(wct16)
        except Exception as e:      # Pillow missing, tmp dir not writable ...: a report line, not a failure of the bench
            passes["cli_pairs_per_s"] = {"error": repr(e)}
        if args.config == "cfg2":
            # BASELINE configs[3]'s frame, 10240x4096, untiled on this ONE GPU (the north_star's target configuration)
            c4 = cu(frame_columns(0, W4, "cfg4"))
            o4 = torch.empty((3, H4, W4), device="cuda")
            ms4 = ev_ms(lambda: wct16.stylize(c4, style, out=o4), n=3, warm=1)
            ms4e = ev_ms(lambda: wct16.encode(4, c4, layout="nhwc"), n=3, warm=1)
            passes["cfg4_single_gpu"] = {"workload": "10240x4096 content, 2048x2048 style, untiled on one GPU", "ms_per_frame": round(ms4, 3),
                                         "MPs": round(H4 * W4 / 1e6 / ms4 * 1e3, 1), "finite": bool(torch.isfinite(o4).all()),
                                         "relu4_1_encode_ms": round(ms4e, 3),
                                         "relu4_1_encode_frac_hbm_8TBs": round(364.0 * H4 * W4 / ms4e / 1e6 / PEAK_HBM_GBS, 4)}
            del o4
            # one rank's share of the 8-GPU jobs, timed on this GPU with its peers emulated (tools/sharded_standins.py LoopbackGroup): ranks 0
            # (edge strip, style level 5) and 3 (interior, level 2).  cfg4: ONE 10240x4096 frame in 8 x 1280 columns (exchange-mode
            # halos, strong scaling); cfg2x8: the driver's default N = 8 workload, 8 x 3840 columns (recompute halos, weak scaling)
            passes["cfg4_rank_sim"] = rank_sim(wct16, style, H4, W4, lambda a, b: c4[:, :, a:b].contiguous(), ms4, "strong")
            del c4
            passes["cfg2x8_rank_sim"] = rank_sim(wct16, style, H, W * 8, lambda a, b: cu(frame_columns(a, b, "cfg2")), dt / args.steps * 1e3, "weak")
            finish_rank_sims()

    if extra and rank == 0 and world == 1 and args.config == "cfg2":
        # the reference's own arithmetic class beside the headline (VERDICT r4 task 3): the SAME frame, the SAME call, with
        # wct_set_conv_mode(0) = exact-fp32 MFMA products everywhere (model_cd.py:724-743 are plain fp32 nn.Conv2d); its dominant
        # family against the 157.3 TF fp32-MFMA roofline and its own distance from the reference's pixels (G13)
        eng32 = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights)
        eng32.set_conv_mode("fp32")
        eng32.reserve(H, W, hs_, ws_)
        o32 = torch.empty((3, H, W), device="cuda")
        step32 = lambda: eng32.stylize(content, style, out=o32)   # noqa: E731
        ms32 = ev_ms(step32, n=5, warm=2)
        rows32, roof32, ksum32 = kernel_profile(eng32, step32)
        got32 = step32().cpu().numpy()[0]
        p32 = {"workload": "the timed frame through wct_set_conv_mode(0): exact-fp32 MFMA products in every convolution (the reference's "
                           "arithmetic class; SURVEY 8d floor for this frame at 157.3 TF: 13.3 ms)",
               "ms_per_frame": round(ms32, 3), "MPs": round(H * W / 1e6 / ms32 * 1e3, 1), "dtype": "f32 (v_mfma_f32_32x32x2_f32 / 16x16x4_f32 products)",
               "algo_TFLOPs": round((200736.0 * H * W + 100368.0 * hs_ * ws_) / ms32 / 1e9, 1),
               "algo_frac_fp32_mfma_157TF": round((200736.0 * H * W + 100368.0 * hs_ * ws_) / ms32 / 1e9 / PEAK_F32_MFMA_TF, 4),
               "headline_f16x3_speedup": round(ms32 / (dt / args.steps * 1e3), 2),
               "kernel_time_sum_ms": ksum32, "roofline": roof32, "kernels": rows32[:8]}
        g13f = load_fixture("g13_cfg2_noise.npz")
        if g13f is not None:
            r32 = compare_to_fixture(got32, g13f)
            p32["parity"] = {"hip_vs_reference": r32["max"], "hip_vs_reference_p9999": r32["lattice_p9999"], "limit": GATE,
                             "ok": bool(r32["max"] <= GATE), "reference": "G13 (the reference's own pixels on the timed frame)"}
        passes["cfg2_fp32_exact"] = p32
        del eng32, o32, got32

    if extra and rank == 0 and world == 1:
        # BASELINE configs[2]: --mode original, generated weights
        eng3 = wct if args.config == "cfg3" else original_engine()
        c3, s3 = cu(noise_frame(3, H3, W3)), cu(noise_frame(4, H3, W3))
        eng3.reserve(H3, W3, H3, W3)
        o3 = torch.empty((3, H3, W3), device="cuda")
        step3 = lambda: eng3.stylize(c3, s3, out=o3)   # noqa: E731
        eng3.saturation_count(reset=True)
        ms3 = ev_ms(step3, n=5, warm=2)
        sat3 = eng3.saturation_count()
        rows3, roof3, ksum3 = kernel_profile(eng3, step3)
        got3 = step3().cpu().numpy()[0]
        p3 = {"weights": "GENERATED (model_zoo.synth_weights('original', 3)); the torch7 checkpoints are absent: real-weight parity unpinned",
              "f16x3_saturated_threads": int(sat3)}
        g14 = load_fixture("g14_cfg3_original.npz")
        if g14 is not None:
            r = compare_to_fixture(got3, g14)
            p3.update({"hip_vs_reference": r["max"], "lattice_p9999": r.get("lattice_p9999"), "lattice_frac_gt_1e-3": r.get("lattice_frac_gt_gate"),
                       "down16": r["down16_max"],
                       "reference": "G14: the reference's own classes (model_original.py) + util_wct.WCT.transform with the same generated "
                                    "weights, torch CPU; random 512-channel stacks are chaotic under five whitenings (oracle vs reference "
                                    "on this frame: oracle_vs_reference) -- gated relative to the oracle; the strict gate of this graph is g15_strict"})
            ovr = (oracle_vs_reference_table() or {}).get("g14_cfg3_original")
            if ovr is not None:
                p3["oracle_vs_reference"] = ovr["oracle_vs_reference"]
                p3["oracle_vs_reference_source"] = "tests/golden/oracle_vs_reference.json (tools/oracle_vs_reference.py, measured in the build container, 8 threads; tests/test_hip_scale.py recomputes it on the GPU box's host, where BLAS threading moves it by a few per cent: 2.32e-3 here, 2.38e-3 there)"
                p3["limit"] = max(1e-3, 1.25 * ovr["oracle_vs_reference"])
                p3["within_limit"] = bool(r["max"] <= p3["limit"])
                p3["limit_source"] = ("committed file, measured on another host (not recomputed in this run): INFORMATIVE ONLY -- the pass / fail "
                                      "signal of this graph is g15_strict (literal 1e-3); tests/test_hip_scale.py recomputes the oracle arm on the box")
            else:
                p3["limit_source"] = "MISSING: tests/golden/oracle_vs_reference.json has no g14_cfg3_original entry -- no relative limit reported"
                sys.stderr.write("bench.py: tests/golden/oracle_vs_reference.json lacks g14_cfg3_original: cfg3 `limit` not reported\n")
        # the same graph at the LITERAL 1e-3: G15, well-conditioned generated weights (paired-isometry layers), the reference's classes
        g15 = load_fixture("g15_cfg3_conditioned_noise.npz")
        if g15 is not None and args.config != "cfg3":
            eng15 = original_engine(model_zoo.synth_weights_conditioned("original", 15))
            eng15.saturation_count(reset=True)
            got15 = eng15.stylize(c3, s3).cpu().numpy()[0]
            r15 = compare_to_fixture(got15, g15)
            ovr15 = (oracle_vs_reference_table() or {}).get("g15_cfg3_conditioned_noise", {})
            p3["g15_strict"] = {"frame": "config 3's frame (seeds 3 / 4), weights model_zoo.synth_weights_conditioned('original', 15); reference = "
                                         "model_original.py Encoder/Decoder{1..5} + util_wct.WCT.transform on them (tools/make_goldens.py gen_g15)",
                                "hip_vs_reference": r15["max"], "lattice_p9999": r15.get("lattice_p9999"), "down16": r15["down16_max"],
                                "oracle_vs_reference": ovr15.get("oracle_vs_reference"), "limit": 1e-3,
                                "f16x3_saturated_threads": int(eng15.saturation_count()), "ok": bool(r15["max"] <= 1e-3)}
            del eng15, got15
        passes["cfg3_original"] = {"workload": "PytorchWCT/WCT.py --mode original, 1920x1080 content + 1920x1080 style -> 1920x1072 (BASELINE configs[2])",
                                   "ms_per_frame": round(ms3, 3), "MPs": round(H3 * W3 / 1e6 / ms3 * 1e3, 2),
                                   # 3 094 272 FLOP per content pixel (enc + dec, 5 levels) + 1 547 136 per style pixel
                                   "algo_TFLOPs": round((3094272.0 + 1547136.0) * H3 * W3 / ms3 / 1e9, 1),
                                   "algo_frac_f16x3_833TF": round((3094272.0 + 1547136.0) * H3 * W3 / ms3 / 1e9 / (PEAK_F16_MFMA_TF / 3.0), 4),
                                   "algo_frac_of_measured_ceiling_420TF": round((3094272.0 + 1547136.0) * H3 * W3 / ms3 / 1e9 / MEASURED_F16X3_CEILING_TF, 4),
                                   "kernel_time_sum_ms": ksum3, "roofline": roof3, "parity": p3,
                                   "kernels": rows3[:10]}
        if eng3 is not wct:
            del eng3

    cpu, parity, parity_ok = None, None, None
    if rank == 0 and world == 1 and args.config == "cfg2" and extra:
        got = wct.stylize(content, style).cpu().numpy()[0]     # the timed call on the timed inputs
        g13 = load_fixture("g13_cfg2_noise.npz")
        parity = {"gate": "hip_vs_reference <= max(1e-3, 1.25 * oracle_vs_reference); reference UHD pair <= 1e-3; no f16x3 saturation",
                  "frame": "the timed frame (%dx%d seed 1 + %dx%d seed 2, numpy default_rng)" % (W, H, WS, HS),
                  "f16x3_saturated_threads": int(saturated)}
        parity_ok = saturated == 0
        rh = None
        if g13 is not None:
            assert abs(float(content.sum(dtype=torch.float64)) - float(g13["content.checksum"])) < 1e-3, "timed frame != G13's input"
            rh = compare_to_fixture(got, g13)
            parity.update({"reference": "G13: util_wct.WCT (real 16x checkpoints, torch %s CPU) on this frame; 1/16 lattice + 8 crops "
                                        "= %d reference pixels" % (str(g13["torch"]), rh["lattice_pixels"] + 8 * 3 * 96 * 96),
                           "hip_vs_reference": rh["max"], "hip_vs_reference_p9999": rh["lattice_p9999"],
                           "hip_frac_pixels_gt_1e-3": rh["lattice_frac_gt_gate"], "hip_vs_reference_down16": rh["down16_max"]})
        else:
            parity_ok = False
            parity["error"] = "tests/golden/g13_cfg2_noise.npz missing"
        ro = None
        if not args.no_cpu_baseline:
            cpu, ref = cpu_baseline(weights, content.cpu().numpy(), style_np)
            parity["hip_vs_oracle"] = float(np.abs(got - ref).max() / np.abs(ref).max())
            if g13 is not None:
                ro = compare_to_fixture(ref, g13)
                parity.update({"oracle_vs_reference": ro["max"], "oracle_vs_reference_p9999": ro["lattice_p9999"],
                               "oracle_frac_pixels_gt_1e-3": ro["lattice_frac_gt_gate"]})
        if rh is not None:
            limit = max(GATE, 1.25 * ro["max"]) if ro is not None else GATE
            parity["limit"] = limit
            parity["within_1e-3_of_reference"] = bool(rh["max"] <= GATE)
            parity_ok = parity_ok and rh["max"] <= limit
        # the reference's own UHD sample pair at the same size (tests/golden/g11_*: data files of the reference + the
        # reference's own output pixels): natural images leave the north_star gate an order of magnitude of room
        up = uhd_pair_parity(wct)
        if up is not None:
            parity["reference_uhd_pair"] = up
            parity_ok = parity_ok and up["ok"]
        parity_ok = bool(parity_ok)
    elif world > 1 and extra:
        # N > 1: a throughput line must carry a correctness signal too (VERDICT r5 missing #4).  Three checks, every rank takes part:
        #   (a) first contact: the library's cascade == wct_hip/sharded.py over torch.distributed on every rank (make_step: bit for bit wherever the
        #       two sum the moments in the same order, <= 1e-5 of the image's maximum otherwise);
        #   (b) every rank's owned strip of the TIMED frame against the untiled frame computed on its own GPU, <= 5e-4 (the sharded tests' bound);
        #   (c) BASELINE configs[3]'s geometry against THE REFERENCE'S OWN PIXELS: G16's 10240x512 frame in N strips, <= 1e-3
        parity = {"gate": "N > 1: library cascade == torch.distributed orchestration (first contact, every rank: bitwise, or <= 1e-5 where two RCCL rings order the sums differently); every rank's strip of the "
                          "timed frame vs the untiled frame on its own GPU <= 5e-4; G16 (10240x512 in N strips) vs the reference's pixels <= 1e-3; "
                          "no f16x3 saturation", "f16x3_saturated_threads": int(saturated), "first_contact": sharded_checks}
        parity.update(sharded_parity or {})
        ok = saturated == 0 and sharded_parity is not None and sharded_parity["timed_frame_strips_vs_untiled_same_gpu"] <= sharded_parity["limit"]
        if "cfg4_strong" in passes:
            ok = ok and passes["cfg4_strong"]["strips_vs_untiled_same_gpu"] <= 1e-3      # (the 4096-row frame: the sharded tests' bound at that size)
        if os.environ.get("WCT_C_CASCADE", "1") != "0":
            ok = ok and all(v["c_cascade_equals_torch_distributed"] for v in sharded_checks.values())
        g16 = load_fixture("g16_cfg4_geometry.npz")
        if g16 is not None:
            from tests.fixture_compare import cfg4_geometry_frames
            c_np, s_np = cfg4_geometry_frames()
            if tuple(s_np.shape) == tuple(style_np.shape) and np.array_equal(s_np, style_np):
                step_g, _, _, content_g = make_step("cfg4", wct16, fh=512, fw=10240, frame=lambda a, b: np.ascontiguousarray(c_np[:, :, a:b]))
                out_g = step_g()
                step_g.runner.check_range()
                parts = [None] * world
                dist.all_gather_object(parts, (step_g.runner.own[0], out_g.cpu().numpy()))
                g_ok = True
                if rank == 0:
                    rg = compare_to_fixture(np.concatenate([p[1] for p in sorted(parts, key=lambda t: t[0])], axis=3)[0], g16)
                    g_ok = bool(rg["max"] <= GATE)
                    parity["g16_cfg4_geometry"] = {"hip_vs_reference": rg["max"], "p9999": rg["lattice_p9999"], "limit": GATE, "strips": world, "ok": g_ok,
                                                   "reference": "G16: util_wct.WCT (real 16x checkpoints, torch CPU) on the 10240x512 seed-5 frame, 983 040 lattice pixels + 8 crops"}
                ok = all_ranks_agree(ok and g_ok)
                del step_g, content_g, out_g, parts
            else:
                parity["g16_cfg4_geometry"] = {"skipped": "the timed style is not G16's (2048x2048, seed 2)"}
                ok = all_ranks_agree(ok)
        else:
            parity["g16_cfg4_geometry"] = {"error": "tests/golden/g16_cfg4_geometry.npz missing"}
            ok = all_ranks_agree(False)
        parity_ok = bool(ok)
    elif saturated:
        parity_ok = False
        parity = {"f16x3_saturated_threads": int(saturated)}

    if rank == 0:
        # HBM bytes per launch of the dominant kernels: PMC passes taken now (N = 1 default run), else the committed profile (a
        # number, or null).  LAST, with every timed pass done: the GPU idles while the children run, and a pass timed right
        # after them ran on cold clocks (relu4_1_encode 0.89 -> 1.05 ms when this sat in front of the passes).
        live_ok = world == 1 and not args.steps_only and not args.no_live_pmc
        live = live_pmc(args.config) if live_ok else None
        pm = pmc_traffic(roof["kernel"], live) or pmc_traffic(roof["kernel"])
        roof["traffic"], roof["traffic_source"] = (pm[0], pm[1]) if pm else (None, None)
        # what ran the timed steps (inside `roofline` so that a record keeping only this object still says what the device did)
        roof["device"] = {"sclk_MHz_avg": (telemetry or {}).get("sclk_MHz_avg"), "power_W_avg": (telemetry or {}).get("power_W_avg"),
                          "ms_per_step": round(dt / args.steps * 1e3, 3),
                          "ms_per_step_sustained": (sustained or {}).get("ms_per_step_sustained"),
                          "latency_ms_scaled_to_2200MHz": (sustained or {}).get("latency_ms_scaled_to_2200MHz"),
                          "latency_ms_at_2200MHz_two_box_model": (sustained or {}).get("latency_ms_at_2200MHz_two_box_model")}
        r3 = passes.get("cfg3_original", {}).get("roofline") if args.config != "cfg3" else None
        if r3 and live:      # (not attempted when the first collection failed: bounded run time)
            live3 = live_pmc("cfg3")
            pm3 = pmc_traffic(r3["kernel"], live3) if live3 else None
            r3["traffic"], r3["traffic_source"] = (pm3[0], pm3[1]) if pm3 else (None, None)
        cfg = args.config
        mode = "original (generated weights)" if cfg == "cfg3" else "16x"
        line = {
            "metric": "content megapixels/sec, end-to-end 5-level WCT (%s)" % {"cfg2": "16x VGG, 4K content, 2K style", "cfg4": "16x VGG, 10240x4096 content, 2K style",
                                                                              "cfg3": "un-pruned VGG-19, generated weights, 1920x1080"}[cfg],
            "value": (round(value, 2) if parity_ok is not False else None), "unit": "MP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "latency_ms_median": lat_med, "latency_ms_min_max": [lat_min, lat_max],
            "latency_note": "SURVEY 8(d): each frame synchronised on its own, median of >= 10 (a lone call cannot hide its launch "
                            "latency or its style lane's tail behind the next frame); `value` / ms_per_step = K back-to-back steps, one sync",
            "gpu_telemetry": telemetry, "sustained": sustained, "higher_is_better": True, "scaling": "strong" if cfg == "cfg4" else "weak",
            "vs_baseline": None, "dtype": "f32 (f16x3 split-MFMA products, fp32 accumulate)", "data": "synthetic",
            "parity_ok": parity_ok, "parity": parity,
            "config": {"workload": "PytorchWCT/WCT.py --mode %s, 5-level WCT, %s, alpha=1, style-side work included, images resident in HBM; %s"
                                   % (mode, desc, {"cfg2": "BASELINE configs[1]" + ("" if world == 1 else " x%d wide (weak scaling)" % world),
                                                   "cfg3": "BASELINE configs[2]", "cfg4": "BASELINE configs[3] (strong scaling)"}[cfg]),
                       "name": ("cfg2" if world == 1 else "cfg2x%d" % world) if cfg == "cfg2" else cfg,
                       "content_total": "%dx%d" % {"cfg2": (W * world, H), "cfg3": (W3, H3), "cfg4": (W4, H4)}[cfg], "parallelism": "content column strips x%d" % world,
                       "dist_backend": (backend if world > 1 else None), "collectives": collectives_timed},
            "roofline": roof, "passes": passes or None, "cpu_baseline": cpu, "kernels": profile,
        }
        if parity_ok is False:
            line["unverified_value"] = round(value, 2)
        os.write(line_fd, (json.dumps(line) + "\n").encode())
    if dist is not None:
        dist.destroy_process_group()
    if parity_ok is False:
        sys.exit(1)


def rank_sim(wct, style, Hf, Wf, strip_of, ms_one_gpu, scaling, world=8, ranks=(0, 3), frames=4):
    """What ONE rank of the `world`-GPU job executes (its strip + halos, its share of the style side, the per-level collectives as
    launches on a 1-rank RCCL communicator, the neighbour exchange with itself as the neighbour), timed on this GPU.
    Three arrangements per rank: `torch_distributed_owner` = round 5's (Python orchestration, style levels dealt out whole: rank 0 carries level
    5 = 45.6 % of the style FLOPs), `torch_distributed` = the same orchestration with the STYLE cut into column strips too (round 6), `c_cascade` =
    the whole frame as ONE library call (wct_stylize_sharded, style strips, collectives issued by the library on a 1-rank RCCL communicator with
    the rank's geometry emulated: debug key shard_emulate).
    `host_enqueue_ms`: wall time until stylize_strip has returned for every frame = `pure_enqueue_ms` (Python + torch.distributed +
    ctypes orchestration, nothing waited for) + `range_flag_wait_ms` (stylize_strip reads the node-wide f16x3 flag of frame k - 2 at a
    FIXED lag, ADVICE r3, and blocks until that frame's read-back has landed: GPU time, not orchestration -- VERDICT r4 weak #6);
    `ms_per_frame`: the same frames with the final sync (the better of the two round-6 paths bounds the prediction;
    `predicted_efficiency_round5_arrangement` is the same figure for round 5's arrangement).  The slowest rank bounds
    the N-GPU frame time: no link time, no skew -- the compute-only scaling prediction.
    strip_of(x0, x1) -> device tensor of content columns [x0, x1); scaling "strong": ms_one_gpu is the WHOLE frame on one GPU;
    "weak": ms_one_gpu is one GPU's own 1/world of the frame (its N = 1 step)."""
    import torch.distributed as tdist
    from tools.sharded_standins import LoopbackGroup
    from wct_hip.sharded import ShardedStylizer
    real = None
    try:
        if not tdist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29591")
            tdist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", torch.cuda.current_device()))
        real = tdist
    except Exception as e:      # noqa: BLE001  -- the simulation still runs, with the collectives as no-ops
        real = None
        note = "1-rank RCCL communicator unavailable (%s): collectives not launched" % type(e).__name__
    else:
        note = "all_reduce / owned broadcasts launched on a 1-rank RCCL communicator (launch cost, no link time)"
    wct.style_prepare(style)
    stats = {L: wct.style_export(L).clone() for L in (5, 4, 3, 2, 1)}
    torch.cuda.synchronize()
    res = {"world": world, "frame": "%dx%d" % (Wf, Hf), "scaling": scaling, "halo_mode": None, "collectives": note, "ranks": {}}
    hs, ws = int(style.shape[-2]), int(style.shape[-1])
    # the frame as ONE library call with RCCL inside (wct_stylize_sharded) needs a communicator on the engine: one rank here, given the
    # GEOMETRY of rank r of the job by debug key "shard_emulate" (WCT_DEBUG-gated: its peers are itself -- right work, other numbers)
    c_ok = False
    if real is not None:
        try:
            if not getattr(wct, "has_comm", False):
                wct.comm_init(real)
            wct.comm_selftest()
            c_ok = True
        except Exception as e:      # noqa: BLE001
            res["c_cascade_error"] = repr(e)
    had_debug = os.environ.get("WCT_DEBUG")
    for r in ranks:
        entry = {}
        for tag, smode, c_cas in (("torch_distributed_owner", "owner", False), ("torch_distributed", "strips", False), ("c_cascade_owner", "owner", True),
                                  ("c_cascade", "strips", True)):
            if c_cas and not c_ok:
                continue
            grp = LoopbackGroup(r, world, real)
            grp.style_stats = stats
            if c_cas:
                os.environ["WCT_DEBUG"] = "1"
                wct.debug_set("shard_emulate", 100 * world + r)
            try:
                sh = ShardedStylizer(wct, grp, Hf, Wf, hs, ws, halo_mode="auto", c_collectives=False, style_mode=smode, c_cascade=c_cas, fast_fold=c_cas)
                res["halo_mode"] = sh.halo_mode
                x0, x1 = sh.input_columns()
                strip = strip_of(x0, x1)
                for _ in range(2):
                    sh.stylize_strip(strip, style)
                torch.cuda.synchronize()
                sh.t_range_wait = 0.0
                t0 = time.perf_counter()
                for _ in range(frames):
                    out = sh.stylize_strip(strip, style)
                t1 = time.perf_counter()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
            finally:
                if c_cas:
                    wct.debug_set("shard_emulate", 0)
                    if had_debug is None:
                        os.environ.pop("WCT_DEBUG", None)
                    else:
                        os.environ["WCT_DEBUG"] = had_debug
            assert bool(torch.isfinite(out).all())
            ms = (t2 - t0) / frames * 1e3
            host, wait = (t1 - t0) / frames * 1e3, sh.t_range_wait / frames * 1e3
            entry[tag] = {"style_side": smode, "ms_per_frame": round(ms, 3), "host_enqueue_ms": round(host, 3), "range_flag_wait_ms": round(wait, 3),
                          "pure_enqueue_ms": round(host - wait, 3), "pure_enqueue_share_of_frame": round((host - wait) / ms, 3)}
            entry.update({"columns_in": x1 - x0, "columns_owned": sh.own[1] - sh.own[0]})
            del strip
        res["ranks"][str(r)] = entry
    if c_ok:
        wct.comm_destroy()
    eff = lambda w_: round(ms_one_gpu / w_ / world, 3) if scaling == "strong" else round(ms_one_gpu / w_, 3)     # noqa: E731
    # a job runs ONE arrangement on all its ranks: its frame time is its SLOWEST simulated rank; the prediction is the best arrangement's
    tags = sorted({k for e in res["ranks"].values() for k, v in e.items() if isinstance(v, dict)})
    slowest = {t: max(e[t]["ms_per_frame"] for e in res["ranks"].values() if t in e) for t in tags}
    res["slowest_rank_ms_by_arrangement"] = {t: round(v, 3) for t, v in slowest.items()}
    res["predicted_efficiency_by_arrangement"] = {t: eff(v) for t, v in slowest.items()}
    best_tag = min((t for t in tags if t != "torch_distributed_owner"), key=lambda t: slowest[t])
    res["arrangement"] = best_tag
    worst = slowest[best_tag]
    for e in res["ranks"].values():
        e["ms_per_frame"] = e[best_tag]["ms_per_frame"]
    res["predicted_efficiency_round5_arrangement"] = eff(slowest["torch_distributed_owner"])     # Python orchestration over torch.distributed, style levels dealt out whole
    res["predicted_ms_per_frame"] = round(worst, 3)
    res["predicted_MPs"] = round(Hf * Wf / 1e6 / worst * 1e3, 1)
    if scaling == "strong":
        res["predicted_speedup_vs_1gpu"] = round(ms_one_gpu / worst, 2)
        res["predicted_efficiency"] = round(ms_one_gpu / worst / world, 3)
    else:
        res["predicted_efficiency"] = round(ms_one_gpu / worst, 3)
    res["note"] = "compute + orchestration of the slowest simulated rank; xGMI transfer time (<= 3.5 MB per exchange, 132 KB per all-reduce) and rank skew not included"
    return res


def finish_rank_sims():
    import torch.distributed as tdist
    if tdist.is_available() and tdist.is_initialized():
        tdist.destroy_process_group()


if __name__ == "__main__":
    main()
