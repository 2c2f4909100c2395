"""GPU parity tests at BENCHMARK scale: the HIP cascade end to end at BASELINE.json's own sizes.

THE GATE (one rule; the same in bench.py and BASELINE.md 3.5 / DESIGN.md 2).  Every full-size frame is compared with THE REFERENCE'S
OWN PIXELS for that frame -- fixtures made by running the reference itself (util_wct.WCT, real 16x checkpoints, torch CPU) in the
build container, tools/make_goldens.py:
    G13  the two SYNTHETIC config-2 frames (SURVEY 8d seeds: numpy default_rng; `noise` = bench.py's timed frame, `smooth`)
    G11  the reference's UHD sample pair (green_park 3840x2160 + style/in1.jpg 2048x2048)
    G14  config 3: the reference's un-pruned classes with GENERATED weights (the torch7 checkpoints are absent)
each holding a 1/16 lattice of the output + crops + a 16x-downsampled image (tests/fixture_compare.py), and
    hip_vs_reference <= max(1e-3, 1.25 * oracle_vs_reference)        max|d| / max|reference| over every reference pixel held
i.e. the north_star's 1e-3 wherever a second valid fp32 implementation of the reference's arithmetic (the oracle: the same op
sequence in C loops) itself reproduces the reference to 1e-3, and otherwise no further from the reference than that (+25 %).
Measured in the build container (oracle vs reference): noise 6.0e-4, smooth 7.7e-4, UHD pair 1.2e-5 -> the limit IS 1e-3 on all
config-2 frames; config 3 (generated weights, five whitenings of random 512-channel stacks: chaotic) 2.3e-3 -> limit 2.9e-3
there, with the level-isolated comparison as the sharp check.

Beside the gate, an error budget against the exact result (test_error_budget_vs_fp64_truth, 1080p): "truth" = wct_oracle with
precision="fp64", the reference's algorithm with every activation and accumulation in fp64 (torch's own fp64 convolutions agree
with it to 1.3e-12); every fp32 implementation sits 4e-4 .. 7e-4 from it on uniform noise, this library included.
Everything that can be compared level-isolated (each level on the oracle's own input) is held 10x tighter, here and in
tests/test_hip_parity.py.
"""
import os
import time
import types

import numpy as np
import pytest

from tests.conftest import rel_err
from tests.fixture_compare import GATE, cfg2_frames, cfg3_frames, compare_to_fixture, smooth_frame as _smooth_frame
from wct_hip import model_zoo

pytestmark = pytest.mark.gpu

# GATE = 1e-3: BASELINE.json north_star, 1e-3 relative per-pixel tolerance, as max|d| / max|ref|


def gate_limit(oracle_vs_reference):
    return max(GATE, 1.25 * oracle_vs_reference)


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


smooth = _smooth_frame


def _threads(oracle):
    oracle.set_num_threads(min(os.cpu_count() or 1, 32))   # the C convolutions stop scaling beyond ~32 threads (bench.py)


def _report(name, **kv):
    print("\n[%s] %s" % (name, "  ".join("%s=%s" % (k, ("%.3e" % v) if isinstance(v, float) else v) for k, v in kv.items())))


# --------------------------------------------------------------------------- config 2 at full size, end to end
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_e2e_config2_full_size(torch_cuda, wct16, oracle, weights16x, golden, kind):
    """BASELINE configs[1] exactly as bench.py times it: --mode 16x, 3840x2160 content + 2048x2048 style, 5 levels, alpha = 1,
    the SURVEY 8(d) frames (numpy seeds 1 / 2; `smooth`: dead channels, ill-conditioned covariances) -- the HIP cascade against
    the REFERENCE'S OWN OUTPUT on that frame (G13) under the one gate of this module, the oracle beside it."""
    torch = torch_cuda
    _threads(oracle)
    g = golden("g13_cfg2_%s.npz" % kind)
    c, s = cfg2_frames(kind)
    assert abs(float(c.sum(dtype=np.float64)) - float(g["content.checksum"])) < 1e-6      # the seeds reproduce the reference's input
    assert abs(float(s.sum(dtype=np.float64)) - float(g["style.checksum"])) < 1e-6
    t0 = time.time()
    ref = oracle.stylize(oracle.Modules("16x", weights16x), c, s, 1.0)
    t1 = time.time()
    wct16.saturation_count(reset=True)
    got = wct16.stylize(cu(torch, c), cu(torch, s)).cpu().numpy()[0]
    assert wct16.saturation_count() == 0
    assert got.shape == ref.shape == (3, 2160, 3840)
    rh, ro = compare_to_fixture(got, g), compare_to_fixture(ref, g)
    _report("e2e cfg2 " + kind, hip_vs_reference=rh["max"], oracle_vs_reference=ro["max"], limit=gate_limit(ro["max"]),
            hip_p9999=rh["lattice_p9999"], oracle_p9999=ro["lattice_p9999"], hip_frac_gt_1e3=rh["lattice_frac_gt_gate"],
            hip_down16=rh["down16_max"], hip_vs_oracle=rel_err(got, ref), reference_pixels=rh["lattice_pixels"], oracle_s=round(t1 - t0, 1))
    assert ro["max"] <= GATE                                   # the oracle reproduces the reference on this frame: the limit is 1e-3
    assert rh["max"] <= gate_limit(ro["max"])                  # THE GATE
    assert rh["down16_max"] <= GATE / 4 and rh["mean_diff"] <= 1e-5
    assert rh["lattice_p9999"] <= GATE / 2                     # and it is not a near miss everywhere: 99.99 % of the pixels within 5e-4


def test_e2e_config2_exact_fp32_mode_under_the_gate(torch_cuda, weights16x, golden):
    """The reference's own arithmetic class (VERDICT r4 task 3): `wct_set_conv_mode(0)` -- exact-fp32 MFMA products in every convolution, as the
    reference's plain fp32 nn.Conv2d (model_cd.py:724-743) -- on the TIMED frame against the reference's own pixels (G13 noise) at the literal
    1e-3, beside the f16x3 default (test_e2e_config2_full_size).  bench.py times the same mode as passes.cfg2_fp32_exact."""
    torch = torch_cuda
    from wct_hip import WCT
    g = golden("g13_cfg2_noise.npz")
    c, s = cfg2_frames("noise")
    assert abs(float(c.sum(dtype=np.float64)) - float(g["content.checksum"])) < 1e-6
    eng = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
    eng.set_conv_mode("fp32")
    got = eng.stylize(cu(torch, c), cu(torch, s)).cpu().numpy()[0]
    assert eng.saturation_count() == 0          # (exact fp32 has no range limit: the counter cannot move)
    r = compare_to_fixture(got, g)
    _report("e2e cfg2 noise, exact-fp32 mode", hip_vs_reference=r["max"], hip_p9999=r["lattice_p9999"], hip_frac_gt_1e3=r["lattice_frac_gt_gate"], limit=GATE)
    assert r["max"] <= GATE and r["lattice_p9999"] <= GATE / 2 and r["down16_max"] <= GATE / 4


def test_e2e_config2_reference_uhd_pair(torch_cuda, wct16, oracle, weights16x, golden):
    """Config 2 on the reference's OWN sample data: content/UHD_content/green_park-wallpaper-3840x2160.jpg (`--UHD`,
    README.md:41-44) + style/in1.jpg (2048x2048) -- the JPEG files are committed as fixtures, the expected output comes from
    the reference itself (tools/make_goldens.py gen_g11: util_wct.WCT, real checkpoints, torch CPU; 16x-downsampled image,
    four 96x96 crops, mean / std / max).  On natural images the cascade is far better conditioned than on noise (the oracle
    sits 5e-5 from the fp64 truth at 1080p), so here the north_star gate is asserted against the oracle AND against the
    reference's own pixels, with room to spare.  uint8 in, like the reference's harness (ToTensor, data_loader.py:57-58)."""
    from PIL import Image
    from tests.conftest import GOLD
    torch = torch_cuda
    _threads(oracle)
    g = golden("g11_uhd_pair.npz")
    c_u8 = np.array(Image.open(os.path.join(GOLD, "g11_uhd_content_3840x2160.jpg")).convert("RGB"))
    s_u8 = np.array(Image.open(os.path.join(GOLD, "g11_style_2048x2048.jpg")).convert("RGB"))
    assert c_u8.shape == (2160, 3840, 3) and s_u8.shape == (2048, 2048, 3)
    c, s = oracle.to_tensor_u8(c_u8), oracle.to_tensor_u8(s_u8)
    t0 = time.time()
    ref = oracle.stylize(oracle.Modules("16x", weights16x), c, s, 1.0)
    t1 = time.time()
    wct16.saturation_count(reset=True)
    got = wct16.stylize(wct16.to_tensor_u8(torch.from_numpy(c_u8)), wct16.to_tensor_u8(torch.from_numpy(s_u8))).cpu().numpy()[0]
    assert wct16.saturation_count() == 0
    mx = float(g["max"])
    nerr = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / mx)     # noqa: E731  normalised by the reference's max
    down = lambda y: y.reshape(3, 135, 16, 240, 16).mean(axis=(2, 4), dtype=np.float64)   # noqa: E731
    crops_hip, crops_or = [], []
    for i in range(4):
        y0, x0 = (int(v) for v in g["crop%d.origin" % i])
        crops_hip.append(nerr(got[:, y0:y0 + 96, x0:x0 + 96], g["crop%d" % i]))
        crops_or.append(nerr(ref[:, y0:y0 + 96, x0:x0 + 96], g["crop%d" % i]))
    e = rel_err(got, ref)
    _report("e2e cfg2 reference UHD pair", hip_vs_oracle=e, hip_vs_reference_crops=max(crops_hip), oracle_vs_reference_crops=max(crops_or),
            hip_vs_reference_down16=nerr(down(got), g["down16"]), oracle_vs_reference_down16=nerr(down(ref), g["down16"]),
            mean_ref=float(g["mean"]), mean_hip=float(got.mean(dtype=np.float64)), oracle_s=round(t1 - t0, 1))
    assert got.shape == ref.shape == tuple(g["shape"])
    assert max(crops_or) <= GATE and nerr(down(ref), g["down16"]) <= GATE / 4          # the oracle against the reference's pixels
    assert e <= GATE and np.allclose(got, ref, rtol=GATE, atol=GATE * float(np.abs(ref).max()))
    assert max(crops_hip) <= GATE and nerr(down(got), g["down16"]) <= GATE / 4         # this library against the reference's pixels
    assert abs(float(got.mean(dtype=np.float64)) - float(g["mean"])) <= 1e-5 and abs(float(got.max()) - mx) <= GATE * mx


# --------------------------------------------------------------------------- error budget against fp64 truth
@pytest.mark.parametrize("kind", ["noise", "smooth"])
def test_error_budget_vs_fp64_truth(torch_cuda, oracle, weights16x, kind):
    """1920x1080 content + 1024x1024 style (bench.py's former CPU sample).  truth = oracle with precision "fp64";
    ref32 = the oracle as the reference computes (fp32 convs); hip = f16x3 (default) and exact-fp32 MFMA modes.
    Cumulative error after every level of the cascade, and the end-to-end figures:
        |hip - truth| must stay within 1.5 x |ref32 - truth| + 1e-4   (the HIP path is in the reference's own accuracy class)
        |hip - ref32| <= 1e-3                                       (north_star gate at this size)"""
    from wct_hip import WCT
    torch = torch_cuda
    _threads(oracle)
    rng = np.random.default_rng(0)
    c = rng.random((3, 1080, 1920), dtype=np.float32) if kind == "noise" else smooth(rng, (3, 1080, 1920))
    s = rng.random((3, 1024, 1024), dtype=np.float32)
    m32, m64 = oracle.Modules("16x", weights16x), oracle.Modules("16x", weights16x, precision="fp64")
    t32, t64 = [], []
    r32 = oracle.stylize(m32, c, s, 1.0, trace=t32)
    r64 = oracle.stylize(m64, c, s, 1.0, trace=t64)
    ref_truth = rel_err(r32, r64)
    cum_ref = [rel_err(a["out"], b["out"]) for a, b in zip(t32, t64)]
    res = {}
    for mode in ("f16x3", "fp32"):
        wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
        wct.set_conv_mode(mode)
        img, cum = cu(torch, c), []
        for L, b in zip((5, 4, 3, 2, 1), t64):     # the cascade level by level (own outputs chained), error vs truth after each
            img = wct.style_transfer_level(L, img, cu(torch, s))
            cum.append(rel_err(img.cpu().numpy()[0], b["out"]))
        got = wct.stylize(cu(torch, c), cu(torch, s)).cpu().numpy()[0]
        assert np.array_equal(got, img.cpu().numpy()[0])      # wct_stylize == its levels chained
        res[mode] = (rel_err(got, r64), rel_err(got, r32), cum)
        assert wct.saturation_count() == 0
    _report("budget " + kind, ref32_vs_truth=ref_truth, hip_vs_truth=res["f16x3"][0], hip_fp32mode_vs_truth=res["fp32"][0],
            hip_vs_ref32=res["f16x3"][1], hip_fp32mode_vs_ref32=res["fp32"][1])
    _report("budget " + kind + " cumulative L5..L1", ref32=" ".join("%.1e" % v for v in cum_ref),
            hip=" ".join("%.1e" % v for v in res["f16x3"][2]), hip_fp32mode=" ".join("%.1e" % v for v in res["fp32"][2]))
    for mode in ("f16x3", "fp32"):
        assert res[mode][0] <= 1.5 * ref_truth + 1e-4, (mode, res[mode][0], ref_truth)
        assert res[mode][0] <= GATE
    assert res["f16x3"][1] <= GATE     # at this size the comparison against the oracle itself still fits the gate


# --------------------------------------------------------------------------- config 3: --mode original at 1920x1080
def test_e2e_config3_original_mode(torch_cuda, oracle, golden):
    """BASELINE configs[2]: --mode original (un-pruned VGG-19 graph, 512/512/256/128/64 channels), 1920x1080 content and
    style -> 1920x1072 output.  The torch7 checkpoints are absent from the reference snapshot (README.md:26), so the
    weights are the generated set model_zoo.synth_weights("original", 3): real-weight parity is UNPINNED.
    Expected output: G14 = the reference's own classes (model_original.py Encoder/Decoder{1..5}) + util_wct.WCT.transform on
    this frame with these weights.  With random 512-channel stacks the cascade is chaotic at the level of the reference's OWN
    arithmetic (oracle vs reference 2.3e-3; oracle vs its fp64 arm 1e-3 .. 4e-3), so the sharp check is level-isolated:
      per level, on the truth's level input   |hip - truth| <= 1.5 |oracle - truth| + 2e-5, and <= 5e-4
      end to end, the module's gate            hip_vs_reference <= max(1e-3, 1.25 * oracle_vs_reference)"""
    from wct_hip import WCT
    torch = torch_cuda
    _threads(oracle)
    g = golden("g14_cfg3_original.npz")
    w = model_zoo.synth_weights("original", 3)
    c, s = cfg3_frames()
    m32, m64 = oracle.Modules("original", w), oracle.Modules("original", w, precision="fp64")
    wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
    t64 = []
    t0 = time.time()
    ref = oracle.stylize(m32, c, s, 1.0)
    t1 = time.time()
    truth = oracle.stylize(m64, c, s, 1.0, trace=t64)
    t2 = time.time()
    iso_hip, iso_ref, img = [], [], c
    for t in t64:
        x32 = np.asarray(img, np.float32)
        y = wct.style_transfer_level(t["level"], cu(torch, x32), cu(torch, s)).cpu().numpy()[0]
        iso_hip.append(rel_err(y, t["out"]))
        iso_ref.append(rel_err(oracle.style_transfer(m32, t["level"], x32, s, 1.0), t["out"]))
        img = t["out"]
    got = wct.stylize(cu(torch, c), cu(torch, s)).cpu().numpy()[0]
    rh, ro = compare_to_fixture(got, g), compare_to_fixture(ref, g)
    _report("e2e cfg3 original", hip_vs_reference=rh["max"], oracle_vs_reference=ro["max"], limit=gate_limit(ro["max"]),
            hip_p9999=rh["lattice_p9999"], oracle_p9999=ro["lattice_p9999"], hip_vs_truth=rel_err(got, truth), oracle_vs_truth=rel_err(ref, truth),
            iso_hip=" ".join("%.1e" % v for v in iso_hip), iso_oracle=" ".join("%.1e" % v for v in iso_ref),
            oracle_s=round(t1 - t0, 1), truth_s=round(t2 - t1, 1), saturated=wct.saturation_count())
    assert got.shape == ref.shape == (3, 1072, 1920)
    assert wct.saturation_count() == 0
    for a, b in zip(iso_hip, iso_ref):
        assert a <= 1.5 * b + 2e-5 and a <= 5e-4
    assert rh["max"] <= gate_limit(ro["max"])                  # THE GATE (the limit is 1.25 x oracle_vs_reference here)
    assert rh["lattice_p9999"] <= 1.5 * ro["lattice_p9999"] + 1e-4


# --------------------------------------------------------------------------- config 3 graph at the LITERAL 1e-3 (G15)
G15_ORACLE_LIMIT = 2.5e-4     # VERDICT r3 task 1b: a frame on which the reference's own arithmetic is not chaotic


@pytest.mark.parametrize("kind", ["noise", "natural"])
def test_e2e_config3_graph_strict_gate(torch_cuda, oracle, golden, kind):
    """The un-pruned graph (`--mode original`, 1920x1080 -> 1920x1072, five levels) END TO END at the north_star's literal 1e-3
    against the reference's own pixels -- G15: model_original.py's Encoder/Decoder{1..5} + util_wct.WCT.transform run on
    model_zoo.synth_weights_conditioned("original", 15) (paired-isometry layers: every channel active on half of the pixels, gain
    ~1 per layer; see its docstring for why He-uniform stacks -- G14 -- are chaotic under the reference's OWN fp32 arithmetic).
    Frames: `noise` = config 3's seeds, `natural` = the reference's UHD sample pair resized to 1920x1080.  On these the oracle (a
    second fp32 implementation of the reference's op sequence) must itself sit within 2.5e-4 of the reference, so the 1e-3 gate is
    unconditional -- no `1.25 x oracle` clause.  The 3->64 first conv runs in exact-fp32 MFMA by default (in3_wide_f32_kernel); the
    same frame is also run with it in f16x3 (debug key in3wide = 1) and on the generic fp32 kernel (0): the A/B of VERDICT r3 task
    1a -- at K = 27, behind conv0's `255 x - mean` fold, split-f16 operands are measurably further from the reference (3.4e-4
    against 2.3e-4 here, 2.29e-3 against 1.58e-3 on G14), which is why that one write-bound layer went back to fp32 products; the
    two fp32 forms must agree closely (same products, another summation order)."""
    from tests.conftest import GOLD
    from tests.fixture_compare import cfg3_natural_frames
    from wct_hip import WCT
    torch = torch_cuda
    _threads(oracle)
    g = golden("g15_cfg3_conditioned_%s.npz" % kind)
    w = model_zoo.synth_weights_conditioned("original", 15)
    assert abs(sum(float(np.abs(v).sum(dtype=np.float64)) for v in w.values()) - float(g["weights.checksum"])) < 1e-6 * float(g["weights.checksum"])
    c, s = cfg3_frames() if kind == "noise" else cfg3_natural_frames(GOLD)
    assert abs(float(c.sum(dtype=np.float64)) - float(g["content.checksum"])) < 1e-6      # the inputs are the reference run's inputs
    assert abs(float(s.sum(dtype=np.float64)) - float(g["style.checksum"])) < 1e-6
    t0 = time.time()
    ref = oracle.stylize(oracle.Modules("original", w), c, s, 1.0)
    t1 = time.time()
    ro = compare_to_fixture(ref, g)
    wct = WCT(types.SimpleNamespace(mode="original", alpha=1.0), weights=w)
    res = {}
    for tag, key in (("default", 2), ("in3_f16x3", 1), ("in3_fp32_generic", 0)):
        wct.debug_set("in3wide", key)
        wct.saturation_count(reset=True)
        got = wct.stylize(cu(torch, c), cu(torch, s)).cpu().numpy()[0]
        assert wct.saturation_count() == 0
        assert got.shape == ref.shape == (3, 1072, 1920)
        res[tag] = compare_to_fixture(got, g)
        res[tag]["vs_oracle"] = rel_err(got, ref)
    wct.debug_set("in3wide", 2)
    rh = res["default"]
    _report("e2e cfg3 graph, conditioned weights, " + kind, hip_vs_reference=rh["max"], oracle_vs_reference=ro["max"], limit=GATE,
            hip_p9999=rh["lattice_p9999"], oracle_p9999=ro["lattice_p9999"], hip_frac_gt_1e3=rh["lattice_frac_gt_gate"],
            hip_down16=rh["down16_max"], hip_vs_oracle=rh["vs_oracle"], in3_f16x3_vs_reference=res["in3_f16x3"]["max"],
            in3_f16x3_p9999=res["in3_f16x3"]["lattice_p9999"], in3_fp32_generic_vs_reference=res["in3_fp32_generic"]["max"], oracle_s=round(t1 - t0, 1))
    assert ro["max"] <= G15_ORACLE_LIMIT                       # the reference's arithmetic is well-conditioned on this frame
    assert rh["max"] <= GATE                                   # THE GATE, literal
    assert res["in3_f16x3"]["max"] <= GATE and res["in3_fp32_generic"]["max"] <= GATE
    assert abs(rh["max"] - res["in3_fp32_generic"]["max"]) <= 0.2 * rh["max"]     # the two exact-fp32 forms of the first conv: same error class
    assert rh["down16_max"] <= GATE / 4 and rh["lattice_p9999"] <= GATE / 2


# --------------------------------------------------------------------------- config 4 size on one GPU (properties)
def test_config4_size_single_gpu_properties(torch_cuda, wct16):
    """BASELINE configs[3]'s content size, 10240x4096 (+ the 2048x2048 style), untiled on ONE MI355X -- the north_star's
    target configuration.  The oracle would need ~5 min and 60 GB here, so size-independent properties: finite, >= 0 (every
    decoder ends in a ReLU, model_cd.py:293), deterministic bit for bit, no f16x3 saturation, device memory < 16 GB, and the
    left 2560 columns of the result equal (to the cascade's own amplification of the statistics' round-off) a sharded
    computation's left strip -- see tests/test_sharded_gpu.py for the strips themselves."""
    torch = torch_cuda
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(5)
    c = torch.rand((1, 3, 4096, 10240), device="cuda", generator=g)
    s = torch.rand((1, 3, 2048, 2048), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2))
    free0, total = torch.cuda.mem_get_info()
    wct16.saturation_count(reset=True)
    a = wct16.stylize(c, s).clone()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    b = wct16.stylize(c, s)
    assert tuple(a.shape) == (1, 3, 4096, 10240)
    assert bool(torch.isfinite(a).all()) and float(a.min()) >= 0.0
    assert torch.equal(a, b)
    assert wct16.saturation_count() == 0
    used = (free0 - free1) / 2**30 + a.numel() * 4 / 2**30   # library workspace growth + the result clone
    _report("cfg4 single GPU", workspace_GiB=round((free0 - free1) / 2**30, 2), max=float(a.max()), mean=float(a.mean()))
    assert used < 16.0


# --------------------------------------------------------------------------- num_run > 1 (WCT.py:120)
def test_num_run_2(torch_cuda, wct16, oracle, weights16x):
    """`--num_run 2` (WCT.py:120: the whole 5-level cascade repeated on its own output).  wct_stylize's ping-pong buffers
    (`which = (5 * num_run) & 1`) against chained calls, bit for bit, for 2 and 3 runs and for the prepared-style form; run 1
    end to end against wct_oracle.stylize; run 2 level-isolated against the oracle's own run-2 trace (a stylised image
    re-stylised is chaotic end to end -- 4e-2 between the oracle and this library, as between any two fp32 implementations --
    so each of its levels is checked on the oracle's input to that level)."""
    torch = torch_cuda
    rng = np.random.default_rng(12)
    c, s = smooth(rng, (3, 200, 264)), rng.random((3, 160, 152), dtype=np.float32)
    mods = oracle.Modules("16x", weights16x)
    trace = []
    ref2 = oracle.stylize(mods, c, s, 1.0, num_run=2, trace=trace)
    ref1 = trace[4]["out"]
    one = wct16.stylize(cu(torch, c), cu(torch, s)).clone()
    two = wct16.stylize(cu(torch, c), cu(torch, s), num_run=2).clone()
    chained = wct16.stylize(one, cu(torch, s)).clone()
    three = wct16.stylize(cu(torch, c), cu(torch, s), num_run=3).clone()
    assert two.shape == one.shape and torch.equal(two, chained)
    assert torch.equal(three, wct16.stylize(chained, cu(torch, s)))
    wct16.style_prepare(cu(torch, s))
    assert torch.equal(wct16.stylize_prepared(cu(torch, c), num_run=2), two)
    e1 = rel_err(one.cpu().numpy()[0], ref1)
    iso, img = [], ref1
    for t in trace[5:]:
        g = wct16.style_transfer_level(t["level"], cu(torch, img), cu(torch, s)).cpu().numpy()[0]
        iso.append(rel_err(g, t["out"]))
        img = t["out"]
    _report("num_run", run1=e1, run2_level_isolated=" ".join("%.1e" % v for v in iso), run2_e2e=rel_err(two.cpu().numpy()[0], ref2))
    assert e1 <= GATE and max(iso) <= 2e-4 and ref2.shape == tuple(two.shape[1:])


# --------------------------------------------------------------------------- f16x3 range: never silent
def test_f16x3_saturation_is_flagged(torch_cuda, weights16x):
    """The f16x3 kernels clamp activations to +-65504; with weights scaled so that relu1_2 of the level-3 encoder reaches
    ~1e6 the flag must come up -- OverflowError from sync() ONCE (reported and cleared: a later error means a later clamp),
    OverflowError from the NEXT compute call of any kind without a sync (the asynchronous read-back of wct_range_poll), a
    non-zero saturation_count -- and the exact-fp32 mode must run the same weights without it and agree with the oracle.
    A NaN pixel (external data) is turned into a finite value by the same clamp instruction and must raise it too."""
    from wct_hip import WCT
    torch = torch_cuda
    w = dict(weights16x)
    w["e3.conv12.weight"] = w["e3.conv12.weight"] * np.float32(3e4)
    w["e3.conv12.bias"] = w["e3.conv12.bias"] * np.float32(3e4)
    wct = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=w)
    g = torch.Generator(device="cuda").manual_seed(8)
    c = torch.rand((1, 3, 96, 128), device="cuda", generator=g)
    assert wct.saturation_count() == 0
    wct.e2(c)
    wct.sync()                                   # level 2 uses other weights: nothing flagged
    y = wct.e3(c)
    assert bool(torch.isfinite(y).all())          # saturates, never inf (this also waits for the call: its read-back has landed)
    with pytest.raises(OverflowError):            # ... so the next call of ANY kind reports it, without a sync()
        wct.e2(c)
    with pytest.raises(OverflowError):
        wct.stylize(c, c)
    assert wct.saturation_count() > 0             # reading does not clear
    with pytest.raises(OverflowError):
        wct.sync()                                # reported once ...
    wct.sync()                                    # ... and cleared
    assert wct.saturation_count() == 0
    wct.e2(c)
    wct.strict_range = False                      # opt out of the per-call check: only sync() / saturation_count() tell
    wct.e3(c)
    torch.cuda.synchronize()
    wct.e3(c)
    assert wct.saturation_count(reset=True) > 0
    wct.strict_range = True
    wct.set_conv_mode("fp32")
    y32 = wct.e3(c)
    wct.sync()
    assert wct.saturation_count() == 0
    from oracle import wct_oracle
    ref = wct_oracle.Modules("16x", w).encode(3, c.cpu().numpy()[0])
    assert rel_err(y32.cpu().numpy()[0], ref) < 2e-5
    # the fused head (conv11 + conv12 + pool) and a huge image value both raise it too
    wct.set_conv_mode("f16x3")
    wct.e5(c * 1e6)
    assert wct.saturation_count(reset=True) > 0
    # NaN in external data: one NaN pixel in the image (fused head, level-1 kernels), one NaN in an fp32 feature map (decoder)
    clean = WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=weights16x)
    clean.strict_range = False
    bad = c.clone()
    bad[0, 1, 40, 50] = float("nan")
    for level in (5, 1):
        clean.encode(level, bad)
        assert clean.saturation_count(reset=True) > 0, level
    clean.stylize(bad, c)
    assert clean.saturation_count(reset=True) > 0
    f = clean.encode(3, c)
    assert clean.saturation_count() == 0
    f[0, 7, 3, 5] = float("nan")
    clean.decode(3, f)
    assert clean.saturation_count(reset=True) > 0
    with pytest.raises(ValueError):               # non-finite weights never get as far as a kernel
        wn = dict(weights16x)
        wn["d2.conv21.weight"] = wn["d2.conv21.weight"].copy()
        wn["d2.conv21.weight"][0, 0, 0, 0] = np.float32("inf")
        WCT(types.SimpleNamespace(mode="16x", alpha=1.0), weights=wn)


# --------------------------------------------------------------------------- reference checkpoints through WCT(args)
def test_reference_pth_layout_through_args(torch_cuda, wct16, weights16x, tmp_path):
    """`WCT(args)` as WCT.py:97 builds it: ten `.pth` files in the reference's own layout (trained_models/wct_se_16x_new/
    {1..5}SE.pth = {"epoch", "model": state_dict with unused conv{k}1_aux heads}, wct_se_16x_new_sd/{1..5}SD.pth;
    model_cd.py:712-718) -> the same device state as the packaged blob converted from them."""
    import torch
    from wct_hip import WCT
    args = types.SimpleNamespace(mode="16x", alpha=1.0)
    for k in range(1, 6):
        for kind, key, name in (("enc", "e%d" % k, "%dSE.pth" % k), ("dec", "d%d" % k, "%dSD.pth" % k)):
            sd = {n[len(key) + 1:]: torch.from_numpy(v.copy()) for n, v in weights16x.items() if n.startswith(key + ".")}
            if kind == "enc":
                sd["conv%d1_aux.weight" % k] = torch.zeros(8, 4, 1, 1)      # training-only head: must be ignored
                payload = {"epoch": 20, "model": sd}
            else:
                payload = sd if k % 2 else {"epoch": 20, "model": sd}        # bare state_dict or wrapped (model_cd.py:714-717)
            path = str(tmp_path / name)
            torch.save(payload, path)
            setattr(args, key, path)
    w = WCT(args)
    g = torch.Generator(device="cuda").manual_seed(6)
    c, s = torch.rand((1, 3, 80, 112), device="cuda", generator=g), torch.rand((1, 3, 64, 72), device="cuda", generator=g)
    assert torch.equal(w.stylize(c, s), wct16.stylize(c, s))
    os.remove(args.d3)
    with pytest.raises(FileNotFoundError):        # a partial set is an error, never a silent substitute
        WCT(args)


def test_out_buffer_validation(torch_cuda, wct16):
    torch = torch_cuda
    c, s = torch.rand((1, 3, 64, 64), device="cuda"), torch.rand((1, 3, 48, 48), device="cuda")
    with pytest.raises(ValueError):
        wct16.stylize(c, s, out=torch.empty(3 * 64 * 64 - 1, device="cuda"))
    with pytest.raises(ValueError):
        wct16.stylize(c, s, out=torch.empty(3 * 64 * 64, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        wct16.stylize(c, s, out=torch.empty(3 * 64 * 64))
    f = wct16.encode(3, c, layout="nhwc")
    with pytest.raises(ValueError):
        wct16.moments(f.permute(0, 3, 1, 2)[0, 0])          # 2-D: not an NHWC feature
    n, sm, sq = wct16.moments(f)
    with pytest.raises(ValueError):
        wct16.solve(n, sm, sq[:, :-1], n, sm, sq)
    # a permuted (non-contiguous) NHWC view and an fp64 tensor are normalised, not misread
    n2, sm2, sq2 = wct16.moments(f.double().permute(0, 2, 1, 3).permute(0, 2, 1, 3))
    assert torch.equal(sm, sm2) and torch.equal(sq, sq2)
