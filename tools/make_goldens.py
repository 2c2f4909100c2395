#!/usr/bin/env python3
"""Generate the torch-free weight blob and the golden vectors from the REFERENCE itself.

Runs only in the build container (needs /root/reference and torch-CPU).  Nothing of the
reference's source travels: the outputs are *data* --
  collaborative-distillation_amd/weights/16x.npz   the ten 16x checkpoints, aux heads dropped
  tests/golden/*.npz                               inputs + expected outputs (G1..G11)
  tests/golden/g11_*.jpg                           the reference's UHD sample content + 2048x2048 sample style (data files)

The reference is imported unmodified with the two shims of SURVEY 8c:
  * torch.utils.serialization.load_lua (removed in torch>=1.0) is injected as a stub;
  * empty `torchvision` / `torchvision.transforms` modules are registered (imported, never used
    by util_wct.py).
`WCT.py` itself is a script (argparse + .cuda()), so its 9-line styleTransfer()/cascade
(WCT.py:98-106, 120-125) is restated in `ref_style_transfer` below, on CPU.

usage: PYTHONDONTWRITEBYTECODE=1 python tools/make_goldens.py
"""
import os
import sys
import types

os.environ.setdefault("PYTHONDONTWRITEBYTECODE", "1")
sys.dont_write_bytecode = True

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
GOLD = os.path.join(REPO, "tests", "golden")
sys.path.insert(0, os.path.join(REPO, "collaborative-distillation_amd"))
from wct_hip import model_zoo  # noqa: E402


def import_reference():
    import torch.utils.serialization as ser

    def _no_lua(*a, **k):
        raise RuntimeError("load_lua is not available (torch>=1.0)")

    ser.load_lua = _no_lua
    for name in ("torchvision", "torchvision.transforms"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, os.path.join(REF, "PytorchWCT"))
    os.chdir(os.path.join(REF, "PytorchWCT"))
    import util_wct  # noqa
    return util_wct


def ref_args(mode="16x", alpha=1.0):
    a = types.SimpleNamespace(mode=mode, numpy=False, alpha=alpha)
    for k in range(1, 6):
        setattr(a, "e%d" % k, "../trained_models/wct_se_16x_new/%dSE.pth" % k)
        setattr(a, "d%d" % k, "../trained_models/wct_se_16x_new_sd/%dSD.pth" % k)
    return a


@torch.no_grad()
def ref_transform(wct, cF, sF, alpha):
    """util_wct.py:210-223 called the way WCT.py:102-104 does (CPU f32 CHW in), with csF
    pre-sized because `csF.data.resize_` no longer resizes the caller's tensor."""
    csF = torch.empty(1, *cF.shape)
    out = wct.transform(cF, sF, csF, alpha)
    assert out.shape == (1,) + tuple(cF.shape)
    return out


@torch.no_grad()
def ref_style_transfer(wct, enc, dec, cImg, sImg, alpha, trace=None):
    sF = enc(sImg).squeeze(0)
    cF = enc(cImg).squeeze(0)
    csF = ref_transform(wct, cF, sF, alpha)
    img = dec(csF)
    if trace is not None:
        trace.append((cF.numpy().copy(), sF.numpy().copy(), csF.squeeze(0).numpy().copy(), img.squeeze(0).numpy().copy()))
    return img


def load_rgb(path, crop=None, origin=None):
    from PIL import Image
    im = np.asarray(Image.open(path).convert("RGB"), dtype=np.float32) / 255.0  # ToTensor(): /255, CHW
    im = np.ascontiguousarray(im.transpose(2, 0, 1))
    if crop is not None:
        h, w = crop
        H, W = im.shape[1:]
        y0, x0 = ((H - h) // 2, (W - w) // 2) if origin is None else origin
        im = np.ascontiguousarray(im[:, y0:y0 + h, x0:x0 + w])
    return im


def t(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct = util_wct.WCT(ref_args())
    wct.eval()
    os.makedirs(GOLD, exist_ok=True)

    # ------------------------------------------------------------------ weights
    blob = {}
    for k in range(1, 6):
        for kind, mod in (("enc", getattr(wct, "e%d" % k)), ("dec", getattr(wct, "d%d" % k))):
            key = model_zoo.module_key(kind, k)
            sd = mod.state_dict()
            layers = model_zoo.encoder_layers("16x", k) if kind == "enc" else model_zoo.decoder_layers("16x", k)
            names = [l.name for l in layers] + (["conv0"] if kind == "enc" else [])
            for n in names:
                blob["%s.%s.weight" % (key, n)] = sd[n + ".weight"].numpy().astype(np.float32)
                blob["%s.%s.bias" % (key, n)] = sd[n + ".bias"].numpy().astype(np.float32)
            for l in layers:  # the graph description must agree with the checkpoints
                assert tuple(sd[l.name + ".weight"].shape) == (l.cout, l.cin, 3, 3), (key, l)
            unused = [n for n in sd if n.split(".")[0] not in names]
            assert all("aux" in n for n in unused), unused
    wpath = os.path.join(REPO, "collaborative-distillation_amd", "weights", "16x.npz")
    np.savez(wpath, **blob)
    nparam = sum(v.size for v in blob.values())
    print("weights: %d tensors, %d params -> %s" % (len(blob), nparam, wpath))

    rng = np.random.default_rng(1234)

    # ------------------------------------------------------------------ G1 per-op
    g1 = {}
    seen = set()
    pad = torch.nn.ReflectionPad2d((1, 1, 1, 1))
    for k in range(5, 0, -1):
        for kind in ("enc", "dec"):
            mod = getattr(wct, ("e%d" if kind == "enc" else "d%d") % k)
            layers = model_zoo.encoder_layers("16x", k) if kind == "enc" else model_zoo.decoder_layers("16x", k)
            for l in layers:
                if (l.cin, l.cout) in seen:
                    continue
                seen.add((l.cin, l.cout))
                x = rng.random((1, l.cin, 13, 11), dtype=np.float32) * 2 - 0.5
                conv = getattr(mod, l.name)
                with torch.no_grad():
                    y = torch.relu(conv(pad(t(x)))).numpy()
                tag = "conv_%s_%s" % (model_zoo.module_key(kind, k), l.name)
                g1[tag + ".x"] = x
                g1[tag + ".y"] = y
    x = rng.random((1, 5, 13, 11), dtype=np.float32)
    g1["maxpool.x"] = x
    g1["maxpool.y"] = torch.nn.MaxPool2d(2, 2)(t(x)).numpy()
    g1["upsample.x"] = x
    g1["upsample.y"] = torch.nn.UpsamplingNearest2d(scale_factor=2)(t(x)).numpy()
    x = rng.random((1, 3, 9, 7), dtype=np.float32)
    with torch.no_grad():
        g1["conv0_e5.x"] = x
        g1["conv0_e5.y"] = wct.e5.conv0(t(x)).numpy()
    np.savez_compressed(os.path.join(GOLD, "g1_ops.npz"), **g1)
    print("G1: %d arrays, distinct convs: %s" % (len(g1), sorted(seen)))

    # ------------------------------------------------------------------ G2 per-module
    g2 = {}
    x = rng.random((1, 3, 48, 80), dtype=np.float32)
    g2["img"] = x
    for k in range(1, 6):
        with torch.no_grad():
            f = getattr(wct, "e%d" % k)(t(x))
            # decoder input: a perturbed feature so it is not tied to the encoder's own output
            fin = (f * (0.5 + t(rng.random(tuple(f.shape), dtype=np.float32)))).contiguous()
            y = getattr(wct, "d%d" % k)(fin)
        g2["e%d.y" % k] = f.numpy()
        g2["d%d.x" % k] = fin.numpy()
        g2["d%d.y" % k] = y.numpy()
    # odd sizes: floor pooling drops rows/cols (135x33 -> L5 128x32 ... SURVEY appendix A)
    xo = rng.random((1, 3, 45, 37), dtype=np.float32)
    g2["img_odd"] = xo
    for k in range(1, 6):
        with torch.no_grad():
            g2["e%d.y_odd" % k] = getattr(wct, "e%d" % k)(t(xo)).numpy()
    np.savez_compressed(os.path.join(GOLD, "g2_modules.npz"), **g2)
    print("G2: %d arrays" % len(g2))

    # ------------------------------------------------------------------ G3 transform
    g3 = {}

    def add_case(name, cF, sF, alpha):
        out = ref_transform(wct, t(cF), t(sF), alpha).numpy()
        g3[name + ".cF"] = cF
        g3[name + ".sF"] = sF
        g3[name + ".alpha"] = np.float64(alpha)
        g3[name + ".out"] = out
        assert np.isfinite(out).all(), name

    # full rank, C=24
    cF = np.maximum(rng.standard_normal((24, 20, 30)).astype(np.float32) + 0.5, 0)
    sF = np.maximum(rng.standard_normal((24, 17, 23)).astype(np.float32) * 2 + 0.3, 0)
    add_case("fullrank24", cF, sF, 1.0)
    add_case("fullrank24_a06", cF, sF, 0.6)
    # ReLU-sparse with exactly-dead and nearly-dead channels, C=32
    cF = np.maximum(rng.standard_normal((32, 16, 24)).astype(np.float32), 0)
    sF = np.maximum(rng.standard_normal((32, 18, 20)).astype(np.float32), 0)
    cF[3] = 0; cF[17] = 0; sF[3] = 0; sF[9] = 0
    cF[5] = 0; cF[5, 7, 11] = 0.75       # one active pixel
    sF[21] = 0; sF[21, 2, 3] = 1.25
    add_case("dead32", cF, sF, 1.0)
    # hw < C on the content side, C=128
    cF = np.maximum(rng.standard_normal((128, 5, 10)).astype(np.float32) + 0.2, 0)
    sF = np.maximum(rng.standard_normal((128, 24, 30)).astype(np.float32) + 0.2, 0)
    add_case("hw_lt_C_content", cF, sF, 1.0)
    # hw < C on the style side
    cF = np.maximum(rng.standard_normal((128, 20, 26)).astype(np.float32) + 0.2, 0)
    sF = np.maximum(rng.standard_normal((128, 6, 9)).astype(np.float32) + 0.2, 0)
    add_case("hw_lt_C_style", cF, sF, 1.0)
    # correlated channels (ill-conditioned but live), C=64
    base = rng.standard_normal((8, 22, 22)).astype(np.float32)
    mix = rng.standard_normal((64, 8)).astype(np.float32)
    cF = np.maximum(np.einsum("ck,khw->chw", mix, base) + 0.05 * rng.standard_normal((64, 22, 22)).astype(np.float32), 0).astype(np.float32)
    sF = np.maximum(rng.standard_normal((64, 19, 21)).astype(np.float32) + 0.1, 0)
    add_case("illcond64", cF, sF, 1.0)
    np.savez_compressed(os.path.join(GOLD, "g3_transform.npz"), **g3)
    print("G3: %d arrays" % len(g3))

    # ------------------------------------------------------------------ G4/G5 cascade on real crops
    g4 = {}
    content = load_rgb(os.path.join(REF, "PytorchWCT/content/in4.jpg"), crop=(128, 128))
    style = load_rgb(os.path.join(REF, "PytorchWCT/style/in3.jpg"), crop=(128, 128))
    for tag, c, s in (("a", content, style),
                      ("b", load_rgb(os.path.join(REF, "PytorchWCT/content/in4.jpg"), crop=(120, 136), origin=(200, 150)),
                       load_rgb(os.path.join(REF, "PytorchWCT/style/in3.jpg"), crop=(96, 112), origin=(100, 300)))):
        trace = []
        img = t(c[None])
        for k in (5, 4, 3, 2, 1):
            img = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), img, t(s[None]), 1.0, trace)
        g4[tag + ".content"] = c
        g4[tag + ".style"] = s
        g4[tag + ".final"] = img.squeeze(0).numpy()
        for k, (cF, sF, csF, out) in zip((5, 4, 3, 2, 1), trace):
            C = cF.shape[0]
            X = cF.reshape(C, -1).astype(np.float64)
            g4["%s.L%d.c_mean" % (tag, k)] = X.mean(1)
            g4["%s.L%d.c_cov" % (tag, k)] = np.cov(X)
            g4["%s.L%d.dead" % (tag, k)] = np.int64((X.max(1) == 0).sum())
            g4["%s.L%d.out" % (tag, k)] = out.astype(np.float32)
            if k >= 4:
                g4["%s.L%d.csF" % (tag, k)] = csF.astype(np.float32)
        print("G4[%s]: final %s range [%.3f, %.3f], dead/level %s" % (
            tag, g4[tag + ".final"].shape, g4[tag + ".final"].min(), g4[tag + ".final"].max(),
            [int(g4["%s.L%d.dead" % (tag, k)]) for k in (5, 4, 3, 2, 1)]))
    np.savez_compressed(os.path.join(GOLD, "g4_cascade.npz"), **g4)

    # ------------------------------------------------------------------ G6 original architecture, generated weights
    from model.model_original import (Encoder1, Encoder2, Encoder3, Encoder4, Encoder5,
                                      Decoder1, Decoder2, Decoder3, Decoder4, Decoder5)
    encs = [Encoder1, Encoder2, Encoder3, Encoder4, Encoder5]
    decs = [Decoder1, Decoder2, Decoder3, Decoder4, Decoder5]
    s = rng.random((1, 3, 48, 80), dtype=np.float32)
    contents = {k: rng.random((1, 3, 64, 64), dtype=np.float32) for k in (5, 4, 3, 2, 1)}
    owct = types.SimpleNamespace(transform=wct.transform)
    # level-isolated (a fresh noise content per level): with random decoders a chained cascade collapses
    # to a constant image after two levels, which would only test the degenerate cov = 0 branch.  A random
    # decoder can also die outright (all-negative pre-activations -> image of zeros): take the first seed
    # for which every level's output is alive.
    for seed in range(2024, 2100):
        ow = model_zoo.synth_weights("original", seed=seed)
        mods = {}
        for k in range(1, 6):
            for kind, cls in (("enc", encs[k - 1]), ("dec", decs[k - 1])):
                m = cls(None)
                key = model_zoo.module_key(kind, k)
                sd = {n[len(key) + 1:]: t(v) for n, v in ow.items() if n.startswith(key + ".")}
                m.load_state_dict(sd, strict=True)
                m.eval()
                mods[key] = m
        g6 = {"seed": np.int64(seed), "style": s[0]}
        alive = True
        for k in (5, 4, 3, 2, 1):
            c = contents[k]
            g6["L%d.content" % k] = c[0]
            with torch.no_grad():
                g6["e%d.y" % k] = mods["e%d" % k](t(c)).numpy()
            img = ref_style_transfer(owct, mods["e%d" % k], mods["d%d" % k], t(c), t(s), 1.0)
            g6["L%d.out" % k] = img.squeeze(0).numpy()
            o = g6["L%d.out" % k]
            alive = alive and all(o[ch].std() > 0.05 for ch in range(3))
        if alive:
            break
    for k in (5, 4, 3, 2, 1):
        o = g6["L%d.out" % k]
        print("G6 seed %d L%d: out range [%.3f, %.3f] std %.3f" % (seed, k, o.min(), o.max(), o.std()))
    np.savez_compressed(os.path.join(GOLD, "g6_original.npz"), **g6)

    # ------------------------------------------------------------------ G8 degenerate: constant content image
    # every feature map is exactly constant; with hw a power of two the reference's centred features are
    # exactly 0, cov = 0, k_c = 0 (util_wct.py:82-86) and the target is the style mean everywhere.
    g8 = {}
    c = np.full((1, 3, 64, 64), 0.3, np.float32)
    s8 = load_rgb(os.path.join(REF, "PytorchWCT/style/in3.jpg"), crop=(64, 96))
    g8["content"], g8["style"] = c[0], s8
    for k in (3, 1):
        g8["L%d.out" % k] = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), t(c), t(s8[None]), 1.0).squeeze(0).numpy()
        assert np.isfinite(g8["L%d.out" % k]).all()
    np.savez_compressed(os.path.join(GOLD, "g8_constant.npz"), **g8)

    # ------------------------------------------------------------------ G7 config 1 (512x512, relu1_1 only)
    r0 = np.random.default_rng(0)
    c = r0.random((1, 3, 512, 512), dtype=np.float32)
    s = r0.random((1, 3, 512, 512), dtype=np.float32)
    out = ref_style_transfer(wct, wct.e1, wct.d1, t(c), t(s), 1.0).squeeze(0).numpy()
    g7 = {"crop": out[:, 200:264, 300:364].copy(), "mean": np.float64(out.mean(dtype=np.float64)),
          "std": np.float64(out.std(dtype=np.float64)), "max": np.float64(out.max()),
          "chan_mean": out.reshape(3, -1).mean(1, dtype=np.float64)}
    np.savez_compressed(os.path.join(GOLD, "g7_config1.npz"), **g7)
    print("G7: mean %.6f std %.6f max %.6f" % (g7["mean"], g7["std"], g7["max"]))

    for f in sorted(os.listdir(GOLD)):
        print("%-24s %8d bytes" % (f, os.path.getsize(os.path.join(GOLD, f))))


def gen_g9():
    """G9 image edge.  ToTensor / save_image live in torchvision (requirements.txt pins torchvision==0.2.1, absent from
    the snapshot and from this image); their published 0.2.1 bodies, at the reference's call sites
    (PytorchWCT/data_loader.py:57-58, PytorchWCT/WCT.py:128), are restated here with the torch ops they execute:
      ToTensor (PIL RGB uint8):  torch.ByteTensor(HWC).permute(2,0,1).float().div(255)
      save_image (single image): tensor.mul(255).clamp(0, 255).byte().permute(1, 2, 0)     [>= 0.4: .add_(0.5) first]"""
    r = np.random.default_rng(9)
    u8 = r.integers(0, 256, size=(37, 53, 3), dtype=np.uint8)
    u8[0, :8, 0] = [0, 1, 2, 127, 128, 254, 255, 3]
    tt = torch.from_numpy(u8).permute(2, 0, 1).float().div(255).contiguous().numpy()
    f = (r.random((3, 29, 41), dtype=np.float32) * 1.5 - 0.2).astype(np.float32)     # below 0 and above 1 (unclamped ReLU output)
    k = np.arange(0, 29 * 41 * 3, dtype=np.int64) % 256
    edge = (k.astype(np.float32) / np.float32(255)).reshape(3, 29, 41)               # exactly representable k/255 boundaries
    f[:, ::3, ::2] = edge[:, ::3, ::2]
    f[0, 1, 1], f[1, 1, 1], f[2, 1, 1] = np.float32(1.0), np.float32(0.99999994), np.float32(254.5 / 255)
    ft = torch.from_numpy(f)
    sv0 = ft.mul(255).clamp(0, 255).byte().permute(1, 2, 0).contiguous().numpy()
    sv1 = ft.mul(255).add(0.5).clamp(0, 255).byte().permute(1, 2, 0).contiguous().numpy()
    np.savez_compressed(os.path.join(GOLD, "g9_image_edge.npz"), u8=u8, to_tensor=tt, f32=f, save_trunc=sv0, save_round=sv1)
    print("G9: ToTensor %s -> %s, save_image %s -> %s" % (u8.shape, tt.shape, f.shape, sv0.shape))


def gen_g10():
    """G10 `--numpy`: the reference's whiten_and_color_np path (util_wct.py:134-208, `+ I` on the content covariance)
    through wct.transform, on relu3_1-like (C = 64, with exactly-dead channels) and relu1_1-like (C = 24) features."""
    util_wct = import_reference()
    a = ref_args()
    a.numpy = True
    wct = util_wct.WCT(a)
    wct.eval()
    r = np.random.default_rng(10)
    g = {}
    for tag, C, hw, hws, alpha in (("c64", 64, (23, 31), (19, 17), 1.0), ("c24", 24, (40, 36), (28, 44), 0.6)):
        cF = np.maximum(r.normal(0.3, 1.0, size=(C,) + hw), 0).astype(np.float32)
        sF = np.maximum(r.normal(0.1, 1.5, size=(C,) + hws), 0).astype(np.float32)
        if C == 64:
            cF[5] = 0; cF[40] = 0; sF[7] = 0     # exactly dead channels: harmless with + I on the content side
        out = ref_transform(wct, t(cF), t(sF), alpha).numpy()
        g[tag + ".cF"], g[tag + ".sF"], g[tag + ".alpha"], g[tag + ".csF"] = cF, sF, np.float64(alpha), out
        print("G10 %s: out mean %.5f" % (tag, out.mean()))
    np.savez_compressed(os.path.join(GOLD, "g10_numpy_variant.npz"), **g)


def gen_g11():
    """G11: the reference's own UHD sample pair at BASELINE config-2 size -- content/UHD_content/green_park-wallpaper-
    3840x2160.jpg (README.md:41-44 `--UHD`) with style/in1.jpg (2048x2048) standing in for the absent UHD styles
    (.MISSING_LARGE_BLOBS).  The two JPEG files are DATA of the reference and are copied byte for byte as fixtures; the
    expected output is the reference itself (util_wct.WCT with the real 16x checkpoints, torch CPU, WCT.py:120-125 restated),
    stored as: a 16x box-downsampled version of the final image, four 96x96 crops, global mean / std / max.  ~5 min here."""
    import shutil
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct = util_wct.WCT(ref_args("16x", 1.0))
    cpath = os.path.join(REF, "PytorchWCT/content/UHD_content/green_park-wallpaper-3840x2160.jpg")
    spath = os.path.join(REF, "PytorchWCT/style/in1.jpg")
    shutil.copyfile(cpath, os.path.join(GOLD, "g11_uhd_content_3840x2160.jpg"))
    shutil.copyfile(spath, os.path.join(GOLD, "g11_style_2048x2048.jpg"))
    c, s = load_rgb(cpath), load_rgb(spath)
    assert c.shape == (3, 2160, 3840) and s.shape == (3, 2048, 2048)
    t0 = time.time()
    img = t(c[None])
    for k in (5, 4, 3, 2, 1):
        img = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), img, t(s[None]), 1.0)
        print("G11: level %d done, %.0f s" % (k, time.time() - t0), flush=True)
    y = img.squeeze(0).numpy()
    assert y.shape == (3, 2160, 3840)
    g = {"shape": np.array(y.shape), "mean": np.float64(y.mean(dtype=np.float64)), "std": np.float64(y.std(dtype=np.float64)),
         "max": np.float32(y.max()), "min": np.float32(y.min()),
         "down16": y.reshape(3, 135, 16, 240, 16).mean(axis=(2, 4), dtype=np.float64).astype(np.float32)}
    for i, (y0, x0) in enumerate(((0, 0), (1032, 1872), (2064, 3744), (500, 3000))):
        g["crop%d.origin" % i] = np.array([y0, x0])
        g["crop%d" % i] = y[:, y0:y0 + 96, x0:x0 + 96].copy()
    np.savez_compressed(os.path.join(GOLD, "g11_uhd_pair.npz"), **g)
    print("G11: mean %.6f std %.6f max %.4f" % (g["mean"], g["std"], g["max"]))


def gen_g12():
    """G12 Resize.  PytorchWCT/data_loader.py:52-56 calls transforms.Resize(size) (torchvision==0.2.1, absent from the snapshot and
    this image); its published body for a PIL image and an int size is restated here with the Pillow call it makes:
        (w, h) = img.size; unchanged if the smaller edge == size; else smaller edge -> size, other -> int(size * long / short);
        img.resize((ow, oh), Image.BILINEAR)
    The pixels come from Pillow itself (this image: the version printed below; requirements.txt pins 8.2.0 -- the 8-bit resampler's
    arithmetic is the same).  Inputs: a crop of the reference's content/in4.jpg (natural data) and seeded noise, portrait and
    landscape; shrinking (antialiased), enlarging, one edge unchanged, no-op."""
    import PIL
    from PIL import Image

    def tv_resize(img, size):
        w, h = img.size
        if (w <= h and w == size) or (h <= w and h == size):
            return img
        if w < h:
            ow, oh = size, int(size * h / w)
        else:
            oh, ow = size, int(size * w / h)
        return img.resize((ow, oh), Image.BILINEAR)

    r = np.random.default_rng(12)
    nat = np.asarray(Image.open(os.path.join(REF, "PytorchWCT", "content", "in4.jpg")).convert("RGB"))[100:260, 80:320].copy()   # 160 x 240
    noise_p = r.integers(0, 256, size=(131, 97, 3), dtype=np.uint8)       # portrait
    noise_l = r.integers(0, 256, size=(45, 200, 3), dtype=np.uint8)       # landscape, strongly non-square
    g = {"nat": nat, "noise_p": noise_p, "noise_l": noise_l, "pillow": np.array(PIL.__version__)}
    for name, img in (("nat", nat), ("noise_p", noise_p), ("noise_l", noise_l)):
        for size in (64, 77, 30, 7, 300 if name == "nat" else 150, min(img.shape[:2])):
            g["%s.resize%d" % (name, size)] = np.asarray(tv_resize(Image.fromarray(img), size))
    # explicit (oh, ow) targets: one edge unchanged, single row / column outputs
    for (oh, ow) in ((160, 100), (50, 240), (1, 1), (333, 17)):
        g["nat.to%dx%d" % (oh, ow)] = np.asarray(Image.fromarray(nat).resize((ow, oh), Image.BILINEAR))
    np.savez_compressed(os.path.join(GOLD, "g12_resize.npz"), **g)
    print("G12: Pillow %s, %d arrays" % (PIL.__version__, len(g)))


def smooth_frame(rng, shape, it=3):
    """The `smooth` synthetic variant of SURVEY 8(d) (box-blurred noise, wrap-around, normalised to [0, 1]); the same function as
    tests/test_hip_scale.py::smooth and bench.py::smooth_frame."""
    x = rng.random(shape, dtype=np.float32)
    for _ in range(it):
        x = (x + np.roll(x, 1, 1) + np.roll(x, 1, 2) + np.roll(x, -1, 1) + np.roll(x, -1, 2)) / 5
    return np.ascontiguousarray((x - x.min()) / (x.max() - x.min()))


G13_CROPS = ((0, 0), (1032, 1872), (2064, 3744), (500, 3000), (0, 3744), (2064, 0), (1500, 700), (300, 1900))


def pack_frame_fixture(y, crops, lattice=(1, 2, 4)):
    """What is kept of a reference output too large to commit whole: global statistics, a 16x box-downsampled image, 96x96
    crops, and a regular lattice of pixels y[:, oy::st, ox::st] (1/16 of the image: error quantiles against the reference)."""
    C, H, W = y.shape
    oy, ox, st = lattice
    g = {"shape": np.array(y.shape), "mean": np.float64(y.mean(dtype=np.float64)), "std": np.float64(y.std(dtype=np.float64)),
         "max": np.float32(y.max()), "min": np.float32(y.min()),
         "down16": y[:, :H // 16 * 16, :W // 16 * 16].reshape(C, H // 16, 16, W // 16, 16).mean(axis=(2, 4), dtype=np.float64).astype(np.float32),
         "lattice.origin_stride": np.array(lattice), "lattice": np.ascontiguousarray(y[:, oy::st, ox::st])}
    for i, (y0, x0) in enumerate(crops):
        g["crop%d.origin" % i] = np.array([y0, x0])
        g["crop%d" % i] = y[:, y0:y0 + 96, x0:x0 + 96].copy()
    return g


def gen_g13(which=("noise", "smooth")):
    """G13: the SYNTHETIC config-2 frames bench.py times and tests/test_hip_scale.py checks, through the reference itself
    (util_wct.WCT, real 16x checkpoints, torch CPU, WCT.py:120-125 restated) -- SURVEY 8(d) seeds: content
    numpy.random.default_rng(1).random((3, 2160, 3840), float32), style default_rng(2).random((3, 2048, 2048), float32); the
    `smooth` variant's content is smooth_frame(default_rng(101), ...).  Inputs are regenerated from the seeds (numpy's PCG64
    stream is stable across versions), the expected outputs are stored by pack_frame_fixture.  ~5 min per frame here."""
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct = util_wct.WCT(ref_args("16x", 1.0))
    s = np.random.default_rng(2).random((3, 2048, 2048), dtype=np.float32)
    for kind in which:
        c = (np.random.default_rng(1).random((3, 2160, 3840), dtype=np.float32) if kind == "noise"
             else smooth_frame(np.random.default_rng(101), (3, 2160, 3840)))
        t0 = time.time()
        img = t(c[None])
        for k in (5, 4, 3, 2, 1):
            img = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), img, t(s[None]), 1.0)
            print("G13 %s: level %d done, %.0f s" % (kind, k, time.time() - t0), flush=True)
        y = img.squeeze(0).numpy()
        assert y.shape == (3, 2160, 3840) and np.isfinite(y).all()
        g = pack_frame_fixture(y, G13_CROPS)
        g["content.checksum"] = np.float64(c.sum(dtype=np.float64))      # guards the seed -> input reproduction
        g["style.checksum"] = np.float64(s.sum(dtype=np.float64))
        g["torch"] = np.array(torch.__version__)
        np.savez_compressed(os.path.join(GOLD, "g13_cfg2_%s.npz" % kind), **g)
        print("G13 %s: mean %.6f std %.6f max %.4f  (%.0f s)" % (kind, g["mean"], g["std"], g["max"], time.time() - t0), flush=True)


def gen_g14():
    """G14: config 3 -- `--mode original` at 1920x1080 -- through the reference's own classes (model_original.py Encoder{k} /
    Decoder{k}, util_wct.WCT.transform) with the GENERATED weights model_zoo.synth_weights("original", 3) (the torch7
    checkpoints are absent: real-weight parity stays unpinned), content default_rng(3), style default_rng(4), both
    (3, 1080, 1920) -- the frame bench.py's cfg3 pass and tests/test_hip_scale.py use.  Output 3 x 1072 x 1920."""
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct16 = util_wct.WCT(ref_args("16x", 1.0))          # only for its .transform (util_wct.py:210-223; mode-independent)
    from model.model_original import (Encoder1, Encoder2, Encoder3, Encoder4, Encoder5,
                                      Decoder1, Decoder2, Decoder3, Decoder4, Decoder5)
    encs = [Encoder1, Encoder2, Encoder3, Encoder4, Encoder5]
    decs = [Decoder1, Decoder2, Decoder3, Decoder4, Decoder5]
    ow = model_zoo.synth_weights("original", 3)
    mods = {}
    for k in range(1, 6):
        for kind, cls in (("enc", encs[k - 1]), ("dec", decs[k - 1])):
            m = cls(None)
            key = model_zoo.module_key(kind, k)
            m.load_state_dict({n[len(key) + 1:]: t(v) for n, v in ow.items() if n.startswith(key + ".")}, strict=True)
            m.eval()
            mods[key] = m
    owct = types.SimpleNamespace(transform=wct16.transform)
    c = np.random.default_rng(3).random((3, 1080, 1920), dtype=np.float32)
    s = np.random.default_rng(4).random((3, 1080, 1920), dtype=np.float32)
    t0 = time.time()
    img = t(c[None])
    g = {}
    for k in (5, 4, 3, 2, 1):
        img = ref_style_transfer(owct, mods["e%d" % k], mods["d%d" % k], img, t(s[None]), 1.0)
        o = img.squeeze(0).numpy()
        g["L%d.mean" % k], g["L%d.max" % k] = np.float64(o.mean(dtype=np.float64)), np.float32(o.max())
        print("G14: level %d done, %.0f s, out %s max %.3f" % (k, time.time() - t0, tuple(o.shape), o.max()), flush=True)
    y = img.squeeze(0).numpy()
    assert y.shape == (3, 1072, 1920) and np.isfinite(y).all()
    g.update(pack_frame_fixture(y, ((0, 0), (488, 912), (976, 1824), (300, 1500)), lattice=(1, 2, 4)))
    g["weights"] = np.array("model_zoo.synth_weights('original', 3)")
    g["torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(GOLD, "g14_cfg3_original.npz"), **g)
    print("G14: mean %.6f std %.6f max %.4f" % (g["mean"], g["std"], g["max"]), flush=True)


def gen_g15(which=("noise", "natural")):
    """G15 (round 4): `--mode original` at config-3 size on a WELL-CONDITIONED generated set -- model_zoo.synth_weights_conditioned
    ("original", 15): paired-isometry layers, see its docstring -- through the reference's own classes (model_original.py
    Encoder{k} / Decoder{k}) and util_wct.WCT.transform.  On G14's He-uniform stacks two valid fp32 implementations of the reference
    end 2.4e-3 apart (chaos of the reference's own arithmetic: nearly-dead channels under five whitenings), so G14 can only gate
    relative to the oracle; here they end < 1e-4 apart and the un-pruned graph is gated end to end at the LITERAL 1e-3.
    Two frames, both (3, 1080, 1920) -> 3 x 1072 x 1920: `noise` = config 3's seeds (content default_rng(3), style default_rng(4));
    `natural` = the reference's UHD sample pair (the committed G11 JPEGs) resized to 1920x1080 with Pillow's bilinear filter
    (tests/fixture_compare.py::cfg3_natural_frames regenerates exactly these arrays; checksums stored)."""
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    sys.path.insert(0, REPO)
    from tests.fixture_compare import cfg3_frames, cfg3_natural_frames
    wct16 = util_wct.WCT(ref_args("16x", 1.0))          # only for its .transform (util_wct.py:210-223; mode-independent)
    from model.model_original import (Encoder1, Encoder2, Encoder3, Encoder4, Encoder5,
                                      Decoder1, Decoder2, Decoder3, Decoder4, Decoder5)
    encs = [Encoder1, Encoder2, Encoder3, Encoder4, Encoder5]
    decs = [Decoder1, Decoder2, Decoder3, Decoder4, Decoder5]
    ow = model_zoo.synth_weights_conditioned("original", 15)
    mods = {}
    for k in range(1, 6):
        for kind, cls in (("enc", encs[k - 1]), ("dec", decs[k - 1])):
            m = cls(None)
            key = model_zoo.module_key(kind, k)
            m.load_state_dict({n[len(key) + 1:]: t(v) for n, v in ow.items() if n.startswith(key + ".")}, strict=True)
            m.eval()
            mods[key] = m
    owct = types.SimpleNamespace(transform=wct16.transform)
    for kind in which:
        c, s = cfg3_frames() if kind == "noise" else cfg3_natural_frames(GOLD)
        t0 = time.time()
        img = t(c[None])
        g = {}
        for k in (5, 4, 3, 2, 1):
            img = ref_style_transfer(owct, mods["e%d" % k], mods["d%d" % k], img, t(s[None]), 1.0)
            o = img.squeeze(0).numpy()
            g["L%d.mean" % k], g["L%d.max" % k] = np.float64(o.mean(dtype=np.float64)), np.float32(o.max())
            print("G15 %s: level %d done, %.0f s, out %s max %.3f" % (kind, k, time.time() - t0, tuple(o.shape), o.max()), flush=True)
        y = img.squeeze(0).numpy()
        assert y.shape == (3, 1072, 1920) and np.isfinite(y).all()
        g.update(pack_frame_fixture(y, ((0, 0), (488, 912), (976, 1824), (300, 1500)), lattice=(1, 2, 4)))
        g["content.checksum"] = np.float64(c.sum(dtype=np.float64))
        g["style.checksum"] = np.float64(s.sum(dtype=np.float64))
        g["weights"] = np.array("model_zoo.synth_weights_conditioned('original', 15)")
        g["weights.checksum"] = np.float64(sum(float(np.abs(v).sum(dtype=np.float64)) for v in ow.values()))
        g["torch"] = np.array(torch.__version__)
        np.savez_compressed(os.path.join(GOLD, "g15_cfg3_conditioned_%s.npz" % kind), **g)
        print("G15 %s: mean %.6f std %.6f max %.4f (%.0f s)" % (kind, g["mean"], g["std"], g["max"], time.time() - t0), flush=True)


G16_CROPS = ((0, 0), (416, 10144), (200, 1232), (100, 2512), (416, 5072), (0, 6352), (300, 7632), (150, 8912))


def gen_g16():
    """G16 (round 5): config 4's GEOMETRY through the reference itself -- a 10240-wide x 512-tall content (BASELINE configs[3]'s width:
    eight 1280-column strips, every level's halo 160/72/24/10/2 inside one strip; a fifth of a megapixel-row budget the reference
    finishes in ~2 min here), numpy.random.default_rng(5).random((3, 512, 10240), float32), + the config-2/4 style (default_rng(2),
    2048x2048), util_wct.WCT with the real 16x checkpoints on torch CPU, WCT.py:120-125 restated.  Six of the eight crops straddle
    strip boundaries (x = 1280 k).  Gated in tests/test_sharded_gpu.py: the untiled frame AND the 8 x 1280 exchange-halo job against
    these pixels."""
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct = util_wct.WCT(ref_args("16x", 1.0))
    s = np.random.default_rng(2).random((3, 2048, 2048), dtype=np.float32)
    c = np.random.default_rng(5).random((3, 512, 10240), dtype=np.float32)
    t0 = time.time()
    img = t(c[None])
    for k in (5, 4, 3, 2, 1):
        img = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), img, t(s[None]), 1.0)
        print("G16: level %d done, %.0f s" % (k, time.time() - t0), flush=True)
    y = img.squeeze(0).numpy()
    assert y.shape == (3, 512, 10240) and np.isfinite(y).all()
    g = pack_frame_fixture(y, G16_CROPS)
    g["content.checksum"] = np.float64(c.sum(dtype=np.float64))
    g["style.checksum"] = np.float64(s.sum(dtype=np.float64))
    g["torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(GOLD, "g16_cfg4_geometry.npz"), **g)
    print("G16: mean %.6f std %.6f max %.4f  (%.0f s)" % (g["mean"], g["std"], g["max"], time.time() - t0), flush=True)


def cfg4_natural_content(gold_dir=GOLD):
    """G17's content: config 4's WIDTH from a natural image -- rows 824..1335 (512 rows around the horizon) of the reference's UHD sample
    (the committed G11 JPEG, 3840 wide), repeated side by side and cropped to 10240 columns.  Same function as tests/fixture_compare.py."""
    c = load_rgb(os.path.join(gold_dir, "g11_uhd_content_3840x2160.jpg"))[:, 824:1336, :]
    return np.ascontiguousarray(np.concatenate([c, c, c], axis=2)[:, :, :10240])


def gen_g17():
    """G17 (round 6, VERDICT r5 task 7): config 4's geometry on a NATURAL image -- cfg4_natural_content() (10240 x 512) + the reference's
    style/in1.jpg (2048 x 2048, the committed G11 JPEG) through util_wct.WCT, real 16x checkpoints, torch CPU, WCT.py:120-125 restated.
    Uniform noise (G16) leaves the reference's own arithmetic ~1e-3 of room; natural images an order of magnitude more, so this frame
    holds the 8-strip job to a bound with real headroom.  Same crops as G16 (six across strip boundaries)."""
    import time
    torch.set_num_threads(8)
    util_wct = import_reference()
    wct = util_wct.WCT(ref_args("16x", 1.0))
    s = load_rgb(os.path.join(GOLD, "g11_style_2048x2048.jpg"))
    c = cfg4_natural_content()
    assert c.shape == (3, 512, 10240) and s.shape == (3, 2048, 2048)
    t0 = time.time()
    img = t(c[None])
    for k in (5, 4, 3, 2, 1):
        img = ref_style_transfer(wct, getattr(wct, "e%d" % k), getattr(wct, "d%d" % k), img, t(s[None]), 1.0)
        print("G17: level %d done, %.0f s" % (k, time.time() - t0), flush=True)
    y = img.squeeze(0).numpy()
    assert y.shape == (3, 512, 10240) and np.isfinite(y).all()
    g = pack_frame_fixture(y, G16_CROPS)
    g["content.checksum"] = np.float64(c.sum(dtype=np.float64))
    g["style.checksum"] = np.float64(s.sum(dtype=np.float64))
    g["torch"] = np.array(torch.__version__)
    np.savez_compressed(os.path.join(GOLD, "g17_cfg4_geometry_natural.npz"), **g)
    print("G17: mean %.6f std %.6f max %.4f  (%.0f s)" % (g["mean"], g["std"], g["max"], time.time() - t0), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g12":
        os.makedirs(GOLD, exist_ok=True)
        gen_g12()
        gen_g13()
        gen_g14()
        gen_g15()
        gen_g16()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g17":
        os.makedirs(GOLD, exist_ok=True)
        gen_g17()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g13":
        os.makedirs(GOLD, exist_ok=True)
        gen_g13(tuple(sys.argv[3:]) or ("noise", "smooth"))
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g14":
        os.makedirs(GOLD, exist_ok=True)
        gen_g14()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g15":
        os.makedirs(GOLD, exist_ok=True)
        gen_g15(tuple(sys.argv[3:]) or ("noise", "natural"))
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g16":
        os.makedirs(GOLD, exist_ok=True)
        gen_g16()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g11":
        os.makedirs(GOLD, exist_ok=True)
        gen_g11()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g10":
        os.makedirs(GOLD, exist_ok=True)
        gen_g10()
    elif len(sys.argv) > 2 and sys.argv[1] == "--only" and sys.argv[2] == "g9":
        os.makedirs(GOLD, exist_ok=True)
        gen_g9()
    else:
        main()
        gen_g9()
        gen_g10()
        gen_g11()
        gen_g12()
        gen_g13()
        gen_g14()
        gen_g15()
        gen_g16()
