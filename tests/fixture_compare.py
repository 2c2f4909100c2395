"""Synthetic benchmark frames (SURVEY 8d seeds) and comparison against the reference-made frame fixtures G11 / G13 / G14.

Plain numpy, no pytest: shared by tests/ and by bench.py's parity leg.  The fixtures hold what tools/make_goldens.py kept of an
output of the REFERENCE ITSELF that is too large to commit whole (pack_frame_fixture): global statistics, a 16x box-downsampled
image, 96x96 crops and -- G13 / G14 -- a regular lattice y[:, oy::st, ox::st] (1/16 of the pixels).  All errors are
max|d| / max|reference|, the reference's maximum taken from the fixture.
"""
import numpy as np

GATE = 1e-3   # BASELINE.json north_star


def smooth_frame(rng, shape, it=3):
    """The `smooth` variant of SURVEY 8(d): box-blurred uniform noise (wrap-around), normalised to [0, 1] -- produces the dead
    channels and ill-conditioned covariances natural images have.  Same function as tools/make_goldens.py::smooth_frame."""
    x = rng.random(shape, dtype=np.float32)
    for _ in range(it):
        x = (x + np.roll(x, 1, 1) + np.roll(x, 1, 2) + np.roll(x, -1, 1) + np.roll(x, -1, 2)) / 5
    return np.ascontiguousarray((x - x.min()) / (x.max() - x.min()))


def noise_frame(seed, h, w):
    """Uniform noise in [0, 1) from numpy's PCG64 stream (stable across numpy versions): reproducible in the build container,
    where the reference runs, and on the GPU box."""
    return np.random.default_rng(seed).random((3, h, w), dtype=np.float32)


def cfg2_frames(kind="noise"):
    """BASELINE configs[1]: 3840x2160 content (seed 1; `smooth`: smooth_frame(seed 101)) + 2048x2048 style (seed 2)."""
    c = noise_frame(1, 2160, 3840) if kind == "noise" else smooth_frame(np.random.default_rng(101), (3, 2160, 3840))
    return c, noise_frame(2, 2048, 2048)


def cfg3_frames():
    """BASELINE configs[2]: 1920x1080 content (seed 3) and style (seed 4)."""
    return noise_frame(3, 1080, 1920), noise_frame(4, 1080, 1920)


def cfg3_natural_frames(gold_dir):
    """G15 `natural`: the reference's UHD sample pair (the committed G11 JPEGs: green_park 3840x2160, style/in1.jpg 2048x2048), each
    resized to 1920x1080 with Pillow's bilinear filter and converted like ToTensor (uint8 / 255)."""
    import os
    from PIL import Image

    def load(name):
        im = Image.open(os.path.join(gold_dir, name)).convert("RGB").resize((1920, 1080), Image.BILINEAR)
        return np.ascontiguousarray(np.asarray(im).transpose(2, 0, 1).astype(np.float32) / np.float32(255))

    return load("g11_uhd_content_3840x2160.jpg"), load("g11_style_2048x2048.jpg")


def cfg4_geometry_frames():
    """G16: BASELINE configs[3]'s WIDTH (10240 = eight 1280-column strips) at a height the reference finishes in a minute: 10240x512
    content (seed 5) + the 2048x2048 style (seed 2)."""
    return noise_frame(5, 512, 10240), noise_frame(2, 2048, 2048)


def cfg4_natural_frames(gold_dir):
    """G17: config 4's geometry on a NATURAL image -- rows 824..1335 of the reference's UHD sample (the committed G11 JPEG, 3840 wide)
    repeated side by side and cropped to 10240 x 512, + the reference's style/in1.jpg (2048 x 2048); ToTensor conversion (uint8 / 255).
    Same function as tools/make_goldens.py::cfg4_natural_content."""
    import os
    from PIL import Image

    def load(name):
        return np.ascontiguousarray(np.asarray(Image.open(os.path.join(gold_dir, name)).convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / np.float32(255))

    c = load("g11_uhd_content_3840x2160.jpg")[:, 824:1336, :]
    return np.ascontiguousarray(np.concatenate([c, c, c], axis=2)[:, :, :10240]), load("g11_style_2048x2048.jpg")


def compare_to_fixture(img, g):
    """img: 3 x H x W result; g: a frame fixture (dict of arrays).  -> dict of errors relative to the reference's maximum."""
    img = np.asarray(img)
    assert tuple(img.shape) == tuple(int(v) for v in g["shape"]), (img.shape, g["shape"])
    mx = float(g["max"])
    C, H, W = img.shape
    res = {}
    crops = []
    i = 0
    while "crop%d" % i in g:
        y0, x0 = (int(v) for v in g["crop%d.origin" % i])
        ref = g["crop%d" % i]
        crops.append(float(np.abs(img[:, y0:y0 + ref.shape[1], x0:x0 + ref.shape[2]].astype(np.float64) - ref).max() / mx))
        i += 1
    res["crops_max"] = max(crops)
    down = img[:, :H // 16 * 16, :W // 16 * 16].reshape(C, H // 16, 16, W // 16, 16).mean(axis=(2, 4), dtype=np.float64)
    res["down16_max"] = float(np.abs(down - g["down16"]).max() / mx)
    res["mean_diff"] = float(abs(img.mean(dtype=np.float64) - float(g["mean"])))
    worst = res["crops_max"]
    if "lattice" in g:
        oy, ox, st = (int(v) for v in g["lattice.origin_stride"])
        d = np.abs(img[:, oy::st, ox::st].astype(np.float64) - g["lattice"]) / mx
        res["lattice_max"] = float(d.max())
        res["lattice_p9999"] = float(np.quantile(d, 0.9999))
        res["lattice_frac_gt_gate"] = float((d > GATE).mean())
        res["lattice_pixels"] = int(d.size)
        worst = max(worst, res["lattice_max"])
    res["max"] = worst     # the figure the gate is applied to: every reference pixel the fixture holds
    return res
