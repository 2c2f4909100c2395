"""Layer graphs of the WCT encoder/decoder cascade (host-side description only).

The reference builds ten nn.Module classes per mode (model/model_cd.py:62-743 for
``16x``, model/model_original.py:11-599 for ``original``).  Every one of them is the
same VGG-19 prefix / mirrored suffix with different channel widths, so here the graph
is data: a list of ``Layer`` records that the HIP library (and the test oracle)
walk.  Nothing in this file computes anything.

Encoder level L  = conv0(1x1 affine, folded) + VGG-19 convs up to relu{L}_1
                   (model_cd.py:346-349, 403-409, 485-494, 589-603, 724-743)
Decoder level L  = mirror, nearest x2 upsample after conv51/41/31/21, ReLU after
                   EVERY conv including the last one (model_cd.py:83-85, 117-122,
                   159-167, 211-224, 276-294)
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Dict, List, Tuple

import numpy as np

MODES = ("original", "16x", "16x_kd2sd")


def arch(mode: str) -> str:
    """Layer graph of a mode: 16x_kd2sd (model/model_kd2sd.py) has the 16x graphs -- its aux 1x1 heads are training-only
    (never called in forward) and only the decoder checkpoints differ (WCT.py:60-70)."""
    return "16x" if mode == "16x_kd2sd" else mode

# channel width of VGG block k (conv{k}_x) per mode
_WIDTHS = {
    "16x": {1: 16, 2: 32, 3: 64, 4: 128, 5: 128},
    "original": {1: 64, 2: 128, 3: 256, 4: 512, 5: 512},
}
# the level-1 pruned encoder keeps 24 filters in conv1_1 (model_cd.py:324,
# SmallDecoder1_16x.conv11 is 24->3, model_cd.py:67)
_L1_WIDTH = {"16x": 24, "original": 64}

# VGG-19 conv order and which convs are followed by a 2x2 max-pool in the encoder
_VGG_ORDER = ["conv11", "conv12", "conv21", "conv22", "conv31", "conv32", "conv33",
              "conv34", "conv41", "conv42", "conv43", "conv44", "conv51"]
_POOL_AFTER = {"conv12", "conv22", "conv34", "conv44"}
_LAST_OF_LEVEL = {1: "conv11", 2: "conv21", 3: "conv31", 4: "conv41", 5: "conv51"}
# decoder: nearest-neighbour x2 after these convs
_UP_AFTER = {"conv51", "conv41", "conv31", "conv21"}


@dataclass(frozen=True)
class Layer:
    name: str
    cin: int
    cout: int
    pool_after: bool = False   # encoder: MaxPool2d(2,2) floor mode after ReLU
    up_after: bool = False     # decoder: UpsamplingNearest2d(2) after ReLU


def _block(name: str) -> int:
    return int(name[4])


def feature_channels(mode: str, level: int) -> int:
    """C of relu{level}_1 for the given mode (128,128,64,32,24 / 512,512,256,128,64)."""
    mode = arch(mode)
    if level == 1:
        return _L1_WIDTH[mode]
    return _WIDTHS[mode][level]


def encoder_layers(mode: str, level: int) -> List[Layer]:
    assert mode in MODES and 1 <= level <= 5
    mode = arch(mode)
    out: List[Layer] = []
    cin = 3
    last = _LAST_OF_LEVEL[level]
    for name in _VGG_ORDER:
        cout = _WIDTHS[mode][_block(name)]
        if level == 1:
            cout = _L1_WIDTH[mode]
        pool = name in _POOL_AFTER and name != last
        out.append(Layer(name, cin, cout, pool_after=pool))
        cin = cout
        if name == last:
            break
    return out


def decoder_layers(mode: str, level: int) -> List[Layer]:
    """Mirror of the encoder: conv XY maps enc.cout -> enc.cin, listed in execution order."""
    enc = encoder_layers(mode, level)
    out: List[Layer] = []
    for i, l in enumerate(reversed(enc)):
        last = i == len(enc) - 1
        out.append(Layer(l.name, l.cout, l.cin, up_after=(l.name in _UP_AFTER and not last)))
    return out


def output_size(level: int, H: int, W: int) -> Tuple[int, int, int, int]:
    """(h, w) of the relu{level}_1 feature and (Ho, Wo) of the decoded image.

    Floor-mode pooling drops odd rows/cols, so a 1080-row image leaves level 5 with
    67 feature rows and decodes to 1072 rows.
    """
    h, w = H, W
    for _ in range(level - 1):
        h, w = h // 2, w // 2
    return h, w, h << (level - 1), w << (level - 1)


# ----------------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------------
def module_key(kind: str, level: int) -> str:
    return "%s%d" % ("e" if kind == "enc" else "d", level)


def load_npz_weights(path: str) -> Dict[str, np.ndarray]:
    """Torch-free weight blob: keys ``e5.conv11.weight`` (OIHW f32), ``e5.conv11.bias``,
    ``e5.conv0.weight`` ([3,3,1,1]), ``e5.conv0.bias``, ``d5.conv51.weight`` ...
    Produced from the reference checkpoints by tools/make_goldens.py."""
    with np.load(path) as z:
        return {k: np.ascontiguousarray(z[k], dtype=np.float32) for k in z.files}


def t7_indices(kind: str, level: int) -> List[int]:
    """`model.get(i)` indices the reference reads a `--mode original` module's convolutions from, in forward order
    (model_original.py:27-28, 59, 92-95, 135-137, 179-184, 232-236, 288-297, 360-368, 471-484, 561-573).
    Encoders are conv0, then [pad, conv, relu] triples with a pool entry after conv12/22/34/44; decoders are
    [pad, conv, relu] triples with an unpool entry after conv51/41/31/21."""
    idx: List[int] = []
    if kind == "enc":
        idx.append(0)
        at = 1
        for l in encoder_layers("original", level):
            idx.append(at + 1)               # pad at `at`, conv at `at + 1`, relu at `at + 2`
            at += 3 + (1 if l.pool_after else 0)
    else:
        at = 0
        for l in decoder_layers("original", level):
            idx.append(at + 1)
            at += 3 + (1 if l.up_after else 0)
    return idx


def load_t7_module(path: str, kind: str, level: int) -> Dict[str, np.ndarray]:
    """One `--mode original` module from its torch7 file: {"conv0.weight", "conv0.bias", "conv11.weight", ...} like the
    state_dict of the same module (what `load_param_from_t7`, utils.py:64-67, copies into the nn.Module)."""
    from . import t7
    convs = t7.sequential_convs(t7.load(path))
    layers = encoder_layers("original", level) if kind == "enc" else decoder_layers("original", level)
    names = (["conv0"] if kind == "enc" else []) + [l.name for l in layers]
    shapes = ([(3, 3, 1, 1)] if kind == "enc" else []) + [(l.cout, l.cin, 3, 3) for l in layers]
    want = t7_indices(kind, level)
    if [c[0] for c in convs] != want:
        raise ValueError("%s: convolutions at Sequential indices %s, the reference reads %s (%s level %d)" % (path, [c[0] for c in convs], want, kind, level))
    out: Dict[str, np.ndarray] = {}
    for (i, w, b), name, shape in zip(convs, names, shapes):
        if w.shape != shape:
            raise ValueError("%s: module %d (%s) has weight %s, expected %s" % (path, i, name, w.shape, shape))
        out[name + ".weight"], out[name + ".bias"] = w, b
    return out


#: fixed conv0 of the un-pruned encoders: RGB[0,1] -> BGR*255 - mean  (model_original.py:428-433)
ORIGINAL_CONV0_W = np.array([[0, 0, 255], [0, 255, 0], [255, 0, 0]], np.float32).reshape(3, 3, 1, 1)
ORIGINAL_CONV0_B = np.array([-103.939, -116.779, -123.68], np.float32)


def synth_weights(mode: str, seed: int) -> Dict[str, np.ndarray]:
    """Deterministic stand-in weights (He-uniform) for a mode whose checkpoints are not
    available (``original``: trained_models/original_wct_models/*.t7 are absent from the
    reference snapshot, README.md:26).  Used by tests, goldens (G6) and bench config 3.
    Only ``Generator.random`` (uniform doubles) is used so the stream is stable across
    numpy versions.
    (Orthogonalised / variance-preserving layers were tried as a better-conditioned alternative: the 5-level cascade of the
    reference's own fp32 arithmetic sits 3.5e-3 from the exact fp64 result with either set at 384x384 -- random 512-channel
    stacks are chaotic under the whitening whatever the init -- so end-to-end parity in this mode is judged against the fp64
    "truth" arm, tests/test_hip_scale.py.)"""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for level in range(1, 6):
        for kind, layers in (("enc", encoder_layers(mode, level)), ("dec", decoder_layers(mode, level))):
            key = module_key(kind, level)
            if kind == "enc":
                w[key + ".conv0.weight"] = ORIGINAL_CONV0_W.copy()
                w[key + ".conv0.bias"] = ORIGINAL_CONV0_B.copy()
            for l in layers:
                a = np.sqrt(6.0 / (9 * l.cin))
                wt = (rng.random((l.cout, l.cin, 3, 3)) * 2.0 - 1.0) * a
                if kind == "dec" and l.cout == 3:
                    wt = wt * (1.0 / 128.0)  # features are O(100) (input x255): keep the decoded image O(1)
                w["%s.%s.weight" % (key, l.name)] = wt.astype(np.float32)
                w["%s.%s.bias" % (key, l.name)] = ((rng.random(l.cout) * 0.1) + (0.2 if (kind == "dec" and l.cout == 3) else 0.0)).astype(np.float32)
    return w


def synth_weights_conditioned(mode: str, seed: int, dense: float = 0.05) -> Dict[str, np.ndarray]:
    """A second deterministic stand-in set for a mode without checkpoints, built so that the reference's OWN fp32 arithmetic is
    NOT chaotic on it (fixture G15, tests/test_hip_scale.py).  Why `synth_weights` is: random ReLU stacks sit on one of two
    sides -- He-uniform filters keep a common-mode component, a third of the 512 channels die and many more fire at a handful of
    pixels (smallest live eigenvalue of the relu4_1 / relu5_1 covariance 1e-8 of the largest), and whitening multiplies every
    fp32-level difference between two implementations by lambda_min^-1/2; zero-mean filters keep every channel alive (cond 10..50)
    but a random zero-mean ReLU layer grows relative perturbations by sqrt(0.5 / 0.34) = 1.21, x140 over an encoder + decoder.
    Here every layer is (nearly) an isometry of a signed signal carried as a channel pair:
        channels [0, m) hold relu(x), channels [m, 2m) hold relu(-x);  the next layer forms x = a - b, applies a random matrix Q with
        unit-norm rows over the 3x3 x m patch (rows in a 9m-dimensional space, cout/2 << 9m: nearly orthonormal) and emits
        relu(Qx), relu(-Qx)
    so signal and perturbations travel with gain ~1, every channel is active on half of the pixels, and relu(z) / relu(-z) pairs of
    decorrelated z give feature covariances with cond ~ 10.  `dense` adds an unstructured U(-1, 1) component of that relative
    size to every weight (no exact structure for a kernel to exploit by accident).  The first conv (3 -> 64) takes the conv0 output
    as x; the last decoder conv (-> 3) is Q x scaled into [0, 1] around +0.5.  Only Generator.random is used (stable stream)."""
    rng = np.random.default_rng(seed)
    w: Dict[str, np.ndarray] = {}
    for level in range(1, 6):
        for kind, layers in (("enc", encoder_layers(mode, level)), ("dec", decoder_layers(mode, level))):
            key = module_key(kind, level)
            if kind == "enc":
                w[key + ".conv0.weight"] = ORIGINAL_CONV0_W.copy()
                w[key + ".conv0.bias"] = ORIGINAL_CONV0_B.copy()
            for l in layers:
                paired_in = l.cin != 3
                m = l.cin // 2 if paired_in else l.cin
                last = kind == "dec" and l.cout == 3
                rows = l.cout if last else l.cout // 2
                q = rng.random((rows, m, 3, 3)) * 2.0 - 1.0
                q /= np.sqrt((q ** 2).sum(axis=(1, 2, 3), keepdims=True))
                qin = np.concatenate([q, -q], axis=1) if paired_in else q          # x = a - b
                wt = qin if last else np.concatenate([qin, -qin], axis=0)            # relu(Qx), relu(-Qx)
                wt = wt + dense * np.sqrt(1.0 / (9 * m)) * (rng.random(wt.shape) * 2.0 - 1.0)
                if last:
                    wt = wt * (0.15 / 73.6)        # the signal carries the image's scale x255 (std of U[0,1) x 255 = 73.6)
                w["%s.%s.weight" % (key, l.name)] = wt.astype(np.float32)
                w["%s.%s.bias" % (key, l.name)] = (rng.random(l.cout) * 0.1 + (0.5 if last else 0.0)).astype(np.float32)
    return w
