"""Host-side mirror of the reference's call surface for the WCT stylisation path.

Same names, argument meaning and error behaviour as `PytorchWCT/util_wct.py` / `WCT.py`:

    wct = WCT(args)                      # util_wct.py:30-59   (args.mode, args.e1..e5, args.d1..d5, args.alpha)
    sF  = wct.e5(styleImg)               # WCT.py:100          NCHW fp32 CUDA tensors in and out
    csF = wct.transform(cF, sF, csF, a)  # util_wct.py:210-223
    img = wct.d5(csF)                    # WCT.py:105
    img = styleTransfer(wct.e5, wct.d5, contentImg, styleImg, csF)   # WCT.py:98-106

All arithmetic happens in libwct_hip.so (hand-written gfx950 kernels) through the C ABI of
include/wct_hip.h; PyTorch only provides device memory, the current HIP stream and (sharded.py)
torch.distributed.  There is no CPU fallback.
"""
from __future__ import annotations

import ctypes
import logging
import os
from ctypes import c_size_t, byref, c_int, c_void_p
from typing import Dict, Optional

import numpy as np
import torch

from . import lib as _lib
from . import model_zoo

_log = logging.getLogger("wct_hip")
_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT_16X_WEIGHTS = os.path.join(_PKG, "weights", "16x.npz")


def _fptr(a: np.ndarray):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _load_state(path: str) -> Dict[str, np.ndarray]:
    """A module's tensors from the reference's own checkpoint format (`{"epoch","model"}` or a bare
    state_dict, model_cd.py:712-718) -- torch is used for un-pickling only."""
    try:
        sd = torch.load(path, map_location="cpu", weights_only=True)   # plain tensors only: no arbitrary unpickling
    except Exception as e:
        if os.environ.get("WCT_ALLOW_UNSAFE_PICKLE") != "1":
            raise RuntimeError("%s is not a plain tensor checkpoint (%s); set WCT_ALLOW_UNSAFE_PICKLE=1 to unpickle it "
                               "with weights_only=False if you trust the file" % (path, e)) from e
        sd = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(sd, dict) and "model" in sd:
        sd = sd["model"]
    return {k: v.detach().cpu().numpy().astype(np.float32) for k, v in sd.items()}


class _Module:
    """`wct.e5` / `wct.d5`: callable like the reference's nn.Module on NCHW fp32 CUDA tensors."""

    def __init__(self, owner: "WCT", kind: str, level: int):
        self.owner, self.kind, self.level = owner, kind, level

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.kind == "enc":
            return self.owner.encode(self.level, x)
        return self.owner.decode(self.level, x)

    forward = __call__

    def __repr__(self):
        return "<wct_hip %s level %d (%s)>" % (self.kind, self.level, self.owner.mode)


class WCT:
    """Drop-in for util_wct.WCT.  `args` needs `.mode` ("16x" | "original" | None) and may carry
    `.e1..e5/.d1..d5` checkpoint paths (reference format, WCT.py:36-58), `.alpha`, `.numpy`.
    `weights` (a dict as produced by model_zoo.load_npz_weights / synth_weights) overrides the paths;
    with neither, mode 16x loads the packaged blob converted from the reference's checkpoints."""

    def __init__(self, args, weights: Optional[Dict[str, np.ndarray]] = None, device: Optional[int] = None):
        mode = getattr(args, "mode", None)
        if mode is None:
            mode = "original"
        if mode not in model_zoo.MODES:
            # the reference prints "Wrong mode. Please check." and exit(1)s (util_wct.py:57-59)
            raise ValueError("Wrong mode. Please check.")
        if not torch.cuda.is_available():
            raise RuntimeError("wct_hip needs a ROCm GPU (torch.cuda.is_available() is False); there is no CPU path")
        self.args = args
        self.mode = mode
        self.alpha = float(getattr(args, "alpha", 1.0))
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.stats_device = "cuda:%d" % self.device   # where style_export() tensors live (wct_hip/replicas.py, sharded.py)
        self.strict_range = True                      # see _stream(): f16x3 clamps are reported by the next call
        self.has_comm = False                         # comm_init(): the context owns an RCCL communicator (level_sharded)
        self._lib = _lib.load()
        self._ctx = c_void_p()
        _lib.check(self._lib, None, self._lib.wct_create(self.device, byref(self._ctx)))
        if weights is None:
            weights = self._weights_from_args(args, mode)
        self._load_modules(weights)
        # --numpy (util_wct.py:204-208): whiten_and_color_np = the same steps with + I on the content covariance (:143)
        self.numpy_variant = bool(getattr(args, "numpy", False))
        if self.numpy_variant:
            self._chk(self._lib.wct_set_numpy_variant(self._ctx, 1))
        for k in range(1, 6):
            setattr(self, "e%d" % k, _Module(self, "enc", k))
            setattr(self, "d%d" % k, _Module(self, "dec", k))

    # ------------------------------------------------------------------ construction
    @staticmethod
    def _weights_from_args(args, mode) -> Dict[str, np.ndarray]:
        paths = {("e%d" % k): getattr(args, "e%d" % k, None) for k in range(1, 6)}
        paths.update({("d%d" % k): getattr(args, "d%d" % k, None) for k in range(1, 6)})
        have = {k: p for k, p in paths.items() if p and os.path.exists(p)}
        if have and len(have) < 10:
            # the reference fails in torch.load on the first missing file (model_cd.py:712-718); never stylise with a
            # silent mix / substitute of weights
            missing = ["%s=%s" % (k, paths[k]) for k in sorted(paths) if k not in have]
            raise FileNotFoundError("checkpoints missing or not set for: " + ", ".join(missing))
        # per module: ".t7" (torch7, --mode original: model_original.py:24-30) or ".pth" (state_dict), as the reference asserts
        if len(have) == 10:
            bad = [p for p in have.values() if not (p.endswith(".pth") or (p.endswith(".t7") and mode == "original"))]
            if bad:
                raise ValueError("unsupported checkpoint extension (expected .pth%s): %s" % (" or .t7" if mode == "original" else "", ", ".join(bad)))
            _log.info("wct_hip: weights from the checkpoints in args.e1..d5 (%s ...)", paths["e1"])
            w = {}
            for key, p in paths.items():
                if p.endswith(".t7"):
                    state = model_zoo.load_t7_module(p, "enc" if key[0] == "e" else "dec", int(key[1]))
                else:
                    state = _load_state(p)
                for n, v in state.items():
                    if "aux" not in n:  # conv{k}1_aux heads are training-only (model_cd.py:700-704)
                        w["%s.%s" % (key, n)] = v
            return w
        if mode == "16x":
            _log.info("wct_hip: no checkpoint path exists -> packaged 16x weights %s", DEFAULT_16X_WEIGHTS)
            return model_zoo.load_npz_weights(DEFAULT_16X_WEIGHTS)
        if mode == "16x_kd2sd":
            raise FileNotFoundError("mode '16x_kd2sd' needs trained_models/wct_se_16x_new_sd_kd2sd/{1..5}SD.pth (WCT.py:60-70; "
                                    "not in the reference snapshot) -- pass the .pth paths in args.d1..d5 or weights=...")
        raise FileNotFoundError("mode 'original' needs the torch7 checkpoints of README.md:26 in args.e1..e5 / args.d1..d5 "
                                "(trained_models/original_wct_models/*.t7, not in the reference snapshot; read by wct_hip/t7.py) "
                                "-- or pass weights=...")

    def _load_modules(self, w: Dict[str, np.ndarray]):
        self._keep = []  # host arrays must outlive wct_load_module only, but keep them for clarity
        for k in range(1, 6):
            for kind, kid, layers in (("enc", _lib.KIND_ENC, model_zoo.encoder_layers(self.mode, k)),
                                      ("dec", _lib.KIND_DEC, model_zoo.decoder_layers(self.mode, k))):
                key = model_zoo.module_key(kind, k)
                arr = (_lib.WctLayer * len(layers))()
                hold = []
                for i, l in enumerate(layers):
                    wt = np.ascontiguousarray(w["%s.%s.weight" % (key, l.name)], np.float32)
                    bs = np.ascontiguousarray(w["%s.%s.bias" % (key, l.name)], np.float32)
                    if wt.shape != (l.cout, l.cin, 3, 3):
                        raise ValueError("%s.%s: weight shape %s != %s" % (key, l.name, wt.shape, (l.cout, l.cin, 3, 3)))
                    hold += [wt, bs]
                    arr[i] = _lib.WctLayer(l.cin, l.cout, int(l.pool_after), int(l.up_after), _fptr(wt), _fptr(bs))
                c0w = c0b = None
                if kind == "enc":
                    c0w = np.ascontiguousarray(w[key + ".conv0.weight"], np.float32).reshape(9)
                    c0b = np.ascontiguousarray(w[key + ".conv0.bias"], np.float32).reshape(3)
                    hold += [c0w, c0b]
                rc = self._lib.wct_load_module(self._ctx, kid, k, len(layers), arr,
                                               _fptr(c0w) if c0w is not None else None,
                                               _fptr(c0b) if c0b is not None else None)
                _lib.check(self._lib, self._ctx, rc)

    def __del__(self):
        try:
            if getattr(self, "_ctx", None) and self._ctx.value:
                self._lib.wct_destroy(self._ctx)
                self._ctx = c_void_p()
        except Exception:
            pass

    def cuda(self, device=None):  # `WCT(args).cuda()` (WCT.py:97): weights already live on the GPU
        return self

    def eval(self):
        return self

    # ------------------------------------------------------------------ helpers
    def _stream(self):
        """Called at the start of every compute method: bind the caller's stream and -- without synchronising -- raise if an
        EARLIER call that has completed meanwhile clamped an activation to the f16x3 range (include/wct_hip.h wct_range_poll):
        the deviation from the fp32 reference is then reported by the very next call on every path (stylize*, e*/d* modules,
        styleTransfer, the split-level calls of sharded.py / pipeline.py / replicas.py), not only by sync().  `strict_range =
        False` turns the check off; saturation_count(reset=True) or a reported sync() acknowledges and clears it."""
        self._chk(self._lib.wct_set_stream(self._ctx, c_void_p(torch.cuda.current_stream(self.device).cuda_stream)))
        if self.strict_range:
            n = ctypes.c_ulonglong()
            self._lib.wct_range_poll(self._ctx, byref(n))
            if n.value:
                raise OverflowError("libwct_hip: an earlier call clamped %d activation(s) to the f16x3 range (|x| >= 65504, or NaN input): "
                                    "its results deviate from the fp32 reference.  Use set_conv_mode('fp32') for these weights / inputs; "
                                    "saturation_count(reset=True) acknowledges the flag" % n.value)

    def range_flag(self) -> torch.Tensor:
        """The saturation counter NOW (in stream order) as a 1-element fp64 device tensor -- what a sharded run folds into the
        all-reduce of its moments so that every rank learns of a clamp on any rank (wct_range_flag_f64)."""
        t = torch.empty(1, device=self.stats_device, dtype=torch.float64)
        self._lib.wct_set_stream(self._ctx, c_void_p(torch.cuda.current_stream(self.device).cuda_stream))
        self._chk(self._lib.wct_range_flag_f64(self._ctx, t.data_ptr()))
        return t

    def _img(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4:
            if x.shape[0] != 1:
                raise ValueError("batch size must be 1 (the reference's DataLoader uses batch_size=1, WCT.py:92)")
            x = x[0]
        if x.dim() != 3 or x.shape[0] != 3:
            raise ValueError("expected a [1,3,H,W] or [3,H,W] image, got %s" % (tuple(x.shape),))
        return self._dev_f32(x)

    def _chk(self, rc):
        _lib.check(self._lib, self._ctx, rc)

    def saturation_count(self, reset: bool = False) -> int:
        """Threads of the f16x3 kernels that clamped an activation to +-65504 since the last reset / report (a deviation from the
        fp32 reference; synchronises the context's streams).  sync() raises OverflowError ONCE for a non-zero count and clears
        it; every compute method raises while a completed call's count is non-zero (strict_range)."""
        n = ctypes.c_ulonglong()
        _lib.check(self._lib, self._ctx, self._lib.wct_saturation_count(self._ctx, int(reset), byref(n)))
        return int(n.value)

    def _dev_f32(self, x: torch.Tensor) -> torch.Tensor:
        return x.to(device=self.stats_device, dtype=torch.float32).contiguous()

    def _dev_f64(self, x: torch.Tensor, numel: int, what: str) -> torch.Tensor:
        if x.numel() != numel:
            raise ValueError("%s: expected %d values, got %d" % (what, numel, x.numel()))
        return x.to(device=self.stats_device, dtype=torch.float64).contiguous()

    def _nhwc(self, feat: torch.Tensor, C: Optional[int] = None) -> torch.Tensor:
        f = feat[0] if feat.dim() == 4 else feat
        if f.dim() != 3:
            raise ValueError("expected an NHWC feature [1,h,w,C] or [h,w,C], got %s" % (tuple(feat.shape),))
        if C is not None and int(f.shape[2]) != C:
            raise ValueError("expected %d channels in the last (NHWC) dimension, got %s" % (C, tuple(f.shape)))
        return self._dev_f32(f)

    def _out_image(self, out: Optional[torch.Tensor], H: int, W: int) -> torch.Tensor:
        """Caller-provided result buffer of stylize(): fp32, on this context's device, contiguous, >= 3*H*W values."""
        if out is None:
            return torch.empty((3, H, W), device=self.stats_device, dtype=torch.float32)
        if out.dtype != torch.float32 or not out.is_cuda or out.device.index != self.device or not out.is_contiguous() \
                or out.numel() < 3 * H * W:
            raise ValueError("out must be a contiguous fp32 tensor on cuda:%d with at least %d values" % (self.device, 3 * H * W))
        return out

    def feature_shape(self, level: int, H: int, W: int):
        C, h, w = c_int(), c_int(), c_int()
        self._chk(self._lib.wct_feature_shape(self._ctx, level, H, W, byref(C), byref(h), byref(w)))
        return C.value, h.value, w.value

    # ------------------------------------------------------------------ reference surface
    @torch.no_grad()
    def encode(self, level: int, img: torch.Tensor, layout: str = "nchw") -> torch.Tensor:
        x = self._img(img)
        H, W = int(x.shape[1]), int(x.shape[2])
        C, h, w = self.feature_shape(level, H, W)
        nchw = layout == "nchw"
        out = torch.empty((1, C, h, w) if nchw else (1, h, w, C), device=x.device, dtype=torch.float32)
        self._stream()
        self._chk(self._lib.wct_encode(self._ctx, level, x.data_ptr(), H, W, out.data_ptr(),
                                       _lib.LAYOUT_NCHW if nchw else _lib.LAYOUT_NHWC))
        return out

    @torch.no_grad()
    def decode(self, level: int, feat: torch.Tensor, layout: str = "nchw") -> torch.Tensor:
        f = feat[0] if feat.dim() == 4 else feat
        if f.dim() != 3:
            raise ValueError("expected a [1,C,h,w] / [C,h,w] (nchw) or [1,h,w,C] / [h,w,C] (nhwc) feature, got %s" % (tuple(feat.shape),))
        f = self._dev_f32(f)
        nchw = layout == "nchw"
        C = model_zoo.feature_channels(self.mode, level)
        if nchw:
            c, h, w = (int(s) for s in f.shape)
        else:
            h, w, c = (int(s) for s in f.shape)
        if c != C:
            raise ValueError("decoder %d expects %d channels, got %d" % (level, C, c))
        out = torch.empty((1, 3, h << (level - 1), w << (level - 1)), device=f.device, dtype=torch.float32)
        self._stream()
        self._chk(self._lib.wct_decode(self._ctx, level, f.data_ptr(), h, w,
                                       _lib.LAYOUT_NCHW if nchw else _lib.LAYOUT_NHWC, out.data_ptr()))
        return out

    @torch.no_grad()
    def transform(self, cF: torch.Tensor, sF: torch.Tensor, csF: Optional[torch.Tensor] = None,
                  alpha: Optional[float] = None) -> torch.Tensor:
        """util_wct.py:210-223.  cF [C,h,w], sF [C,h',w'] fp32 (CPU like the reference, or CUDA) ->
        csF [1,C,h,w] fp32 on the GPU.  If `csF` is given it is resized in place, filled and returned
        (the reference's `csF.data.resize_().copy_()`, which silently stopped working in torch>=1.1)."""
        alpha = self.alpha if alpha is None else float(alpha)
        dev = self.stats_device
        c = self._dev_f32(cF[0] if cF.dim() == 4 else cF)
        s = self._dev_f32(sF[0] if sF.dim() == 4 else sF)
        if c.dim() != 3 or s.dim() != 3 or c.shape[0] != s.shape[0]:
            raise ValueError("transform expects cF [C,h,w] and sF [C,h',w'] with equal C")
        C, h, w = (int(v) for v in c.shape)
        hs, ws = int(s.shape[1]), int(s.shape[2])
        out = torch.empty((1, C, h, w), device=dev, dtype=torch.float32)
        self._stream()
        self._chk(self._lib.wct_transform(self._ctx, c.data_ptr(), C, h, w, s.data_ptr(), hs, ws, alpha,
                                          _lib.LAYOUT_NCHW, out.data_ptr()))
        if csF is not None and csF.is_cuda:
            csF.resize_(out.shape).copy_(out)
            return csF
        return out

    # ------------------------------------------------------------------ split form (used by the sharded path)
    @torch.no_grad()
    def moments(self, feat_nhwc: torch.Tensor, x0: int = 0, x1: Optional[int] = None):
        """Raw fp64 sums over columns [x0,x1) of an NHWC feature [1,h,w,C]: (n, sum[C], sumsq[C,C])."""
        f = self._nhwc(feat_nhwc)
        h, w, C = (int(v) for v in f.shape)
        x1 = w if x1 is None else x1
        s = torch.empty(C, device=f.device, dtype=torch.float64)
        ss = torch.empty(C, C, device=f.device, dtype=torch.float64)
        self._stream()
        self._chk(self._lib.wct_moments(self._ctx, f.data_ptr(), C, h, w, x0, x1, s.data_ptr(), ss.data_ptr()))
        return float(h * (x1 - x0)), s, ss

    @torch.no_grad()
    def solve(self, n_c, sum_c, sumsq_c, n_s, sum_s, sumsq_s, alpha: Optional[float] = None, want_info=False):
        alpha = self.alpha if alpha is None else float(alpha)
        C = int(sum_c.numel())
        sum_c, sumsq_c = self._dev_f64(sum_c, C, "sum_c"), self._dev_f64(sumsq_c, C * C, "sumsq_c")
        sum_s, sumsq_s = self._dev_f64(sum_s, C, "sum_s"), self._dev_f64(sumsq_s, C * C, "sumsq_s")
        M = torch.empty(C, C, device=sum_c.device, dtype=torch.float64)
        b = torch.empty(C, device=sum_c.device, dtype=torch.float64)
        info = (c_int * 2)()
        self._stream()
        self._chk(self._lib.wct_solve(self._ctx, C, float(n_c), sum_c.data_ptr(), sumsq_c.data_ptr(), float(n_s),
                                      sum_s.data_ptr(), sumsq_s.data_ptr(), alpha, M.data_ptr(), b.data_ptr(),
                                      info if want_info else None))
        return (M, b, (info[0], info[1])) if want_info else (M, b)

    @torch.no_grad()
    def decode_affine(self, level: int, feat_nhwc: torch.Tensor, M: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
        C = model_zoo.feature_channels(self.mode, level)
        f = self._nhwc(feat_nhwc, C)
        M, b = self._dev_f64(M, C * C, "M"), self._dev_f64(b, C, "b")
        h, w, _ = (int(v) for v in f.shape)
        out = torch.empty((1, 3, h << (level - 1), w << (level - 1)), device=f.device, dtype=torch.float32)
        self._stream()
        self._chk(self._lib.wct_decode_affine(self._ctx, level, f.data_ptr(), h, w, M.data_ptr(), b.data_ptr(), out.data_ptr()))
        return out

    # ------------------------------------------------------------------ split level (content-sharded runs)
    @torch.no_grad()
    def style_prepare(self, styleImg: torch.Tensor, levels=(5, 4, 3, 2, 1)):
        """Style side of the given levels on the context's side stream (overlaps whatever follows)."""
        s = self._img(styleImg)
        self._style_keep = s   # the side stream reads it asynchronously
        mask = 0
        for L in levels:
            mask |= 1 << int(L)
        self._stream()
        self._chk(self._lib.wct_style_prepare_levels(self._ctx, s.data_ptr(), int(s.shape[1]), int(s.shape[2]), mask))

    def style_stats_count(self, level: int) -> int:
        n = c_size_t()
        self._chk(self._lib.wct_style_stats_count(self._ctx, level, byref(n)))
        return int(n.value)

    def style_export(self, level: int) -> torch.Tensor:
        """Style statistics of a prepared level as one fp64 vector: cov_s^(1/2) [C*C] then mu_s [C]."""
        n = c_size_t()
        self._chk(self._lib.wct_style_stats_count(self._ctx, level, byref(n)))
        buf = torch.empty(n.value, device=self.stats_device, dtype=torch.float64)
        self._stream()
        self._chk(self._lib.wct_style_export(self._ctx, level, buf.data_ptr()))
        return buf

    def style_import(self, level: int, stats: torch.Tensor):
        n = c_size_t()
        self._chk(self._lib.wct_style_stats_count(self._ctx, level, byref(n)))
        if stats.dtype != torch.float64 or not stats.is_cuda or stats.numel() != n.value:
            raise ValueError("style_import: expected %d fp64 values on the GPU" % n.value)
        stats = stats.to(self.stats_device).contiguous()
        self._stream()
        self._chk(self._lib.wct_style_import(self._ctx, level, stats.data_ptr()))

    @torch.no_grad()
    def stylize_prepared(self, contentImg: torch.Tensor, alpha: Optional[float] = None, num_run: int = 1,
                         out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The cascade against the style statistics already in the context (style_prepare / style_import): content x style
        batches pay the style side once per style (data_loader.py:32-36 builds the Cartesian product)."""
        alpha = self.alpha if alpha is None else float(alpha)
        c = self._img(contentImg)
        H, W = int(c.shape[1]), int(c.shape[2])
        out = self._out_image(out, H, W)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_stylize_prepared(self._ctx, c.data_ptr(), H, W, alpha, int(num_run), out.data_ptr(), byref(ho), byref(wo)))
        return out.view(-1)[: 3 * ho.value * wo.value].view(1, 3, ho.value, wo.value)

    # ------------------------------------------------------------------ image edge (ToTensor / save_image on the device)
    def _u8(self, img: torch.Tensor) -> torch.Tensor:
        if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
            raise ValueError("expected a uint8 H x W x 3 image, got %s %s" % (img.dtype, tuple(img.shape)))
        return img.to(self.stats_device).contiguous()

    @torch.no_grad()
    def to_tensor_u8(self, img_u8: torch.Tensor) -> torch.Tensor:
        """transforms.ToTensor() of data_loader.py:57-58 on the GPU: uint8 HWC -> fp32 1x3xHxW in [0,1]."""
        x = self._u8(img_u8)
        H, W = int(x.shape[0]), int(x.shape[1])
        out = torch.empty((1, 3, H, W), device=x.device, dtype=torch.float32)
        self._stream()
        self._chk(self._lib.wct_u8_to_planar(self._ctx, x.data_ptr(), H, W, out.data_ptr()))
        return out

    def resize_shape(self, H: int, W: int, size: int):
        """Output (H, W) of transforms.Resize(size) (torchvision 0.2.1: the smaller edge becomes `size`; 0 = no resize)."""
        oh, ow = c_int(), c_int()
        if self._lib.wct_resize_shape(int(H), int(W), int(size), byref(oh), byref(ow)) != 0:
            raise ValueError("resize_shape: bad arguments %s x %s -> %s" % (H, W, size))
        return oh.value, ow.value

    @torch.no_grad()
    def resize_u8(self, img_u8: torch.Tensor, size, to_tensor: bool = False) -> torch.Tensor:
        """transforms.Resize(size) of data_loader.py:52-56 on the GPU, bit-exact with Pillow's bilinear Image.resize: `size` is an
        int (smaller edge, torchvision's rule) or an (oH, oW) pair.  to_tensor=True returns ToTensor()'s fp32 1x3xoHxoW instead of
        uint8 HWC (one pass less)."""
        x = self._u8(img_u8)
        H, W = int(x.shape[0]), int(x.shape[1])
        oH, oW = self.resize_shape(H, W, size) if isinstance(size, int) else (int(size[0]), int(size[1]))
        self._stream()
        if to_tensor:
            out = torch.empty((1, 3, oH, oW), device=x.device, dtype=torch.float32)
            self._chk(self._lib.wct_resize_u8_to_planar(self._ctx, x.data_ptr(), H, W, out.data_ptr(), oH, oW))
        else:
            out = torch.empty((oH, oW, 3), device=x.device, dtype=torch.uint8)
            self._chk(self._lib.wct_resize_u8(self._ctx, x.data_ptr(), H, W, out.data_ptr(), oH, oW))
        return out

    @torch.no_grad()
    def to_u8(self, img: torch.Tensor, round_mode: int = 0) -> torch.Tensor:
        """save_image's conversion (WCT.py:128; torchvision 0.2.1: mul(255).clamp(0,255).byte()) on the GPU: fp32 CHW -> uint8 HWC."""
        x = self._img(img)
        H, W = int(x.shape[1]), int(x.shape[2])
        out = torch.empty((H, W, 3), device=x.device, dtype=torch.uint8)
        self._stream()
        self._chk(self._lib.wct_planar_to_u8(self._ctx, x.data_ptr(), H, W, out.data_ptr(), int(round_mode)))
        return out

    @torch.no_grad()
    def stylize_u8(self, content_u8: torch.Tensor, style_u8: torch.Tensor, alpha: Optional[float] = None, num_run: int = 1,
                   round_mode: int = 0) -> torch.Tensor:
        """uint8 HWC content + style -> uint8 HWC result (ToTensor -> cascade -> save_image conversion), one call."""
        alpha = self.alpha if alpha is None else float(alpha)
        c, s = self._u8(content_u8), self._u8(style_u8)
        H, W, Hs, Ws = int(c.shape[0]), int(c.shape[1]), int(s.shape[0]), int(s.shape[1])
        out = torch.empty((H, W, 3), device=c.device, dtype=torch.uint8)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_stylize_u8(self._ctx, c.data_ptr(), H, W, s.data_ptr(), Hs, Ws, alpha, int(num_run), out.data_ptr(),
                                           byref(ho), byref(wo), int(round_mode)))
        return out.view(-1)[: 3 * ho.value * wo.value].view(ho.value, wo.value, 3)

    @torch.no_grad()
    def content_encode(self, level: int, img: torch.Tensor, x0: int = 0, x1: int = -1):
        """cF = encoder(img), kept inside the context; returns (h, w, sum[C], sumsq[C,C]) over feature columns [x0,x1)."""
        x = self._img(img)
        self._content_keep = x
        H, W = int(x.shape[1]), int(x.shape[2])
        C = model_zoo.feature_channels(self.mode, level)
        s = torch.empty(C, device=x.device, dtype=torch.float64)
        ss = torch.empty(C, C, device=x.device, dtype=torch.float64)
        h, w = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_content_encode(self._ctx, level, x.data_ptr(), H, W, x0, x1, s.data_ptr(), ss.data_ptr(), byref(h), byref(w)))
        return h.value, w.value, s, ss

    @torch.no_grad()
    def content_solve(self, level: int, n_c: float, sum_c: torch.Tensor, sumsq_c: torch.Tensor, alpha: Optional[float] = None):
        alpha = self.alpha if alpha is None else float(alpha)
        C = model_zoo.feature_channels(self.mode, level)
        sum_c, sumsq_c = self._dev_f64(sum_c, C, "sum_c"), self._dev_f64(sumsq_c, C * C, "sumsq_c")
        M = torch.empty(C, C, device=sum_c.device, dtype=torch.float64)
        b = torch.empty(C, device=sum_c.device, dtype=torch.float64)
        self._stream()
        self._chk(self._lib.wct_content_solve(self._ctx, level, float(n_c), sum_c.data_ptr(), sumsq_c.data_ptr(), alpha, M.data_ptr(), b.data_ptr()))
        return M, b

    @torch.no_grad()
    def content_decode(self, level: int, M: torch.Tensor, b: torch.Tensor, H: int, W: int) -> torch.Tensor:
        """H, W: size of the image wct.content_encode() was given (the output is floor-shrunk like the reference's)."""
        C, h, w = self.feature_shape(level, H, W)
        M, b = self._dev_f64(M, C * C, "M"), self._dev_f64(b, C, "b")
        out = torch.empty((1, 3, h << (level - 1), w << (level - 1)), device=M.device, dtype=torch.float32)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_content_decode(self._ctx, level, M.data_ptr(), b.data_ptr(), out.data_ptr(), byref(ho), byref(wo)))
        assert (ho.value, wo.value) == tuple(out.shape[2:])
        return out

    # ------------------------------------------------------------------ RCCL behind the boundary (include/wct_hip.h wct_comm_*, wct_level_sharded)
    def comm_init(self, dist, group=None) -> None:
        """Give this engine's context its own RCCL communicator over the ranks of `dist` (torch.distributed, any backend: it only carries
        the 128-byte unique id from rank 0): every rank calls this once.  The library then runs the per-level all-reduce of a sharded
        job itself (level_sharded); RCCL is the librccl.so torch has loaded -- one RCCL per process."""
        import os as _os
        path = _os.path.join(_os.path.dirname(torch.__file__), "lib", "librccl.so")
        if self._lib.wct_comm_load(path.encode() if _os.path.exists(path) else None) != 0:
            raise RuntimeError("libwct_hip: librccl.so could not be loaded (%s)" % (path if _os.path.exists(path) else "the loader's search path, /opt/rocm/lib"))
        rank, world = dist.get_rank(), dist.get_world_size()
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            if self._lib.wct_comm_unique_id(buf) != 0:
                raise RuntimeError("libwct_hip: ncclGetUniqueId failed")
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            buf = ctypes.create_string_buffer(box[0], 128)
        with torch.cuda.device(self.device):
            self._chk(self._lib.wct_comm_init(self._ctx, world, rank, buf))
        self.has_comm = True

    def comm_destroy(self) -> None:
        self._chk(self._lib.wct_comm_destroy(self._ctx))
        self.has_comm = False
        self._coll_keep = None

    def comm_info(self):
        """(nranks, rank) of the communicator / transport the context holds; (0, 0) without one."""
        n, r = c_int(), c_int()
        self._chk(self._lib.wct_comm_info(self._ctx, byref(n), byref(r)))
        return n.value, r.value

    def comm_attach_collectives(self, all_reduce, broadcast, sendrecv, nranks: int, rank: int) -> None:
        """Install a caller-supplied transport for stylize_sharded / level_sharded (include/wct_hip.h wct_collectives): three Python
        callables `all_reduce(buf_ptr, count, stream) -> int`, `broadcast(buf_ptr, nbytes, root, stream) -> int`,
        `sendrecv(ops, stream) -> int` with ops = [(peer, is_send, buf_ptr, nbytes)]; device pointers and the HIP stream arrive as
        integers, 0 = success.  (Test infrastructure drives the C cascade through rank threads this way; a production transport would be C.)"""
        def _sr(user, ops, n, stream):
            return int(sendrecv([(ops[i].peer, ops[i].is_send, ops[i].buf, ops[i].bytes) for i in range(n)], stream))
        table = _lib.WctCollectives(None, _lib.ALL_REDUCE_FN(lambda user, buf, count, stream: int(all_reduce(buf, count, stream))),
                                    _lib.BROADCAST_FN(lambda user, buf, nbytes, root, stream: int(broadcast(buf, nbytes, root, stream))),
                                    _lib.SENDRECV_FN(_sr))
        self._chk(self._lib.wct_comm_attach_collectives(self._ctx, byref(table), int(nranks), int(rank)))
        self._coll_keep = table          # the C side copied the table; the callback thunks must outlive it
        self.has_comm = True

    def comm_selftest(self) -> None:
        """Exercise the context's transport between the job's ranks with known data (all-reduce, broadcast, ring and neighbour
        send / recv) and raise if anything comes back wrong (wct_comm_selftest; collective: every rank calls it)."""
        self._stream()
        self._chk(self._lib.wct_comm_selftest(self._ctx))

    def shard_geometry(self, W_total: int, nranks: int, rank: int, halo_mode: str = "auto"):
        """-> (own0, own1, in0, in1, resolved halo mode) of `rank` in an `nranks`-strip job over a W_total-wide content (wct_shard_geometry)."""
        v = [c_int() for _ in range(5)]
        rc = self._lib.wct_shard_geometry(int(W_total), int(nranks), int(rank), _lib.HALO_MODES[halo_mode], *[byref(x) for x in v])
        if rc != 0:
            raise ValueError("shard_geometry: width %d cannot be cut into %d strips under halo mode %r" % (W_total, nranks, halo_mode))
        return v[0].value, v[1].value, v[2].value, v[3].value, {0: "auto", 1: "recompute", 2: "exchange"}[v[4].value]

    @torch.no_grad()
    def stylize_sharded(self, content_ext: torch.Tensor, style: torch.Tensor, W_total: int, in0: int, in1: int, alpha: Optional[float] = None,
                        halo_mode: str = "auto", style_mode: str = "auto", broadcast_map: bool = False,
                        range_total: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None, fast_fold: bool = False) -> torch.Tensor:
        """The WHOLE column-sharded cascade in ONE library call (wct_stylize_sharded): this rank's strip + level-5 margin in
        (columns [in0, in1) of the W_total-wide content, see shard_geometry), its owned columns of the stylised image out
        ([1, 3, H', own_w']).  Geometry, crops, the style side (strips / owner / replicate), the per-level all-reduce, the optional
        broadcasts and the neighbour exchange all run inside the library on the context's communicator (comm_init) or transport."""
        alpha = self.alpha if alpha is None else float(alpha)
        c, s = self._img(content_ext), self._img(style)
        H, Win = int(c.shape[1]), int(c.shape[2])
        if Win != in1 - in0:
            raise ValueError("stylize_sharded: content_ext has %d columns, [in0, in1) = [%d, %d)" % (Win, in0, in1))
        nr, rk = self.comm_info()
        own0, own1 = self.shard_geometry(W_total, max(nr, 1), rk, halo_mode)[:2]
        if out is None:
            out = torch.empty(3 * H * (own1 - own0), device=c.device, dtype=torch.float32)
        elif out.dtype != torch.float32 or not out.is_cuda or not out.is_contiguous() or out.numel() < 3 * H * (own1 - own0):
            raise ValueError("out must be a contiguous fp32 CUDA tensor with at least %d values" % (3 * H * (own1 - own0)))
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_stylize_sharded(self._ctx, c.data_ptr(), H, int(W_total), int(in0), int(in1), s.data_ptr(), int(s.shape[1]), int(s.shape[2]),
                                                alpha, _lib.HALO_MODES[halo_mode], _lib.STYLE_MODES[style_mode], (_lib.SHARD_BROADCAST_MAP if broadcast_map else 0) | (_lib.SHARD_FAST_FOLD if fast_fold else 0),
                                                out.data_ptr(), byref(ho), byref(wo), range_total.data_ptr() if range_total is not None else None))
        self._style_keep = s
        return out.view(-1)[: 3 * ho.value * wo.value].view(1, 3, ho.value, wo.value)

    @torch.no_grad()
    def style_moments(self, level: int, style_strip: torch.Tensor, x0: int = 0, x1: int = -1):
        """Encoder of a STRIP of the style image + raw fp64 moments over its feature columns [x0, x1) -> (sum[C], sumsq[C, C]) on the
        context's side stream, visible to the caller's stream (wct_style_moments; the style side of sharded.py's "strips" mode)."""
        x = self._img(style_strip)
        self._style_keep = x
        C = model_zoo.feature_channels(self.mode, level)
        s = torch.empty(C, device=x.device, dtype=torch.float64)
        ss = torch.empty(C, C, device=x.device, dtype=torch.float64)
        self._stream()
        self._chk(self._lib.wct_style_moments(self._ctx, level, x.data_ptr(), int(x.shape[1]), int(x.shape[2]), int(x0), int(x1), s.data_ptr(), ss.data_ptr()))
        return s, ss

    @torch.no_grad()
    def style_solve(self, level: int, n_s: float, sum_s: torch.Tensor, sumsq_s: torch.Tensor) -> None:
        """Global style moments of a level -> cov_s^(1/2), mu_s inside the context (what style_prepare leaves), on the side stream."""
        C = model_zoo.feature_channels(self.mode, level)
        sum_s, sumsq_s = self._dev_f64(sum_s, C, "sum_s"), self._dev_f64(sumsq_s, C * C, "sumsq_s")
        self._solve_keep = getattr(self, "_solve_keep", {})
        self._solve_keep[level] = (sum_s, sumsq_s)       # read asynchronously by the side stream
        self._stream()
        self._chk(self._lib.wct_style_solve(self._ctx, level, float(n_s), sum_s.data_ptr(), sumsq_s.data_ptr()))

    @torch.no_grad()
    def level_sharded(self, level: int, img: torch.Tensor, x0: int, x1: int, n_total: float, alpha: Optional[float] = None,
                      range_total: Optional[torch.Tensor] = None) -> torch.Tensor:
        """One level of a column-sharded cascade in ONE library call (encoder + owned-column moments -> ncclAllReduce -> solve -> fold ->
        decoder; wct_level_sharded).  img: this rank's strip + halo [3, H, W]; [x0, x1): owned FEATURE columns; n_total: feature
        pixels of the whole image.  range_total: optional 1-element fp64 device tensor receiving the node-wide f16x3 clamp count."""
        alpha = self.alpha if alpha is None else float(alpha)
        x = self._img(img)
        H, W = int(x.shape[1]), int(x.shape[2])
        C, h, w = self.feature_shape(level, H, W)
        out = torch.empty((1, 3, h << (level - 1), w << (level - 1)), device=x.device, dtype=torch.float32)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_level_sharded(self._ctx, level, x.data_ptr(), H, W, int(x0), int(x1), float(n_total), alpha, out.data_ptr(),
                                              byref(ho), byref(wo), range_total.data_ptr() if range_total is not None else None))
        assert (ho.value, wo.value) == tuple(out.shape[2:])
        return out

    # ------------------------------------------------------------------ fused level / cascade
    @torch.no_grad()
    def style_transfer_level(self, level: int, contentImg: torch.Tensor, styleImg: torch.Tensor,
                             alpha: Optional[float] = None) -> torch.Tensor:
        alpha = self.alpha if alpha is None else float(alpha)
        c, s = self._img(contentImg), self._img(styleImg)
        H, W, Hs, Ws = int(c.shape[1]), int(c.shape[2]), int(s.shape[1]), int(s.shape[2])
        _, h, w = self.feature_shape(level, H, W)
        out = torch.empty((1, 3, h << (level - 1), w << (level - 1)), device=c.device, dtype=torch.float32)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_style_transfer_level(self._ctx, level, c.data_ptr(), H, W, s.data_ptr(), Hs, Ws, alpha,
                                                     out.data_ptr(), byref(ho), byref(wo)))
        assert (ho.value, wo.value) == tuple(out.shape[2:])
        return out

    @torch.no_grad()
    def stylize(self, contentImg: torch.Tensor, styleImg: torch.Tensor, alpha: Optional[float] = None,
                num_run: int = 1, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The 5 -> 1 cascade of WCT.py:120-125 in one call (no host round trips, no allocation after the
        first call of a size)."""
        alpha = self.alpha if alpha is None else float(alpha)
        c, s = self._img(contentImg), self._img(styleImg)
        H, W, Hs, Ws = int(c.shape[1]), int(c.shape[2]), int(s.shape[1]), int(s.shape[2])
        out = self._out_image(out, H, W)
        ho, wo = c_int(), c_int()
        self._stream()
        self._chk(self._lib.wct_stylize(self._ctx, c.data_ptr(), H, W, s.data_ptr(), Hs, Ws, alpha, int(num_run),
                                        out.data_ptr(), byref(ho), byref(wo)))
        return out.view(-1)[: 3 * ho.value * wo.value].view(1, 3, ho.value, wo.value)

    def set_conv_mode(self, mode: str):
        """'f16x3' (default: split-f16 MFMA, fp32-class accuracy) or 'fp32' (exact fp32 MFMA)."""
        self._chk(self._lib.wct_set_conv_mode(self._ctx, {"fp32": 0, "f16x3": 1}[mode]))

    def debug_set(self, key: str, value) -> None:
        """Experiment switches of the context ("fuse", "sp", "l1fuse", "u8fuse", "upconv"): kernel formulations of the same operators."""
        self._chk(self._lib.wct_debug_set(self._ctx, key.encode(), float(value)))

    def debug_get(self, key: str) -> float:
        """Health counters of the context: "nscoop_solves", "nscoop_aborts", "nscoop_off" (include/wct_hip.h wct_debug_get)."""
        v = ctypes.c_double()
        self._chk(self._lib.wct_debug_get(self._ctx, key.encode(), byref(v)))
        return float(v.value)

    def set_overlap(self, on: bool):
        self._chk(self._lib.wct_set_overlap(self._ctx, int(on)))

    def reserve(self, H, W, Hs, Ws):
        self._chk(self._lib.wct_reserve(self._ctx, H, W, Hs, Ws))

    def sync(self):
        self._chk(self._lib.wct_sync(self._ctx))

    # ------------------------------------------------------------------ profiling (bench.py roofline leg)
    def profile(self, on: bool):
        self._chk(self._lib.wct_profile_enable(self._ctx, int(on)))

    def profile_reset(self):
        self._chk(self._lib.wct_profile_reset(self._ctx))

    def profile_read(self):
        n = c_int()
        buf = (_lib.WctProfEntry * 64)()
        self._chk(self._lib.wct_profile_read(self._ctx, buf, 64, byref(n)))
        return [{"name": buf[i].name.decode(), "ms": buf[i].ms, "flops": buf[i].flops, "bytes": buf[i].bytes,
                 "launches": buf[i].launches} for i in range(min(n.value, 64))]


@torch.no_grad()
def styleTransfer(encoder, decoder, contentImg, styleImg, csF=None, alpha=None):
    """WCT.py:98-106.  With an encoder/decoder pair of one wct_hip.WCT level this runs the fused HIP
    level (features never leave HBM, M/b folded into the decoder); otherwise it falls back to the
    reference's literal op sequence on the given callables (still GPU: they are wct_hip modules)."""
    if isinstance(encoder, _Module) and isinstance(decoder, _Module) and encoder.owner is decoder.owner \
            and encoder.level == decoder.level and encoder.kind == "enc" and decoder.kind == "dec":
        return encoder.owner.style_transfer_level(encoder.level, contentImg, styleImg, alpha)
    owner = getattr(encoder, "owner", None) or getattr(decoder, "owner", None)
    if owner is None:
        raise TypeError("styleTransfer needs wct_hip modules")
    sF = encoder(styleImg)
    cF = encoder(contentImg)
    csF = owner.transform(cF.squeeze(0), sF.squeeze(0), csF, alpha)
    return decoder(csF)
