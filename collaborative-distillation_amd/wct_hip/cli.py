"""Command line of the stylisation path: the flags, pairing, naming and log lines of PytorchWCT/WCT.py, on libwct_hip.

    python -m wct_hip.cli --mode 16x --contentPath content --stylePath style --outf stylized_results [--alpha 0.6] ...

What the reference script does around the hot path and where it lives here:
  WCT.py:15-35      argparse flags                     -> build_parser()   (same names, defaults and choices)
  WCT.py:37-75      checkpoint paths per --mode        -> checkpoint_args()
  data_loader.py:22-36   content x style pairs filtered by --picked_*_mark, listdir order   -> list_pairs()
  data_loader.py:50-59   PIL decode (.convert('RGB')), optional Resize, ToTensor            -> load_rgb_u8() + wct_resize_u8_to_planar / wct_u8_to_planar (GPU)
  WCT.py:120-125    the 5-level cascade, --num_run times -> wct_stylize (one C call)
  WCT.py:127-128    output name and save_image          -> out_name() + wct_planar_to_u8 (GPU) + PIL save
A frame crosses PCIe as uint8 (3 B/px each way).  The reference's loop is strictly serial -- decode, .cuda(), cascade, save_image
(WCT.py:112-131: its timer covers cascade + save) -- which leaves the GPU idle for ~90 % of a folder run once the cascade takes 10 ms:
`--pipeline N` (default 3; 0 = the serial loop) keeps N pairs in flight: a decode-ahead pool (PIL decode + copy into pinned memory, GIL
released), asynchronous H2D / D2H around the GPU work of the SAME single engine in the SAME order, and a writer pool for Image.save.
Output files are byte-identical to the serial loop's (tests/test_cli.py).  --numpy selects the reference's whiten_and_color_np semantics (+ I on the
content covariance); --synthesis is rejected (broken in the reference: data_loader.py:74 calls torch.rand_like on a PIL image).
Decoding/encoding files needs Pillow on the host (the reference's own dependency); the GPU library is mandatory: there
is no CPU fallback.
"""
from __future__ import annotations

import argparse
import os
import sys
import time
from typing import List, Optional, Tuple

IMG_EXT = (".png", ".jpg", ".jpeg")   # data_loader.py:15


def build_parser() -> argparse.ArgumentParser:
    p = argparse.ArgumentParser(description="WCT on MI355X (libwct_hip)")
    p.add_argument("--UHD_contentPath", type=str, default="content/UHD_content")
    p.add_argument("--UHD_stylePath", type=str, default="style/UHD_style")
    p.add_argument("--contentPath", type=str, default="content")
    p.add_argument("--stylePath", type=str, default="style")
    p.add_argument("--texturePath", type=str, default="style/texture")
    p.add_argument("--outf", type=str, default="stylized_results", help="folder to output images")
    p.add_argument("--picked_content_mark", type=str, default=".")
    p.add_argument("--picked_style_mark", type=str, default=".")
    p.add_argument("--mode", type=str, default=None, choices=["original", "16x", "16x_kd2sd"], help="to choose different trained models")
    p.add_argument("--UHD", action="store_true", help="if use the UHD images")
    p.add_argument("--synthesis", action="store_true", help="for style synthesis")
    p.add_argument("--content_size", type=int, default=0, help="resize content, leave it to 0 if not resize")
    p.add_argument("--style_size", type=int, default=0, help="resize style, leave it to 0 if not resize")
    p.add_argument("--alpha", type=float, default=1, help="hyperparameter to blend wct feature and content feature")
    p.add_argument("--log_mark", type=str, default=time.strftime("%Y%m%d-%H%M"))
    p.add_argument("--num_run", type=int, default=1, help="you can run WCT for multiple times")
    p.add_argument("--debug", action="store_true")
    p.add_argument("--numpy", action="store_true", help="if use numpy for stylization rather than torch")
    # not in the reference: where its trained_models/ directory is (default: the paths of WCT.py:37-75 relative to the cwd)
    p.add_argument("--models_root", type=str, default="..", help="directory that holds trained_models/ (reference layout)")
    p.add_argument("--round", dest="round_mode", type=int, default=0, choices=[0, 1],
                   help="uint8 conversion of the result: 0 = truncation (torchvision 0.2.1, the reference's pin), 1 = +0.5")
    p.add_argument("--pipeline", type=int, default=3, help="pairs in flight (decode-ahead, async copies, writer pool); 0 = the reference's serial loop")
    p.add_argument("--io_threads", type=int, default=8, help="threads of the decode pool and of the writer pool (--pipeline > 0)")
    return p


def checkpoint_args(args) -> None:
    """e1..e5 / d1..d5 exactly as WCT.py:37-75 assigns them (relative to --models_root)."""
    root = os.path.join(args.models_root, "trained_models")
    if args.mode == "original" or args.mode is None:
        enc = [os.path.join(root, "original_wct_models", "vgg_normalised_conv%d_1.t7" % k) for k in range(1, 6)]
        dec = [os.path.join(root, "original_wct_models", "feature_invertor_conv%d_1.t7" % k) for k in range(1, 6)]
    else:
        sd = "wct_se_16x_new_sd" if args.mode == "16x" else "wct_se_16x_new_sd_kd2sd"
        enc = [os.path.join(root, "wct_se_16x_new", "%dSE.pth" % k) for k in range(1, 6)]
        dec = [os.path.join(root, sd, "%dSD.pth" % k) for k in range(1, 6)]
    for k in range(1, 6):
        setattr(args, "e%d" % k, enc[k - 1])
        setattr(args, "d%d" % k, dec[k - 1])


def is_image_file(name: str) -> bool:
    return any(name.endswith(e) for e in IMG_EXT)


def list_pairs(content_dir: str, style_dir: str, content_mark: str = ".", style_mark: str = ".") -> List[Tuple[str, str]]:
    """The Cartesian product of data_loader.py:32-36: contents outer, styles inner, os.listdir order, substring filters."""
    cs = [x for x in os.listdir(content_dir) if is_image_file(x) and content_mark in x]
    ss = [x for x in os.listdir(style_dir) if is_image_file(x) and style_mark in x]
    return [(c, s) for c in cs for s in ss]


def pair_name(content_file: str, style_file: str) -> str:
    """data_loader.py:60: '<content stem>+<style stem>.jpg' (stem = text before the FIRST dot)."""
    return content_file.split(".")[0] + "+" + style_file.split(".")[0] + ".jpg"


def out_name(args, imname: str) -> str:
    """WCT.py:127 (str(1) for the integer default of --alpha, like '%s' % args.alpha there)."""
    return os.path.join(args.outf, "%s_mode=%s_alpha=%s_%s" % (args.log_mark, args.mode, args.alpha, imname))


def load_rgb_u8(path: str, size: int = 0):
    """default_loader of data_loader.py:18-19: RGB uint8 HWC.  With `size`, also transforms.Resize(size) of :52-56 by Pillow on the
    host -- the CLI itself passes 0 and resizes on the GPU (WCT.resize_u8, bit-identical); this form is the host-side reference of
    tests/test_cli.py."""
    import numpy as np
    from PIL import Image
    img = Image.open(path).convert("RGB")
    if size:
        w, h = img.size
        if not ((w <= h and w == size) or (h <= w and h == size)):
            if w < h:
                ow, oh = size, int(size * h / w)
            else:
                oh, ow = size, int(size * w / h)
            img = img.resize((ow, oh), Image.BILINEAR)
    return np.array(img, dtype=np.uint8)   # a writable, contiguous copy


class LogPrinter:     # WCT.py:78-82
    def __init__(self, debug: bool, path: str):
        self.log = sys.stdout if debug else open(path, "a+")

    def __call__(self, sth):
        print(str(sth), file=self.log, flush=True)


def _to_tensor(wct, u8, size):
    """Resize (data_loader.py:52-56) + ToTensor (:57-58) on the GPU from the decoded uint8 frame."""
    return wct.resize_u8(u8, size, to_tensor=True) if size else wct.to_tensor_u8(u8)


def _fp32_fallback(wct, args, logprinter, c_f32, style_u8_dev):
    """An activation left the f16x3 range (+-65504) and was clamped: a deviation from the fp32 reference -- never silent.
    Recompute this pair with the exact-fp32 convolutions (style statistics included)."""
    logprinter("WARNING: f16x3 range exceeded for this pair -> recomputing it with exact-fp32 convolutions")
    wct.set_conv_mode("fp32")
    res = wct.stylize(c_f32, _to_tensor(wct, style_u8_dev, args.style_size), args.alpha, args.num_run)
    wct.sync()
    wct.set_conv_mode("f16x3")
    return res


def run_serial(args, wct, pairs, content_dir, style_dir, logprinter) -> float:
    """The reference's loop (WCT.py:112-131), one pair at a time; returns the summed per-pair time (cascade + save, like its timer)."""
    import torch
    from PIL import Image
    avg = 0.0
    # style statistics are computed once per style image and reused for every content it is paired with
    # (data_loader.py:32-36 builds the content x style product; the reference re-encodes the style for every pair)
    style_cache = {}
    for i, (cfile, sfile) in enumerate(pairs):
        imname = pair_name(cfile, sfile)
        logprinter("\n" + "*" * 30 + ' #%s: Transferring "%s"' % (i, imname))
        # decoded uint8 frames cross PCIe as they are (3 B/px, pinned); Resize and ToTensor run on the GPU
        c_u8 = torch.from_numpy(load_rgb_u8(os.path.join(content_dir, cfile))).pin_memory().cuda(non_blocking=True)
        s_u8 = None
        if sfile not in style_cache:
            s_u8 = torch.from_numpy(load_rgb_u8(os.path.join(style_dir, sfile))).pin_memory().cuda(non_blocking=True)
        t0 = time.time()
        if s_u8 is not None:
            wct.style_prepare(_to_tensor(wct, s_u8, args.style_size))
            style_cache[sfile] = {L: wct.style_export(L) for L in (5, 4, 3, 2, 1)}
        else:
            for L, stats in style_cache[sfile].items():
                wct.style_import(L, stats)
        c_f32 = _to_tensor(wct, c_u8, args.content_size)
        res = wct.stylize_prepared(c_f32, args.alpha, args.num_run)
        if wct.saturation_count(reset=True):
            if s_u8 is None:
                s_u8 = torch.from_numpy(load_rgb_u8(os.path.join(style_dir, sfile))).cuda()
            res = _fp32_fallback(wct, args, logprinter, c_f32, s_u8)
            style_cache.pop(sfile, None)          # its cached statistics may carry the clamp too
        out = wct.to_u8(res, args.round_mode).cpu().numpy()   # .cpu() syncs
        Image.fromarray(out).save(out_name(args, imname))
        dt = time.time() - t0
        avg += dt
        logprinter("Elapsed time is: %.4f seconds" % dt)
    return avg


def run_pipelined(args, wct, pairs, content_dir, style_dir, logprinter) -> float:
    """The same pairs, the same engine, the same GPU work in the same order -- with the host work taken off the GPU's critical path:
      decode pool   PIL decode + copy into pinned memory of the next pairs' files (both release the GIL), `--io_threads` wide
      main thread   H2D (async), style statistics (once per style) / import, cascade, uint8 conversion, D2H (async) + an event
      writer pool   waits for a pair's event, Image.save (JPEG encode releases the GIL)
    At most `--pipeline` pairs are between "enqueued on the GPU" and "handed to a writer".  The f16x3 range flag of pair i is read
    (counter in stream order, copied with the image) when the pair leaves that window; a clamped pair is recomputed in exact fp32
    and everything enqueued after it is discarded and redone (its style statistics may carry the clamp) -- the serial loop's
    results.  Returns the wall time of the whole run."""
    import collections
    import concurrent.futures as cf
    import torch
    from PIL import Image
    n = len(pairs)
    depth = max(1, int(args.pipeline))
    io = max(1, int(args.io_threads))
    ahead = depth + io                                # pairs whose files are decoded (or being decoded) ahead of the GPU
    dec = cf.ThreadPoolExecutor(io, thread_name_prefix="wct-decode")
    wr = cf.ThreadPoolExecutor(io, thread_name_prefix="wct-write")

    def decode(path):
        return torch.from_numpy(load_rgb_u8(path)).pin_memory()

    def save(rec):
        rec["ev"].synchronize()
        Image.fromarray(rec["host"].numpy()).save(rec["path"])
        return time.time()

    strict = wct.strict_range
    wct.strict_range = False                          # the flag is read per pair below, never raised in the middle of the window
    t_start = time.time()
    fut = {}                                          # ("c" | "s", file) -> decode future
    style_cache = {}                                  # sfile -> {level: statistics}
    style_dev = {}                                    # sfile -> uint8 device image while pairs using it may still need the fp32 fallback
    style_by = {}                                     # sfile -> index of the pair whose style_prepare produced the cached statistics
    inflight = collections.deque()
    writes = []
    last_flag = 0.0
    nxt = 0
    done_log = t_start

    def want(i):
        for j in range(i, min(n, i + ahead)):
            cfile, sfile = pairs[j]
            if ("c", cfile) not in fut:
                fut[("c", cfile)] = dec.submit(decode, os.path.join(content_dir, cfile))
            if sfile not in style_cache and ("s", sfile) not in fut:
                fut[("s", sfile)] = dec.submit(decode, os.path.join(style_dir, sfile))
        live = {("c", pairs[j][0]) for j in range(i, min(n, i + ahead))} | {("s", pairs[j][1]) for j in range(i, min(n, i + ahead))}
        for k in [k for k in fut if k not in live]:
            del fut[k]

    def retire(rec):
        """Pair leaves the window: its flag has been copied with its image."""
        nonlocal last_flag, nxt, done_log
        rec["ev"].synchronize()
        flag = float(rec["flag_host"][0])
        if flag > last_flag:
            # clamped: redo this pair in fp32, drop what was enqueued behind it (same statistics objects) and resume after it
            torch.cuda.synchronize()
            inflight.clear()
            wct.saturation_count(reset=True)          # acknowledged: the counter restarts at zero
            sfile = rec["sfile"]
            s_u8 = style_dev.get(sfile)
            if s_u8 is None:
                s_u8 = torch.from_numpy(load_rgb_u8(os.path.join(style_dir, sfile))).cuda()
            res = _fp32_fallback(wct, args, logprinter, rec["c_f32"], s_u8)
            rec["host"].copy_(wct.to_u8(res, args.round_mode))
            rec["ev"] = torch.cuda.Event()
            rec["ev"].record()
            # the reset above acknowledged EVERY clamp so far, also one inside a style_prepare of a discarded pair (index > i): statistics
            # prepared by this pair or any later one are dropped with the pairs, so the redone pairs prepare them again under a live flag
            # (only the clamped pair's own style used to be dropped)
            for f in [f for f, by in style_by.items() if by >= rec["i"]] + [sfile]:
                style_cache.pop(f, None)
                style_dev.pop(f, None)
                style_by.pop(f, None)
            last_flag = 0.0
            nxt = rec["i"] + 1
        else:
            last_flag = flag
        rec.pop("c_f32", None)
        rec.pop("keep", None)                         # the event has passed: the GPU is done with the pair's device buffers
        writes.append(wr.submit(save, rec))
        # bound the writer queue too: a slow disk must not let pinned result buffers pile up behind Image.save
        for w in [w for w in writes if w.done()]:
            w.result()                                # (re-raises a writer's exception here, not at the end of the folder)
            writes.remove(w)
        while len(writes) > 2 * io:
            writes.pop(0).result()
        now = time.time()
        logprinter('#%s "%s" left the pipeline, %.4f seconds after the previous pair' % (rec["i"], rec["imname"], now - done_log))
        done_log = now

    try:
        while nxt < n or inflight:
            while inflight and (len(inflight) >= depth or nxt >= n):
                retire(inflight.popleft())
            if nxt >= n:
                continue
            i = nxt
            nxt += 1
            want(i)
            cfile, sfile = pairs[i]
            imname = pair_name(cfile, sfile)
            logprinter("\n" + "*" * 30 + ' #%s: Transferring "%s"' % (i, imname))
            c_u8 = fut[("c", cfile)].result().cuda(non_blocking=True)
            if sfile not in style_cache:
                s_u8 = fut[("s", sfile)].result().cuda(non_blocking=True)
                style_dev[sfile] = s_u8
                while len(style_dev) > depth + 2:           # the fallback reloads a style it no longer finds here: keep the few that can be in flight
                    style_dev.pop(next(iter(style_dev)))
                wct.style_prepare(_to_tensor(wct, s_u8, args.style_size))
                style_cache[sfile] = {L: wct.style_export(L) for L in (5, 4, 3, 2, 1)}
                style_by[sfile] = i
            else:
                for L, stats in style_cache[sfile].items():
                    wct.style_import(L, stats)
            c_f32 = _to_tensor(wct, c_u8, args.content_size)
            res = wct.stylize_prepared(c_f32, args.alpha, args.num_run)
            out_dev = wct.to_u8(res, args.round_mode)
            host = torch.empty(out_dev.shape, dtype=torch.uint8, pin_memory=True)
            host.copy_(out_dev, non_blocking=True)
            flag_host = torch.zeros(1, dtype=torch.float64).pin_memory()
            flag_host.copy_(wct.range_flag(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            inflight.append({"i": i, "imname": imname, "path": out_name(args, imname), "host": host, "flag_host": flag_host, "ev": ev,
                             "sfile": sfile, "c_f32": c_f32, "keep": (out_dev, res, c_u8)})
        for w in writes:
            w.result()
    finally:
        wct.strict_range = strict
        dec.shutdown(wait=True)
        wr.shutdown(wait=True)
    return time.time() - t_start


def main(argv: Optional[List[str]] = None) -> int:
    args = build_parser().parse_args(argv)
    checkpoint_args(args)
    if args.synthesis:
        raise NotImplementedError("--synthesis is broken in the reference (data_loader.py:74) and not part of this path")
    os.makedirs(args.outf, exist_ok=True)
    logprinter = LogPrinter(args.debug, os.path.join(args.outf, "log_%s_%s.txt" % (args.log_mark, args.mode)))
    logprinter(sorted(vars(args).items()))
    content_dir = args.UHD_contentPath if args.UHD else args.contentPath
    style_dir = args.UHD_stylePath if args.UHD else args.stylePath
    pairs = list_pairs(content_dir, style_dir, args.picked_content_mark, args.picked_style_mark)

    from .wct import WCT      # raises ImportError if libwct_hip.so is missing: no CPU fallback
    wct = WCT(args)
    logprinter("Number of content-style pairs: %s" % len(pairs))
    if args.pipeline > 0:
        wall = run_pipelined(args, wct, pairs, content_dir, style_dir, logprinter)
        if pairs:
            logprinter("Processed %d images. Average processing time per pair is: %.4f seconds (pipelined: wall time / pairs, decode and "
                       "save included)" % (len(pairs), wall / len(pairs)))
    else:
        avg = run_serial(args, wct, pairs, content_dir, style_dir, logprinter)
        if pairs:
            logprinter("Processed %d images. Average processing time per pair is: %.4f seconds" % (len(pairs), avg / len(pairs)))
    return 0


if __name__ == "__main__":
    sys.exit(main())
