"""Content-parallel (column-strip) sharding of ONE stylisation across the GPUs of a node.

The reference is single-GPU; "ultra-resolution" there means pruning + CPU offload.  What makes the
path shardable is that every operator is local except the content statistics: per level the only global
coupling is (n, SUM x, SUM x x^T) of the content feature map (util_wct.py:68-70).  So:

  * the content image is cut into `world` column strips whose origins are multiples of 16, so that all four
    2x2 pooling grids coincide with the untiled image's;
  * rank r works on its strip plus a halo; a level's encode->decode receptive field is (160, 72, 24, 10, 2) columns per
    side at level 5..1 (LEVEL_HALO), reflect padding happens only at the true image borders, and with those
    margins a strip's owned columns are bit-identical to the untiled level given the same (M, b).  Two ways to feed the
    next level its halo (`halo_mode`):
      "recompute"  halos are CUMULATIVE over the cascade: level L gets A_L = (272, 112, 40, 16, 6)[5-L] extra columns per
                   interior side, enough that after the level's own receptive field the still-exact region covers what
                   level L-1 needs.  NO neighbour exchange between levels; the extra columns cost FLOPs:
                   2 A_L / strip width per level = +6 % of a frame at 3840-wide strips, +26 % at the 1280-wide strips of
                   config 4 on 8 GPUs (42 % at level 5, which is 46 % of the FLOPs);
      "exchange"   every level gets exactly its own margin (+15.6 % at 1280-wide strips) and, between levels, each rank
                   sends its neighbours the (72, 24, 10, 2) outermost OWNED columns of the image it just decoded (exact
                   there) -- two point-to-point messages per level boundary, 3 x H x halo fp32 (<= 3.5 MB at H = 4096) over
                   xGMI -- SURVEY 8e's design;
      "auto"       exchange for strips narrower than 2560 columns (config 4 on 8 GPUs), recompute otherwise;
  * moments are accumulated over OWNED columns only and summed with one all-reduce (fp64, C*C + C values:
    132 KB at C = 128, latency-bound on xGMI) per level;
  * the style side (five encodes + moments + matrix square roots, a third of a single-GPU step) depends only on the
    style image (`style_mode`):
      "strips"     the style image is cut into column strips exactly like the content: every rank encodes ITS strip + the
                   ENCODER's receptive field (STYLE_HALO = 80 / 32 / 12 / 4 / 1 columns per interior side at level 5..1) at every
                   level and sums raw moments over its owned feature columns -- the same statistic as the content's, summed the same
                   way: level L's style sums ride in level L's all-reduce of the content moments (no extra collective) and every
                   rank takes the level's matrix square root itself right behind it.  Per-rank style work is 1 / world of the style side (+ margins:
                   1.3x at 256-column strips) on EVERY rank;
      "owner"      level L's style statistics are computed whole by rank (5 - L) mod world on its side stream and broadcast
                   (C*C + C fp64 values per level, 132 KB at C = 128); rank 0 carries level 5 = 45.6 % of the style FLOPs;
      "replicate"  every rank repeats all five levels (no communication; styles too narrow to cut);
      "auto"       = "owner": measured at 2, 4 and 8 ranks (profiles/r06_style_arrangement_per_level_join.txt) it beats "strips" by 0.5-1.3 ms per
                   rank frame -- every style-side matrix square root occupies an XCD that the content lane's persistent kernels then wait
                   for, and strips / replicate run five of them on EVERY rank where owner runs at most three on one;
  * the colouring map (M, b): every rank now holds the same global content moments (the all-reduce returns identical bits
    everywhere) and the same style statistics, and the solver is deterministic, so every rank solves for itself and folds
    the SAME matrices into its decoder -- two collectives per level.  `broadcast_map=True` keeps the other arrangement
    (rank 0 solves and broadcasts (M, b), <= 2.1 MB at C = 512; style statistics only travel to rank 0): three collectives
    per level, one solve per node.

The orchestration is backend-agnostic: `engine` is a wct_hip.WCT on the GPU (RCCL = torch.distributed "nccl"),
and tests run the same code under gloo with a CPU checker as the engine.

`c_cascade=True` hands the WHOLE of the above to the library: ONE call per frame (include/wct_hip.h wct_stylize_sharded --
geometry, crops, style side, collectives on the engine's own RCCL communicator, neighbour exchange, all on the context's streams).
The Python orchestration in this file then is the CHECKER of that path: same geometry, same arithmetic, bit-identical results
(tests/test_sharded_gpu.py).
"""
from __future__ import annotations

import time
from typing import List, Optional, Tuple

import torch

#: composite encode->decode receptive-field margin per side, in image columns of that level
LEVEL_HALO = {5: 160, 4: 72, 3: 24, 2: 10, 1: 2}
#: the ENCODER's receptive field alone (style strips: 70 / 30 / 10 / 4 / 1 columns, rounded up to a multiple of 2^(L-1))
STYLE_HALO = {5: 80, 4: 32, 3: 12, 2: 4, 1: 1}
#: cumulative halo needed at the INPUT of level L (multiples of 2^(L-1); A_L - LEVEL_HALO[L] >= A_{L-1})
CUM_HALO = {5: 272, 4: 112, 3: 40, 2: 16, 1: 6}
#: strips narrower than this take the neighbour exchange under halo_mode="auto"
AUTO_EXCHANGE_BELOW = 2560
#: share of a frame's convolution FLOPs per level 5..1 (45 792 / 30 816 / 14 688 / 7 776 / 1 296 per pixel)
LEVEL_FLOP_SHARE = {5: 0.456, 4: 0.307, 3: 0.146, 2: 0.077, 1: 0.013}


def halo_flop_overhead(strip_width: int, mode: str) -> float:
    """Extra convolution FLOPs of an interior strip relative to its owned columns (both sides carry a halo)."""
    halo = CUM_HALO if mode == "recompute" else LEVEL_HALO
    return sum(LEVEL_FLOP_SHARE[L] * 2.0 * halo[L] / strip_width for L in (5, 4, 3, 2, 1))


def strip_bounds(W: int, world: int) -> List[Tuple[int, int]]:
    """Owned column range of every rank: origins are multiples of 16, the last strip takes the remainder."""
    xs = [min(W, (r * W // world) // 16 * 16) for r in range(world)] + [W]
    for r in range(world):
        if xs[r + 1] - xs[r] < 16 and world > 1:
            raise ValueError("image width %d too small for %d strips" % (W, world))
    return [(xs[r], xs[r + 1]) for r in range(world)]


def ext_bounds(own: Tuple[int, int], W: int, halo: int) -> Tuple[int, int]:
    """The strip extended by `halo` columns towards the image interior (never beyond the image)."""
    return max(0, own[0] - halo), min(W, own[1] + halo)


class ShardedStylizer:
    def __init__(self, engine, dist, H: int, W_total: int, Hs: int, Ws: int, rank: Optional[int] = None,
                 world: Optional[int] = None, alpha: float = 1.0, broadcast_map: bool = False, halo_mode: str = "auto",
                 c_collectives: Optional[bool] = None, style_mode: str = "auto", c_cascade: bool = False, fast_fold: bool = False):
        self.e, self.dist = engine, dist
        self.broadcast_map = broadcast_map
        # c_collectives: a level's encode -> all-reduce -> solve -> decode chain as ONE library call on the engine's own RCCL communicator
        # (engine.comm_init(dist); include/wct_hip.h wct_level_sharded) instead of three calls + torch.distributed.all_reduce + tensor glue.
        # None = whenever the engine has a communicator (and every rank solves for itself: not with broadcast_map).  Bit-identical.
        # c_cascade: the WHOLE frame as one library call (wct_stylize_sharded); opt-in, needs the engine's communicator as well.
        has = bool(getattr(engine, "has_comm", False))
        self.c_cascade = bool(c_cascade)
        # fast_fold (c_cascade only): the single-GPU cascade's fold without (M, b) on the critical path (WCT_SHARD_FAST_FOLD); fp32 round-off from the
        # default form, so no longer bit-identical to this file's split-level orchestration
        self.fast_fold = bool(fast_fold)
        if self.fast_fold and not self.c_cascade:
            raise ValueError("fast_fold is an option of the library's cascade (c_cascade=True)")
        self.c_collectives = (has and not broadcast_map and not self.c_cascade) if c_collectives is None else bool(c_collectives)
        if self.c_collectives and (not has or broadcast_map):
            raise ValueError("c_collectives needs engine.comm_init(dist) and broadcast_map=False")
        if self.c_cascade and not has:
            raise ValueError("c_cascade needs a communicator or transport on the engine (engine.comm_init(dist) / comm_attach_collectives)")
        self.t_range_wait = 0.0     # seconds stylize_strip() has spent WAITING for an old frame's range flag (not enqueueing): bench.py splits on it
        self.rank = dist.get_rank() if rank is None else rank
        self.world = dist.get_world_size() if world is None else world
        if (self.c_collectives or self.c_cascade) and not getattr(dist, "emulates_peers", False):
            # the library all-reduces over ITS communicator while n_total and the strip geometry come from (world, rank): a communicator
            # over another group would hand every rank part of the moments under the full n_total -- silently wrong pictures
            info = tuple(engine.comm_info())
            if info != (self.world, self.rank):
                raise ValueError("the engine's communicator is rank %d of %d, this job is rank %d of %d: the library's collectives would "
                                 "run over another group than the strip geometry assumes" % (info[1], info[0], self.rank, self.world))
        if style_mode not in ("auto", "owner", "strips", "replicate"):
            raise ValueError("style_mode must be auto, owner, strips or replicate")
        if self.world == 1:
            style_mode = "replicate"
        elif style_mode == "auto":
            style_mode = "owner"
        self.style_mode = style_mode
        self.style_bounds = strip_bounds(Ws, self.world) if style_mode == "strips" else None
        self.H, self.W, self.Hs, self.Ws = H, W_total, Hs, Ws
        self.alpha = alpha
        self.bounds = strip_bounds(W_total, self.world)
        self.own = self.bounds[self.rank]
        if halo_mode not in ("auto", "recompute", "exchange"):
            raise ValueError("halo_mode must be auto, recompute or exchange")
        narrowest = min(b[1] - b[0] for b in self.bounds)
        if halo_mode == "auto":
            halo_mode = "exchange" if (self.world > 1 and narrowest < AUTO_EXCHANGE_BELOW and narrowest >= 2 * LEVEL_HALO[4]) else "recompute"
        if halo_mode == "exchange" and self.world > 1 and narrowest < 2 * LEVEL_HALO[4]:
            # a neighbour must own the columns it is asked for (72 at most), also after the last strip's floor-pooling shrink
            raise ValueError("halo_mode='exchange' needs strips of at least %d columns (narrowest: %d)" % (2 * LEVEL_HALO[4], narrowest))
        self.halo_mode = halo_mode
        self.halo = LEVEL_HALO if halo_mode == "exchange" else CUM_HALO
        if broadcast_map and getattr(dist, "emulates_peers", False):
            raise ValueError("a peer-emulating measurement group (tools/sharded_standins.LoopbackGroup) emulates the style-statistics "
                             "broadcasts only: broadcast_map=True is not supported by it")
        if self.c_cascade:
            g = engine.shard_geometry(W_total, self.world, self.rank, self.halo_mode)    # the library's geometry must be this file's
            if (g[0], g[1]) != self.own or (g[2], g[3]) != self.input_columns() or g[4] != self.halo_mode:
                raise RuntimeError("wct_shard_geometry %r disagrees with sharded.py (%r, %r, %s)" % (g, self.own, self.input_columns(), self.halo_mode))
        self._range = []            # [(pinned host value, event)], oldest first: node-wide f16x3 clamp totals of past stylize_strip calls

    def input_columns(self) -> Tuple[int, int]:
        """Columns of the full content image this rank must be given (its strip + the level-5 halo of its halo mode)."""
        return ext_bounds(self.own, self.W, self.halo[5])

    # ---- f16x3 range flag, node-wide
    #: frames a node-wide range flag may stay unread before stylize_strip() waits for it.  The read-back of frame k is recorded
    #: behind frame k's last all-reduce; frame k + 1 reads frame k - 1's, which has long landed: no stall in a pipelined loop
    RANGE_LAG = 1

    def check_range(self, wait: bool = True) -> None:
        """Raise OverflowError on EVERY rank if any rank's f16x3 kernels clamped an activation during a past stylize_strip (each
        rank's saturation counter rides in the all-reduce of the moments, so all ranks hold the same total and raise -- or fall
        back to set_conv_mode('fp32') -- together).  wait=True: every frame issued so far (synchronises with the last one).
        wait=False: only frames older than RANGE_LAG -- a FIXED lag, never "whatever has landed": which frame's flag is read
        must not depend on a rank's timing, or one rank would raise and skip a frame's collectives while its peers enter them
        (ADVICE r3).  The values are identical on all ranks (all-reduced), the point of reading is identical by construction."""
        keep = 0 if wait else self.RANGE_LAG
        total = 0.0
        while len(self._range) > keep:
            host, ev = self._range.pop(0)
            t0 = time.perf_counter()
            ev.synchronize()
            self.t_range_wait += time.perf_counter() - t0
            total = max(total, float(host[0]))
        if total > 0:
            self._range.clear()      # the counter is cumulative: later frames would only repeat the report
            raise OverflowError("wct_hip.sharded: %d activation(s) were clamped to the f16x3 range on some rank(s) of this job: the "
                                "frame deviates from the fp32 reference; every rank should switch to set_conv_mode('fp32') and "
                                "acknowledge with saturation_count(reset=True)" % int(total))

    # ---- neighbour exchange (halo_mode "exchange")
    def _p2p(self, sends, recvs):
        """sends: [(tensor, peer)], recvs: [(tensor, peer)] -- posted together, completed before returning.  NCCL (= RCCL):
        one batched group on device buffers.  gloo has no device-side send/recv: staged through host memory."""
        dist = self.dist
        stage = dist.get_backend() != "nccl"
        ops, staged = [], []
        for t, peer in sends:
            ops.append(dist.P2POp(dist.isend, t.cpu().contiguous() if stage else t.contiguous(), peer))
        for t, peer in recvs:
            buf = torch.empty(t.shape, dtype=t.dtype, device="cpu") if stage else t
            staged.append((t, buf))
            ops.append(dist.P2POp(dist.irecv, buf, peer))
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        if stage:
            for t, buf in staged:
                t.copy_(buf)

    def _exchange(self, out: torch.Tensor, lo: int, own: Tuple[int, int], W_cur: int, halo: int) -> Tuple[torch.Tensor, int]:
        """`out` holds image columns [lo, lo + out.shape[-1]) of the level just decoded, exact on `own`.  Returns the next
        level's input: `own` extended by `halo` columns per interior side, the extensions received from the neighbours
        (their outermost owned columns), and its first column."""
        rank, world = self.rank, self.world
        owned = out[..., own[0] - lo:own[1] - lo]
        nb = [(b[0], min(b[1], W_cur)) for b in self.bounds]          # every rank's owned range at this level
        left_w = min(halo, nb[rank - 1][1] - nb[rank - 1][0]) if rank > 0 else 0
        right_w = min(halo, nb[rank + 1][1] - nb[rank + 1][0]) if rank + 1 < world else 0
        my_w = own[1] - own[0]
        sends, recvs = [], []
        if rank > 0:
            sends.append((owned[..., :min(halo, my_w)], rank - 1))
        if rank + 1 < world:
            sends.append((owned[..., my_w - min(halo, my_w):], rank + 1))
        shape = list(owned.shape)
        left = torch.empty(shape[:-1] + [left_w], dtype=owned.dtype, device=owned.device)
        right = torch.empty(shape[:-1] + [right_w], dtype=owned.dtype, device=owned.device)
        if left_w:
            recvs.append((left, rank - 1))
        if right_w:
            recvs.append((right, rank + 1))
        self._p2p(sends, recvs)
        return torch.cat([left, owned, right], dim=-1).contiguous(), own[0] - left_w

    @torch.no_grad()
    def stylize_strip(self, content_ext: torch.Tensor, style: torch.Tensor) -> torch.Tensor:
        """content_ext: [3, H, x1-x0] columns `input_columns()` of the content; style: [3, Hs, Ws].
        Returns this rank's owned columns of the stylised image, [1, 3, H', own_w'] (H' = 16*floor(H/16)).

        Engine interface (wct_hip.WCT on the GPU; tests supply a CPU checker with the same four methods):
          style_prepare(style, levels)               style side of the given levels (GPU: side stream, overlaps the content)
          style_stats_count(L) / style_export(L) / style_import(L, stats)     the level's style statistics as one fp64 vector
          content_encode(L, img, f0, f1) -> h, w, sum, sumsq    encoder + raw moments over owned feature columns
          content_solve(L, n, sum, sumsq, alpha) -> M, b
          content_decode(L, M, b, H, W) -> image     decoder with (M, b) folded into its first conv
        """
        e, dist = self.e, self.dist
        self.check_range(wait=False)     # the node-wide flag of the frame before the previous one: same decision on every rank
        range_flag = getattr(e, "range_flag", None)
        # the engine's own per-call range check must not fire on ONE rank in the middle of a frame -- its peers would wait for it in
        # the next collective for ever; the flag travels in the all-reduce instead and check_range() raises on every rank
        strict = getattr(e, "strict_range", None)
        if strict is not None:
            e.strict_range = False
        try:
            if self.c_cascade:
                return self._stylize_strip_c(content_ext, style, range_flag)
            return self._stylize_strip(content_ext, style, range_flag)
        finally:
            if strict is not None:
                e.strict_range = strict

    def _note_range(self, flag):
        """Queue the frame's node-wide clamp total (device, 1 double) for check_range(): an asynchronous copy to pinned host memory + an event."""
        host = torch.zeros(1, dtype=torch.float64).pin_memory()
        host.copy_(flag, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._range.append((host, ev))

    def _stylize_strip_c(self, content_ext, style, range_flag):
        """The frame as ONE library call (wct_stylize_sharded)."""
        x0, x1 = self.input_columns()
        flag = torch.empty(1, dtype=torch.float64, device=content_ext.device) if range_flag is not None else None
        out = self.e.stylize_sharded(content_ext, style, self.W, x0, x1, alpha=self.alpha, halo_mode=self.halo_mode, style_mode=self.style_mode,
                                     broadcast_map=self.broadcast_map, range_total=flag, fast_fold=self.fast_fold)
        if flag is not None:
            self._note_range(flag)
        return out

    def _style_strip_moments(self, style):
        """style_mode "strips": this rank's raw style moments of every level as flat fp64 parts [sum_5, sumsq_5, sum_4, ...]."""
        e = self.e
        s0, s1 = self.style_bounds[self.rank]
        img = style if style.dim() == 3 else style[0]
        parts = []
        for L in (5, 4, 3, 2, 1):
            sh = L - 1
            lo, hi = ext_bounds((s0, s1), self.Ws, STYLE_HALO[L])
            f0 = (s0 - lo) >> sh
            f1 = -1 if s1 >= self.Ws else (s1 - lo) >> sh
            s, ss = e.style_moments(L, img[..., lo:hi].contiguous(), f0, f1)
            parts += [s.reshape(-1), ss.reshape(-1)]
        return parts

    def _style_strip_solve(self, flat, levels=(5, 4, 3, 2, 1)):
        """`flat`: the all-reduced concatenation of _style_strip_moments' parts of `levels` -> those levels' style statistics inside the engine."""
        o = 0
        for L in levels:
            sh = L - 1
            C = self._style_C[L]
            self.e.style_solve(L, float((self.Hs >> sh) * (self.Ws >> sh)), flat[o:o + C], flat[o + C:o + C + C * C].reshape(C, C))
            o += C + C * C

    def _stylize_strip(self, content_ext, style, range_flag):
        e, dist = self.e, self.dist
        flags = []
        img = content_ext if content_ext.dim() == 4 else content_ext[None]
        W_cur = self.W                       # width of the (virtual) full image at the current level
        own = self.own
        halo, exchange = self.halo, self.halo_mode == "exchange" and self.world > 1
        lo, hi = ext_bounds(own, W_cur, halo[5])
        assert img.shape[-1] == hi - lo, "expected columns [%d,%d) of the content" % (lo, hi)
        world, rank = self.world, self.rank
        owner = lambda lvl: (5 - lvl) % world           # rank 0 (the solver) owns level 5, the first one it needs
        strips, owner_mode = self.style_mode == "strips", self.style_mode == "owner"
        style_parts = None
        if strips:
            style_parts = self._style_strip_moments(style)
            self._style_C = {L: int(style_parts[2 * i].numel()) for i, L in enumerate((5, 4, 3, 2, 1))}
            if self.c_collectives:
                flat = torch.cat(style_parts)                          # the per-level library call carries the content moments only
                dist.all_reduce(flat)
                self._style_strip_solve(flat)
        elif owner_mode:
            e.style_prepare(style, levels=[lvl for lvl in (5, 4, 3, 2, 1) if owner(lvl) == rank])
        else:
            e.style_prepare(style, levels=[5, 4, 3, 2, 1])
        for L in (5, 4, 3, 2, 1):
            sh = L - 1
            if hasattr(dist, "set_level"):
                dist.set_level(L)             # measurement stand-ins (LoopbackGroup) key their emulated peers' data on the level
            # crop the running image to this level's extended strip
            nlo, nhi = ext_bounds(own, W_cur, halo[L])
            assert lo <= nlo and nhi <= hi, (L, lo, hi, nlo, nhi)
            img = img[..., nlo - lo:nhi - lo].contiguous()
            lo, hi = nlo, nhi
            H_in, W_in = int(img.shape[-2]), int(img.shape[-1])
            f0 = (own[0] - lo) >> sh                                   # owned feature columns
            f1 = -1 if own[1] >= W_cur else (own[1] - lo) >> sh        # last strip: to the (floored) end
            if self.c_collectives:
                # the whole level in one call; its style statistics were broadcast one level AHEAD (level 5's up front): the export makes the
                # caller's stream wait for the owner's style lane, so issued right in front of its own level it would order that level's
                # content encoder behind the style side on the owner rank and give the overlap away
                def share_stats(lvl):
                    if self.world > 1 and owner_mode:
                        if rank == owner(lvl):
                            stats = e.style_export(lvl)
                        else:
                            stats = torch.empty(e.style_stats_count(lvl), dtype=torch.float64, device=img.device)
                        dist.broadcast(stats, src=owner(lvl))
                        if rank != owner(lvl):
                            e.style_import(lvl, stats)
                if L == 5:
                    share_stats(5)
                h = H_in >> sh
                flag = torch.empty(1, dtype=torch.float64, device=img.device) if range_flag is not None else None
                img = e.level_sharded(L, img, f0, f1, float(h * (W_cur >> sh)), self.alpha, flag)
                if flag is not None:
                    flags.append(flag)
                if L > 1:
                    if hasattr(dist, "set_level"):
                        dist.set_level(L - 1)
                    share_stats(L - 1)
                W_cur = (W_cur >> sh) << sh
                hi = lo + int(img.shape[-1])
                own = (own[0], min(own[1], W_cur))
                if exchange and L > 1:
                    img, lo = self._exchange(img, lo, own, W_cur, halo[L - 1])
                    hi = lo + int(img.shape[-1])
                continue
            h, w_ext, sum_c, sumsq_c = e.content_encode(L, img, f0, f1)
            C = int(sum_c.numel())
            parts = [sum_c.reshape(-1), sumsq_c.reshape(-1)]
            if range_flag is not None:
                parts.append(range_flag())                             # this rank's f16x3 clamp counter so far: summed over the ranks below
            n_own = sum(int(p.numel()) for p in parts)
            if strips:
                i = 2 * (5 - L)
                parts += style_parts[i:i + 2]                          # this level's style sums ride in the same all-reduce
            packed = torch.cat(parts)
            if self.world > 1:
                dist.all_reduce(packed)                                # SUM, fp64, C*C + C (+ 1) values (+ the level's style sums)
            if range_flag is not None:
                flags.append(packed[C + C * C:C + C * C + 1])
            if strips:
                self._style_strip_solve(packed[n_own:], levels=(L,))
            solvers = (0,) if self.broadcast_map else range(self.world)   # ranks that need the level's style statistics
            if self.world > 1 and owner_mode and any(r != owner(L) for r in solvers):
                if rank == owner(L):
                    stats = e.style_export(L)
                else:
                    stats = torch.empty(e.style_stats_count(L), dtype=torch.float64, device=packed.device)
                dist.broadcast(stats, src=owner(L))
                if rank != owner(L) and rank in solvers:
                    e.style_import(L, stats)
            n_c = float(h * (W_cur >> sh))                             # feature pixels of the whole image
            if self.broadcast_map and self.world > 1:
                Mb = torch.empty(C * C + C, dtype=torch.float64, device=packed.device)
                if rank == 0:
                    M, b = e.content_solve(L, n_c, packed[:C], packed[C:C + C * C].reshape(C, C), self.alpha)
                    Mb[:C * C] = M.reshape(-1)
                    Mb[C * C:] = b
                dist.broadcast(Mb, src=0)                              # the colouring map, identical on every rank
                M, b = Mb[:C * C].reshape(C, C), Mb[C * C:]
            else:
                M, b = e.content_solve(L, n_c, packed[:C], packed[C:C + C * C].reshape(C, C), self.alpha)
            img = e.content_decode(L, M, b, H_in, W_in)               # [1,3,h<<sh, w_ext<<sh]
            # floor-mode pooling may have dropped trailing columns/rows of the full image
            W_cur = (W_cur >> sh) << sh
            hi = lo + int(img.shape[-1])
            own = (own[0], min(own[1], W_cur))
            if exchange and L > 1:
                # the decoded strip is exact on the owned columns only: the next level's margin comes from the neighbours
                img, lo = self._exchange(img, lo, own, W_cur, halo[L - 1])
                hi = lo + int(img.shape[-1])
        if flags:
            # levels 5..2 are covered by later all-reduces (the counter is cumulative); level 1's decode by the next frame's
            self._note_range(flags[-1])
        return img[..., own[0] - lo:own[1] - lo].contiguous()
