"""Measurement / checking stand-ins for torch.distributed on ONE device -- NOT product code (moved out of wct_hip/sharded.py in
round 5).  Used by bench.py's rank simulations (LoopbackGroup) and tests/test_sharded_gpu.py (InProcessWorld:
every rank of a column-strip job as a thread of one process, collectives on DEVICE buffers ordered by events only).  The product
path is wct_hip.sharded.ShardedStylizer over torch.distributed ("nccl" = RCCL)."""
from __future__ import annotations

import os
import sys

import torch

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "collaborative-distillation_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
from wct_hip.sharded import ShardedStylizer  # noqa: E402


class LoopbackGroup:
    """ONE rank of a `world`-rank job with its peers emulated on the same GPU -- a measurement stand-in for torch.distributed
    (bench.py `passes.cfg4_rank_sim`: what a rank of the 8-GPU config-4 job executes, timed on the one GPU there is).
    The rank's own work is exactly the sharded path's (same ShardedStylizer code, same C-ABI calls, same torch ops); the peers'
    contributions are replaced by data of the right shape already on the device:
      all_reduce   forwarded to `real` (a 1-rank RCCL communicator: the collective's kernel is launched, nothing crosses a link)
                   or a no-op; the moments stay those of the strip, so (M, b) are the strip's own -- same work, other numbers
      broadcast    levels this rank owns: forwarded / no-op; the others: a device copy of `style_stats[level]` (set by the caller
                   from a complete style_prepare) into the receive buffer
      send / recv  every received halo is a device copy of the equally wide block this rank sends the other way
    Results are NOT the sharded job's results (tests/test_sharded_*.py check those); timings are."""

    class _Done:
        def wait(self):
            return None

    emulates_peers = True      # ShardedStylizer refuses broadcast_map=True on such a group

    def __init__(self, rank: int, world: int, real=None):
        self.rank, self.world, self.real = rank, world, real
        self.style_stats = {}
        self._bcast = 0
        self._level = None      # set by ShardedStylizer before every level's collectives (set_level)

    def set_level(self, level: int):
        self._level = level

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.world

    def get_backend(self):
        return "nccl"          # device buffers, no host staging (sharded._p2p)

    def all_reduce(self, t, op=None):
        if self.real is not None:
            self.real.all_reduce(t)

    def broadcast(self, t, src=0):
        # the level comes from ShardedStylizer (set_level), not from counting calls: a skipped level or an extra broadcast per
        # level (broadcast_map) would shift a count and copy statistics of the wrong level and size
        level = self._level
        self._bcast += 1
        if src == self.rank:
            if self.real is not None:
                self.real.broadcast(t, src=0)
        else:
            stats = self.style_stats[level]
            if stats.numel() != t.numel():
                raise RuntimeError("LoopbackGroup: level %d statistics hold %d values, the receive buffer %d" % (level, stats.numel(), t.numel()))
            t.copy_(stats)

    def barrier(self):
        return None

    # point-to-point: P2POp(dist.isend | dist.irecv, tensor, peer) + batch_isend_irecv(ops)
    isend, irecv = "isend", "irecv"

    @staticmethod
    def P2POp(kind, tensor, peer):
        return (kind, tensor, peer)

    def batch_isend_irecv(self, ops):
        sends = [t for k, t, _ in ops if k == "isend"]
        for k, t, _ in ops:
            if k != "irecv":
                continue
            src = next((s for s in sends if s.shape == t.shape), None)
            if src is not None:
                t.copy_(src)
            elif sends:                      # a narrower last strip: whatever block there is, cropped
                t.copy_(sends[0][..., :t.shape[-1]])
        return [self._Done() for _ in ops]


def dist_transport(dist):
    """torch.distributed as the three callables of wct_hip.WCT.comm_attach_collectives: the library's cascade (wct_stylize_sharded) over a
    process group the library has no transport of its own for -- gloo between ranks that SHARE one GPU (bench.py / tests on a 1-GPU box; host-staged,
    synchronising) -- so that the C cascade's geometry and ordering are exercised by multi-process jobs there too.  With "nccl" the engine's own
    RCCL table (comm_init) is the product path; this adapter is test infrastructure."""
    stage = dist.get_backend() != "nccl"

    def guard(fn):
        def run(*a):
            try:
                fn(*a)
                return 0
            except BaseException as e:      # noqa: BLE001
                sys.stderr.write("dist_transport: %r\n" % (e,))
                return 1
        return run

    def all_reduce(ptr, count, stream):
        t = dev_tensor(ptr, count, torch.float64)
        if stage:
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        else:
            dist.all_reduce(t)

    def broadcast(ptr, nbytes, root, stream):
        t = dev_tensor(ptr, nbytes, torch.uint8)
        if stage:
            h = t.cpu()
            dist.broadcast(h, src=root)
            t.copy_(h)
        else:
            dist.broadcast(t, src=root)

    def sendrecv(ops, stream):
        reqs, staged = [], []
        for peer, is_send, ptr, nbytes in ops:
            t = dev_tensor(ptr, nbytes // 4, torch.float32)
            if is_send:
                reqs.append(dist.P2POp(dist.isend, t.cpu() if stage else t, peer))
            else:
                buf = torch.empty(t.shape, dtype=t.dtype) if stage else t
                staged.append((t, buf))
                reqs.append(dist.P2POp(dist.irecv, buf, peer))
        for r in dist.batch_isend_irecv(reqs):
            r.wait()
        if stage:
            for t, buf in staged:
                t.copy_(buf)
    return guard(all_reduce), guard(broadcast), guard(sendrecv)


class InProcessWorld:
    """ALL ranks of a `world`-rank job as threads of ONE process on ONE device, each with its own engine and its own stream --
    a stand-in for RCCL whose collectives move DEVICE buffers and are ordered by events only (no host staging, no stream or
    device synchronisation anywhere), so that what the job computes is checkable (tests/test_sharded_gpu.py): the same
    ShardedStylizer code path as under torch.distributed "nccl" -- `get_backend()` says "nccl", sharded._p2p takes its
    device-buffer branch -- with real asynchrony between the ranks' streams.  A missing stream dependency between the library's
    lanes and the caller's stream (e.g. style_export -> broadcast -> style_import, moments -> all_reduce, decoded columns ->
    send) shows up as a wrong picture here exactly as it would over xGMI; under gloo it cannot (host staging synchronises).

    Semantics, per collective (every rank's thread calls it, like NCCL):
      all_reduce(t)        SUM in rank order 0..world-1 on every rank (identical bits everywhere), in place
      broadcast(t, src)    src's buffer copied device-to-device into every other rank's t
      batch_isend_irecv    every rank posts its sends and receives together; a receive is a device copy of the matching send
    Host threads meet at a barrier only to hand each other event handles; the data movement itself is enqueued on the ranks'
    streams behind those events.  `group(rank)` is the per-rank object with torch.distributed's call surface."""

    def __init__(self, world: int, sync_every: bool = False):
        import threading
        self.world = world
        self.sync_every = sync_every           # checking aid: a device-wide synchronisation around every collective (the "gloo-like" order)
        self._bar = threading.Barrier(world)
        self._slot = [None] * world            # per rank: what it deposited for the collective in progress
        self._mail = {}                        # (src, dst) -> [(tensor, event)] in posting order
        self._done = [None] * world

    def group(self, rank: int) -> "InProcessGroup":
        return InProcessGroup(self, rank)


class _DevPtr:
    """A raw device pointer as something torch.as_tensor() wraps without copying (__cuda_array_interface__)."""

    def __init__(self, ptr: int, count: int, typestr: str):
        self.__cuda_array_interface__ = {"shape": (int(count),), "typestr": typestr, "data": (int(ptr), False), "version": 2}


def dev_tensor(ptr: int, count: int, dtype) -> torch.Tensor:
    return torch.as_tensor(_DevPtr(ptr, count, {torch.float64: "<f8", torch.float32: "<f4", torch.uint8: "|u1"}[dtype]), device="cuda")


class InProcessGroup:
    isend, irecv = "isend", "irecv"

    class _Done:
        def wait(self):
            return None

    def __init__(self, w: InProcessWorld, rank: int):
        self.w, self.rank = w, rank
        self.calls = {"all_reduce": 0, "broadcast": 0, "p2p": 0}

    def get_rank(self):
        return self.rank

    def get_world_size(self):
        return self.w.world

    def get_backend(self):
        return "nccl"

    def barrier(self):
        self.w._bar.wait()

    def c_transport(self):
        """This group as the three callables of wct_hip.WCT.comm_attach_collectives (include/wct_hip.h wct_collectives): the library's own
        cascade (wct_stylize_sharded) then talks to its peers -- the other rank threads -- through exactly the collectives above.  The
        callbacks run on the rank's thread inside the library call, whose stream is the thread's current stream."""
        def guard(fn):
            def run(*a):
                try:
                    fn(*a)
                    return 0
                except BaseException as e:      # noqa: BLE001  an exception must not unwind through the C frames
                    self.error = e
                    try:
                        self.w._bar.abort()
                    except Exception:           # noqa: BLE001
                        pass
                    return 1
            return run

        def all_reduce(ptr, count, stream):
            self.all_reduce(dev_tensor(ptr, count, torch.float64))

        def broadcast(ptr, nbytes, root, stream):
            self.broadcast(dev_tensor(ptr, nbytes, torch.uint8), src=root)

        def sendrecv(ops, stream):
            self.batch_isend_irecv([(self.isend if is_send else self.irecv, dev_tensor(ptr, nbytes // 4, torch.float32), peer)
                                    for peer, is_send, ptr, nbytes in ops])
        return guard(all_reduce), guard(broadcast), guard(sendrecv)

    def _event(self):
        if self.w.sync_every:
            torch.cuda.synchronize()
        ev = torch.cuda.Event()
        ev.record()                             # on the calling thread's current stream
        return ev

    def all_reduce(self, t, op=None):
        w, r = self.w, self.rank
        self.calls["all_reduce"] += 1
        w._slot[r] = (t, self._event())
        w._bar.wait()
        cur = torch.cuda.current_stream()
        for q in range(w.world):
            if q != r:
                cur.wait_event(w._slot[q][1])
        acc = w._slot[0][0].clone()
        for q in range(1, w.world):
            acc += w._slot[q][0]                # rank order: the same bits on every rank
        w._done[r] = self._event()              # this rank has finished READING its peers' buffers
        w._bar.wait()
        for q in range(w.world):
            if q != r:
                cur.wait_event(w._done[q])      # nobody still reads t
        t.copy_(acc)
        w._bar.wait()                           # slots free for the next collective

    def broadcast(self, t, src=0):
        w, r = self.w, self.rank
        self.calls["broadcast"] += 1
        if r == src:
            w._slot[src] = (t, self._event())
        w._bar.wait()
        cur = torch.cuda.current_stream()
        if r != src:
            cur.wait_event(w._slot[src][1])
            t.copy_(w._slot[src][0])
        w._done[r] = self._event()
        w._bar.wait()
        if r == src:
            for q in range(w.world):
                if q != r:
                    cur.wait_event(w._done[q])  # the source may overwrite its buffer only after every peer has copied it
        w._bar.wait()

    @staticmethod
    def P2POp(kind, tensor, peer):
        return (kind, tensor, peer)

    def batch_isend_irecv(self, ops):
        w, r = self.w, self.rank
        self.calls["p2p"] += 1
        for kind, t, peer in ops:
            if kind == "isend":
                w._mail.setdefault((r, peer), []).append((t, self._event()))
        w._bar.wait()
        cur = torch.cuda.current_stream()
        taken = {}
        for kind, t, peer in ops:
            if kind == "irecv":
                k = taken.get(peer, 0)
                src, ev = w._mail[(peer, r)][k]
                taken[peer] = k + 1
                if tuple(src.shape) != tuple(t.shape):
                    raise RuntimeError("rank %d: receive %s from rank %d does not match its send %s" % (r, tuple(t.shape), peer, tuple(src.shape)))
                cur.wait_event(ev)
                t.copy_(src)
        w._done[r] = self._event()
        w._bar.wait()
        for q in range(w.world):
            if q != r:
                cur.wait_event(w._done[q])      # the sent blocks may be reused only after the peers have copied them
        w._bar.wait()
        if r == 0:
            w._mail.clear()
        w._bar.wait()
        return [self._Done() for _ in ops]


def run_in_process(world: int, make_engine, content: torch.Tensor, style: torch.Tensor, sync_every: bool = False, **kw):
    """Stylise `content` [3, H, W] as a `world`-rank column-strip job inside this process (InProcessWorld): returns (the
    assembled image [1, 3, H', W'], the per-rank groups).  `make_engine()` is called once per rank (each rank owns a context);
    kw goes to ShardedStylizer (halo_mode, broadcast_map, alpha, style_mode).  c_cascade=True: every rank's frame is ONE library call
    (wct_stylize_sharded) whose transport table is this world's collectives (InProcessGroup.c_transport)."""
    import threading
    H, W = int(content.shape[-2]), int(content.shape[-1])
    wd = InProcessWorld(world, sync_every)
    engines = [make_engine() for _ in range(world)]
    groups = [wd.group(r) for r in range(world)]
    streams = [torch.cuda.Stream() for _ in range(world)]
    outs, errs = [None] * world, [None] * world
    dev = torch.cuda.current_device()
    ready = torch.cuda.Event()
    ready.record()

    def work(r):
        try:
            torch.cuda.set_device(dev)
            with torch.cuda.stream(streams[r]):
                streams[r].wait_event(ready)
                if kw.get("c_cascade"):
                    engines[r].comm_attach_collectives(*groups[r].c_transport(), world, r)
                sh = ShardedStylizer(engines[r], groups[r], H, W, int(style.shape[-2]), int(style.shape[-1]), **kw)
                x0, x1 = sh.input_columns()
                outs[r] = (sh.own, sh.stylize_strip(content[:, :, x0:x1].contiguous(), style))
        except BaseException as e:      # noqa: BLE001  a dead rank must not leave its peers in a barrier
            errs[r] = e
            wd._bar.abort()

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    errs = [getattr(g, "error", None) for g in groups] + errs      # what a transport callback swallowed inside the C frames comes first
    first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in errs if e is not None), None)
    if first is not None:
        raise first
    torch.cuda.synchronize()
    for e in engines:
        e.sync()
    full = torch.cat([o[1] for o in sorted(outs, key=lambda o: o[0][0])], dim=3)
    return full, groups
