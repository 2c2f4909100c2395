"""Several content frames in flight on ONE GPU against one style (the per-GPU half of BASELINE config 5, and the video case).

A single frame's critical path has stretches where the chip is nearly idle: the matrix functions of a level are one-workgroup
or 16-workgroup kernels (0.15-0.3 ms per level at C <= 128), and the level cannot go on before them.  Another frame's
convolutions fit into those gaps, so `slots` engines -- each with its own context and stream, all holding the SAME style
statistics (computed once, copied with style_export / style_import) -- are fed round robin.  Measured on the 4K bench
content against cached style statistics: 8.10 ms per frame with one frame in flight, 7.57 with two, 7.28 with three
(tools/experiments/two_in_flight.py).  Latency per frame does not improve; throughput does.
Results are those of `WCT.stylize_prepared` on the same frame, bit for bit (same kernels, same statistics).
"""
from __future__ import annotations

from typing import Callable, Iterable, List, Optional

import torch


class FramePipeline:
    def __init__(self, make_engine: Callable[[], object], slots: int = 2, alpha: float = 1.0):
        """make_engine() -> a wct_hip.WCT (called `slots` times; each engine owns its device buffers)."""
        if slots < 1:
            raise ValueError("slots must be >= 1")
        self.engines = [make_engine() for _ in range(slots)]
        self.streams = [torch.cuda.Stream() for _ in range(slots)]
        self.alpha = alpha
        self._have_style = False

    @torch.no_grad()
    def set_style(self, style: torch.Tensor) -> None:
        """Style statistics once (engine 0), then copied into the other engines."""
        first = self.engines[0]
        first.style_prepare(style)
        for lvl in (5, 4, 3, 2, 1):
            stats = first.style_export(lvl)
            for e in self.engines[1:]:
                e.style_import(lvl, stats)
        torch.cuda.synchronize()
        self._have_style = True

    @torch.no_grad()
    def stylize_many(self, contents: Iterable[torch.Tensor], style: Optional[torch.Tensor] = None, num_run: int = 1) -> List[torch.Tensor]:
        """Stylise every content frame (CHW or 1CHW fp32 CUDA tensors, any sizes); results in input order."""
        if style is not None:
            self.set_style(style)
        if not self._have_style:
            raise RuntimeError("FramePipeline: set_style(style) first (or pass style=)")
        outs: List[torch.Tensor] = []
        caller = torch.cuda.current_stream()
        for s in self.streams:
            s.wait_stream(caller)           # frames produced on the caller's stream are visible to the slots
        for i, frame in enumerate(contents):
            k = i % len(self.engines)
            with torch.cuda.stream(self.streams[k]):
                frame.record_stream(self.streams[k])
                out = self.engines[k].stylize_prepared(frame, self.alpha, num_run)
                out.record_stream(caller)   # allocated on the slot's stream, consumed (and freed) on the caller's
                outs.append(out)
        for s in self.streams:
            caller.wait_stream(s)           # and the results to the caller's stream
        return outs
