"""Independent contents, one style, N GPUs (BASELINE config 5: "replicas only").

Every rank stylises its OWN content image; nothing about the contents is exchanged.  The only thing the ranks share is the
style side (five encodes + moments + matrix square roots of the ONE style image, a third of a single-GPU step): level L's
style statistics are computed by rank (5 - L) mod N on its side stream and broadcast (C*C + C fp64 values, <= 132 KB),
instead of all N ranks repeating all five levels (data_loader.py:32-36 builds content x style pairs; SURVEY 8e "cfg5").
The engine is a wct_hip.WCT (RCCL = torch.distributed "nccl"); tests run the same code under gloo with a CPU checker.
"""
from __future__ import annotations

from typing import Optional

import torch


class ReplicaStylizer:
    def __init__(self, engine, dist, rank: Optional[int] = None, world: Optional[int] = None, alpha: float = 1.0):
        self.e, self.dist = engine, dist
        self.rank = (dist.get_rank() if dist is not None else 0) if rank is None else rank
        self.world = (dist.get_world_size() if dist is not None else 1) if world is None else world
        self.alpha = alpha

    @torch.no_grad()
    def share_style(self, style: torch.Tensor) -> None:
        """Compute this rank's share of the style statistics and exchange them: afterwards every rank holds all five levels."""
        e, dist, world, rank = self.e, self.dist, self.world, self.rank
        owner = lambda lvl: (5 - lvl) % world
        e.style_prepare(style, levels=[lvl for lvl in (5, 4, 3, 2, 1) if owner(lvl) == rank])
        if world == 1:
            return
        for L in (5, 4, 3, 2, 1):
            stats = e.style_export(L) if rank == owner(L) else None
            if stats is None:
                stats = torch.empty(e.style_stats_count(L), dtype=torch.float64, device=getattr(e, "stats_device", "cpu"))
            dist.broadcast(stats, src=owner(L))
            if rank != owner(L):
                e.style_import(L, stats)

    @torch.no_grad()
    def stylize(self, content: torch.Tensor, style: Optional[torch.Tensor] = None, num_run: int = 1) -> torch.Tensor:
        """This rank's content against the shared style statistics (`style` given: share_style(style) first)."""
        if style is not None:
            self.share_style(style)
        return self.e.stylize_prepared(content, self.alpha, num_run)
