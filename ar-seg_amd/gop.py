"""Data-parallel GOP runner: frames of GOP-12 clips sharded over the GPUs of one node.

The reference has no multi-GPU inference path (``nn.DataParallel`` is bypassed on the hot path,
evaluation.py:190-193, and evaluation runs batch 1); this is the design ``BASELINE.json: north_star``
mandates on top of it:

* a batch is ``world`` GOPs; rank ``g`` owns keyframe ``g`` and runs the HR forward for it;
* ONE exchange step: an all-gather of the (un-warped) keyframe features ``ref_p`` over RCCL/xGMI
  (``torch.distributed`` backend "nccl"); each non-keyframe needs only its own pixels, its own
  accumulated MV map and its GOP's ``ref_p`` -- no frame-to-frame recurrence (evaluation.py:161-193);
* the ``world * (gop-1)`` non-keyframes are dealt round-robin (``frame f -> rank f % world``), so every
  rank processes ``gop-1`` of them; outputs stay rank-local; only the confusion matrix is all-reduced
  (the reference's dormant ``dist.all_reduce(hist)``, evaluation.py:134-135).

The runner is generic over the two per-frame functions so that the sharding / exchange logic is testable
on CPU with the gloo backend (tests/test_gop_runner.py); bench.py and the GPU tests plug in the HIP path.
"""
from __future__ import annotations

from typing import Callable, List, Sequence, Tuple

import torch
import torch.distributed as dist


def frame_plan(n_gops: int, gop: int, world: int) -> List[List[Tuple[int, int]]]:
    """plan[rank] = [(gop index, distance d to the keyframe, 1 <= d < gop), ...] in processing order."""
    plan: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    f = 0
    for g in range(n_gops):
        for d in range(1, gop):
            plan[f % world].append((g, d))
            f += 1
    return plan


def keyframe_owner(g: int, world: int) -> int:
    return g % world


class GopRunner:
    """key_fn(keyframe) -> ref_p tensor; nonkey_fn(ref_p, frame, mv) -> output.

    ``run(keyframes, frames, mvs)``: ``keyframes`` = {gop index: keyframe tensor} for the GOPs this rank
    owns, ``frames`` / ``mvs`` = {(gop index, d): tensor} for the non-keyframes of this rank's plan.
    Returns {(gop index, d): output} for this rank's frames.
    """

    def __init__(self, key_fn: Callable, nonkey_fn: Callable, n_gops: int, gop: int = 12, group=None):
        self.key_fn, self.nonkey_fn = key_fn, nonkey_fn
        self.n_gops, self.gop, self.group = n_gops, gop, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        if n_gops % self.world:
            raise ValueError(f"n_gops ({n_gops}) must be a multiple of the world size ({self.world})")
        self.plan = frame_plan(n_gops, gop, self.world)[self.rank]
        self.my_gops = [g for g in range(n_gops) if keyframe_owner(g, self.world) == self.rank]

    def exchange(self, local_refs: Sequence[torch.Tensor]) -> List[torch.Tensor]:
        """All-gather of the keyframe features: returns ref_p for every GOP, indexed by gop."""
        if self.world == 1:
            return list(local_refs)
        per_rank = len(local_refs)
        stacked = torch.stack(list(local_refs)) if per_rank > 1 else local_refs[0].unsqueeze(0)
        flat = torch.empty((self.world * per_rank,) + tuple(stacked.shape[1:]), dtype=stacked.dtype, device=stacked.device)
        dist.all_gather_into_tensor(flat, stacked.contiguous(), group=self.group)        # concatenation along dim 0
        gathered = flat.view((self.world, per_rank) + tuple(stacked.shape[1:]))
        refs = [None] * self.n_gops
        for r in range(self.world):
            owned = [g for g in range(self.n_gops) if keyframe_owner(g, self.world) == r]
            for i, g in enumerate(owned):
                refs[g] = gathered[r, i]
        return refs

    def run(self, keyframes, frames, mvs):
        local_refs = [self.key_fn(keyframes[g]) for g in self.my_gops]
        refs = self.exchange(local_refs)
        return {(g, d): self.nonkey_fn(refs[g], frames[(g, d)], mvs[(g, d)]) for (g, d) in self.plan}

    def run_batched(self, keyframes, frames_stacked, mvs_stacked, batch_fn):
        """Same schedule with this rank's non-keyframes processed as ONE batch: ``frames_stacked`` / ``mvs_stacked`` hold
        the frames of ``self.plan`` in plan order along dim 0; ``batch_fn(refs_per_frame, frames, mvs)`` -> outputs."""
        local_refs = [self.key_fn(keyframes[g]) for g in self.my_gops]
        refs = self.exchange(local_refs)
        return batch_fn([refs[g] for (g, _) in self.plan], frames_stacked, mvs_stacked)
