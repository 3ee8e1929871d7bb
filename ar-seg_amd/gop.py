"""Data-parallel GOP runner: frames of GOP-12 clips sharded over the GPUs of one node.

The reference has no multi-GPU inference path (``nn.DataParallel`` is bypassed on the hot path,
evaluation.py:190-193, and evaluation runs batch 1); this is the design ``BASELINE.json: north_star``
mandates on top of it.  Two plans, both with ONE exchange step over RCCL/xGMI
(``torch.distributed`` backend "nccl") and nothing else on the data path:

* **batched GOPs** (``n_gops`` a multiple of the world size; BASELINE configs[3]): rank ``g`` owns keyframe ``g`` and runs
  the HR forward for it; one all-gather of the (un-warped) keyframe features ``ref_p``; the
  ``n_gops * (gop-1)`` non-keyframes are dealt round-robin (``frame f -> rank f % world``), ``gop-1`` per rank per GOP;
* **single GOP** (``n_gops == 1`` with ``world > 1``; the literal reading of the north star): the owner (rank 0) runs the
  HR forward and broadcasts ``ref_p``; the ``gop-1`` non-keyframes are dealt round-robin over the ranks (11 frames over 8
  ranks: 2 + 1, the imbalance SURVEY.md section 8e notes).

* **neighbor** (``deal="neighbor"``; round 6, the same sharding idea with xGMI's point-to-point links in mind): frames are dealt in
  CONTIGUOUS runs that straddle a GOP boundary -- rank ``r`` takes the second half of each GOP it owns (d = 6..11) and the first half of
  the next GOP (d = 1..5), so every GOP's frames are still sharded over two GPUs, but a rank needs exactly ONE foreign keyframe
  feature per owned GOP: one point-to-point send to rank ``r-1`` and one receive from rank ``r+1`` (134 MB inbound per GOP for the
  PSPNet feature, over one direct link, against 7 x 134 MB inbound from an all-gather whose round-robin deal makes every rank's 11
  frames span all eight GOPs).  Same work per rank, bit-equal outputs; ``bench.py --gpus N`` prints it beside the other plans.

* **local** (``local=True``; the zero-communication comparison line of SURVEY.md section 8e, not the mandated design): rank ``g`` keeps
  GOP ``g`` whole -- its own keyframe and that GOP's ``gop-1`` non-keyframes -- and nothing is exchanged.  Same work per rank as the
  batched plan, so the two rates differ by exactly what the exchange costs (``bench.py --gpus N`` prints both).

Each non-keyframe needs only its own pixels, its own accumulated MV map and its GOP's ``ref_p`` -- no
frame-to-frame recurrence (evaluation.py:161-193).  Outputs stay rank-local; only the confusion matrix
is all-reduced (the reference's dormant ``dist.all_reduce(hist)``, evaluation.py:134-135).

Overlap: the LR backbone (phase 1) of a rank's non-keyframes does not depend on ``ref_p``.  ``run_overlapped`` therefore
enqueues the exchange on a side stream right after the HR forward, runs phase 1 of the whole local batch on the main
stream while the collective is in flight, and joins the streams only in front of phase 2 (warp + CReFF).  The
gather buffer is allocated once per shape and reused.

The runner is generic over the per-frame functions so that the sharding / exchange logic is testable on
CPU with the gloo backend (tests/test_gop_runner.py); bench.py and the GPU tests plug in the HIP path.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

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


def neighbor_plan(n_gops: int, gop: int, world: int) -> List[List[Tuple[int, int]]]:
    """The contiguous-run deal: plan[rank] = for each GOP g the rank owns, the second half of g (d > (gop-1)//2 ... ) followed by the first
    half of GOP (g+1) % n_gops (owned by rank+1).  Every frame exactly once, gop-1 frames per owned GOP per rank."""
    first = (gop - 1) // 2                 # frames d = 1..first of a GOP go to the previous GOP's owner
    plan: List[List[Tuple[int, int]]] = [[] for _ in range(world)]
    for g in range(n_gops):
        r = keyframe_owner(g, world)
        plan[r] += [(g, d) for d in range(first + 1, gop)]
        plan[r] += [((g + 1) % n_gops, d) for d in range(1, first + 1)]
    return plan


class GopRunner:
    """key_fn(keyframe) -> ref_p tensor; nonkey_fn(ref_p, frame, mv) -> output.

    ``run(keyframes, frames, mvs)``: ``keyframes`` = {gop index: keyframe tensor} for the GOPs this rank
    owns, ``frames`` / ``mvs`` = {(gop index, d): tensor} for the non-keyframes of this rank's plan.
    Returns {(gop index, d): output} for this rank's frames.
    """

    def __init__(self, key_fn: Callable, nonkey_fn: Callable, n_gops: int, gop: int = 12, group=None, local: bool = False, loopback: bool = False,
                 deal: str = "round_robin"):
        if deal not in ("round_robin", "neighbor"):
            raise ValueError(f"deal must be 'round_robin' or 'neighbor', got {deal!r}")
        self.key_fn, self.nonkey_fn = key_fn, nonkey_fn
        self.n_gops, self.gop, self.group = n_gops, gop, group
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if self.world > 1 else 0
        # loopback: a process group of ONE rank still issues the collective (side stream, gather buffer, the RCCL call itself) instead of
        # short-cutting it -- the way to put the exchange code through RCCL on a 1-GPU box (tests, bench.py ARSEG_RCCL_LOOPBACK=1)
        # (True / "all_gather": the batched plan's collective; "broadcast": the single-GOP plan's)
        self.loopback = bool(loopback) and self.world == 1 and dist.is_available() and dist.is_initialized()
        self.local = bool(local) and self.world > 1
        self.timing = None           # enable_timing(): HIP events around the exchange and phase 1 of every run_overlapped step
        self.single_gop = n_gops < self.world or (self.loopback and loopback == "broadcast" and n_gops == 1)
        if self.local and self.single_gop:
            raise ValueError("the local plan keeps whole GOPs per rank: it needs n_gops to be a multiple of the world size")
        if self.single_gop and n_gops != 1:
            raise ValueError(f"n_gops ({n_gops}) must be 1 (single-GOP plan) or a multiple of the world size ({self.world})")
        if not self.single_gop and n_gops % self.world:
            raise ValueError(f"n_gops ({n_gops}) must be a multiple of the world size ({self.world})")
        self.my_gops = [g for g in range(n_gops) if keyframe_owner(g, self.world) == self.rank]
        self.neighbor = deal == "neighbor" and not self.local and not self.single_gop
        self.plan = ([(g, d) for g in self.my_gops for d in range(1, gop)] if self.local else
                     neighbor_plan(n_gops, gop, self.world)[self.rank] if self.neighbor else frame_plan(n_gops, gop, self.world)[self.rank])
        # gather buffer (1 GB at world 8 for the PSPNet feature) and side stream, one pair PER LAUNCH STREAM: steps rotated over several
        # streams run concurrently, and a single buffer would be overwritten by the next step's collective while this step's warp +
        # CReFF still read it (the side stream's wait on ITS launch stream orders a buffer's reuse behind its previous consumer)
        self._gather_bufs: Dict[int, torch.Tensor] = {}
        self._side_streams: Dict[int, "torch.cuda.Stream"] = {}

    # ------------------------------------------------------------------ the exchange step
    def _buffer(self, stacked: torch.Tensor, replicas: Optional[int] = None) -> torch.Tensor:
        """The lane's exchange buffer: `replicas` x stacked.shape[0] entries (all-gather: one slot per rank; broadcast: exactly one)."""
        shape = ((self.world if replicas is None else replicas) * stacked.shape[0],) + tuple(stacked.shape[1:])
        key = self._lane if stacked.is_cuda else 0
        b = self._gather_bufs.get(key)
        if b is None or tuple(b.shape) != shape or b.dtype != stacked.dtype or b.device != stacked.device:
            b = self._gather_bufs[key] = torch.empty(shape, dtype=stacked.dtype, device=stacked.device)
        return b

    def _index(self, gathered: torch.Tensor, per_rank: int) -> List[torch.Tensor]:
        refs: List[Optional[torch.Tensor]] = [None] * self.n_gops
        for r in range(self.world):
            owned = [g for g in range(self.n_gops) if keyframe_owner(g, self.world) == r]
            for i, g in enumerate(owned):
                refs[g] = gathered[r, i]
        return refs

    _lane = 0          # launch stream of the step being enqueued (key of its gather buffer / side stream)

    def exchange(self, local_refs: Sequence[torch.Tensor], like: Optional[torch.Tensor] = None) -> List[torch.Tensor]:
        """Keyframe features for every GOP, indexed by gop.  Batched plan: one all-gather; single-GOP plan: one broadcast from
        the owner (``like``: a tensor with the feature's shape / dtype / device for the ranks that own no keyframe)."""
        if self.world == 1 and not self.loopback:
            return list(local_refs)
        if self.local:                    # whole GOPs per rank: the features never leave the rank (indexed by gop like the other plans)
            return dict(zip(self.my_gops, local_refs))
        if self.neighbor:
            return self._exchange_neighbor(local_refs)
        if self.single_gop:
            if self.rank == 0:
                buf = local_refs[0].contiguous()
            else:
                if like is None:
                    raise ValueError("single-GOP plan: ranks without the keyframe need `like` (shape / dtype / device of ref_p)")
                buf = self._buffer(like.unsqueeze(0), replicas=1)[0]          # exactly ref_p's shape (not world x: 1 GB at world 8)
            dist.broadcast(buf, src=0, group=self.group)
            return [buf]
        per_rank = len(local_refs)
        stacked = torch.stack(list(local_refs)) if per_rank > 1 else local_refs[0].unsqueeze(0)
        flat = self._buffer(stacked)
        dist.all_gather_into_tensor(flat, stacked.contiguous(), group=self.group)        # concatenation along dim 0
        return self._index(flat.view((self.world, per_rank) + tuple(stacked.shape[1:])), per_rank)

    def _exchange_neighbor(self, local_refs: Sequence[torch.Tensor]) -> Dict[int, torch.Tensor]:
        """One point-to-point step: this rank's keyframe features go to rank-1 (which holds the first halves of those GOPs' successors...
        precisely: rank r-1 processes the first half of every GOP rank r owns), the features of rank+1's GOPs arrive here.  Grouped
        send/recv (ncclSend / ncclRecv on RCCL: one direct xGMI link each way); world 1 (loopback): the successor GOP is local."""
        refs: Dict[int, torch.Tensor] = dict(zip(self.my_gops, local_refs))
        if self.world == 1:
            return refs
        nxt, prv = (self.rank + 1) % self.world, (self.rank - 1) % self.world
        stacked = (torch.stack(list(local_refs)) if len(local_refs) > 1 else local_refs[0].unsqueeze(0)).contiguous()
        inbox = self._buffer(stacked, replicas=1)
        # (rehearsals on the gloo backend with device tensors: gloo's point-to-point path takes host memory -- stage through it; RCCL sends device memory)
        via_host = stacked.is_cuda and dist.get_backend(self.group) == "gloo"
        snd, rcv = (stacked.cpu(), torch.empty(inbox.shape, dtype=inbox.dtype)) if via_host else (stacked, inbox)
        ops_ = [dist.P2POp(dist.isend, snd, prv, self.group), dist.P2POp(dist.irecv, rcv, nxt, self.group)]
        if self.world == 2 and self.rank == 1:      # (two ranks: both peers are the same process -- post the pair in the same order on both sides)
            ops_.reverse()
        for w in dist.batch_isend_irecv(ops_):
            w.wait()
        if via_host:
            inbox.copy_(rcv)
        theirs = [g for g in range(self.n_gops) if keyframe_owner(g, self.world) == nxt]
        for i, g in enumerate(theirs):
            refs[g] = inbox[i]
        return refs

    # ------------------------------------------------------------------ schedules
    def run(self, keyframes, frames, mvs, like: Optional[torch.Tensor] = None):
        local_refs = [self.key_fn(keyframes[g]) for g in self.my_gops]
        probe = local_refs[0] if local_refs else (like if like is not None else next(iter(frames.values()), None))
        if probe is not None and probe.is_cuda:          # also on ranks that own no keyframe (single-GOP plan): their buffer is per lane too
            self._lane = torch.cuda.current_stream().cuda_stream
        refs = self.exchange(local_refs, like)
        return {(g, d): self.nonkey_fn(refs[g], frames[(g, d)], mvs[(g, d)]) for (g, d) in self.plan}

    def run_batched(self, keyframes, frames_stacked, mvs_stacked, batch_fn, like: Optional[torch.Tensor] = None):
        """Same schedule with this rank's non-keyframes processed as ONE batch: ``frames_stacked`` / ``mvs_stacked`` hold
        the frames of ``self.plan`` in plan order along dim 0; ``batch_fn(refs_per_frame, frames, mvs)`` -> outputs."""
        local_refs = [self.key_fn(keyframes[g]) for g in self.my_gops]
        if frames_stacked.is_cuda:
            self._lane = torch.cuda.current_stream().cuda_stream
        refs = self.exchange(local_refs, like)
        return batch_fn([refs[g] for (g, _) in self.plan], frames_stacked, mvs_stacked)

    def run_overlapped(self, keyframes, frames_stacked, mvs_stacked, phase1_fn, phase2_fn, like: Optional[torch.Tensor] = None):
        """HR forward -> exchange (side stream) || ``phase1_fn(frames)`` (main stream) -> join -> ``phase2_fn(feat, refs_per_frame,
        mvs)``.  Phase 1 (frame downscale + LR backbone) does not read ``ref_p``, so the collective's latency hides behind it.  On
        CPU tensors (gloo tests) the exchange simply runs first; the result is identical."""
        local_refs = [self.key_fn(keyframes[g]) for g in self.my_gops]
        on_gpu = frames_stacked.is_cuda and (self.world > 1 or self.loopback) and not self.local
        if on_gpu:
            main = torch.cuda.current_stream()
            self._lane = main.cuda_stream
            side = self._side_streams.get(self._lane)
            if side is None:
                side = self._side_streams[self._lane] = torch.cuda.Stream(device=frames_stacked.device)
            side.wait_stream(main)                                  # the HR forward has produced local_refs; the lane's previous step is done with its buffer
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)] if self.timing is not None else None
            with torch.cuda.stream(side):
                if ev:
                    ev[0].record()
                refs = self.exchange(local_refs, like)
                if ev:
                    ev[1].record()
            if ev:
                ev[2].record()
            feat = phase1_fn(frames_stacked)                        # overlaps the collective
            if ev:
                ev[3].record()
                # bytes of ONE keyframe feature set as this rank sees it: from what the exchange returned (ranks that own no keyframe -- the
                # single-GOP plan's receivers -- have no local_refs)
                r0 = refs[0] if not isinstance(refs, dict) else next(iter(refs.values()))
                nbytes = r0.numel() * r0.element_size() * max(1, len(local_refs))
                self.timing.append((ev, nbytes))
                if len(self.timing) > 4096:          # bounded: a long run with timing left on keeps the most recent steps
                    del self.timing[:2048]
            main.wait_stream(side)
            for r in (refs.values() if isinstance(refs, dict) else refs):
                r.record_stream(main)
        else:
            refs = self.exchange(local_refs, like)
            feat = phase1_fn(frames_stacked)
        return phase2_fn(feat, [refs[g] for (g, _) in self.plan], mvs_stacked)

    # ------------------------------------------------------------------ diagnostics of the exchange step (bench.py --gpus N)
    def enable_timing(self, on: bool = True):
        """Record HIP events around the side-stream collective and around phase 1 of every ``run_overlapped`` step from now on (cheap: four
        event records per step).  ``exchange_stats()`` reads them."""
        self.timing = [] if on else None

    def exchange_stats(self):
        """What the first multi-GPU run needs to be self-diagnosing: per step the collective's duration on the side
        stream, the bytes this rank receives, the rate that implies, phase 1's duration on the main stream beside it, and whether the
        exchange ended before phase 1 did (= hidden).  Synchronises the device.  None if nothing was recorded."""
        if not self.timing:
            return None
        torch.cuda.synchronize()
        ex = [e[0].elapsed_time(e[1]) for e, _ in self.timing]
        p1 = [e[2].elapsed_time(e[3]) for e, _ in self.timing]
        lag = [e[3].elapsed_time(e[1]) for e, _ in self.timing]           # > 0: the collective finished AFTER phase 1 (exposed by that much)
        sent = self.timing[0][1]
        peers = 1 if self.neighbor else self.world - 1
        recv = sent * peers if not self.single_gop else (0 if self.rank == 0 else sent)
        n = len(ex)
        ex_ms = sum(ex) / n
        self.timing = []                  # the events are consumed: the next call reports the steps recorded from here on
        return {"steps": n, "plan": "broadcast" if self.single_gop else "neighbor_sendrecv" if self.neighbor else "all_gather", "exchange_ms": ex_ms, "exchange_ms_max": max(ex),
                "phase1_ms": sum(p1) / n, "bytes_sent_per_rank": sent, "bytes_received_per_rank": recv,
                "exchange_GBps_in": recv / (ex_ms * 1e-3) / 1e9 if ex_ms > 0 else None,
                "exposed_ms": sum(max(v, 0.0) for v in lag) / n, "hidden_behind_phase1": all(v <= 0.0 for v in lag)}
