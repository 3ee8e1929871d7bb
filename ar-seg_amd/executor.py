"""Static executor for the GOP step: per-shape plans + one memory arena + HIP-graph replay.

The first calls of a step run eagerly: they build the per-shape launch plans (conv tile / split-K / Winograd choices, cached in
``ops``) and size the workspaces.  ``GopGraph`` then captures ``lanes`` independent copies of the step -- consecutive GOPs have no
mutual dependence -- either on forked HIP streams into ONE graph (joined at the end of every replay) or, ``independent=True`` (what ``bench.py``
uses since round 3), as one graph per lane replayed on the lane's own stream: a replay re-issues the ~150 kernel launches of each step without any
Python, ctypes or allocator work (every intermediate lives at a fixed address in the graph's private pool: zero ``torch.empty`` in
the steady state), and the lanes keep the cross-step overlap the eager path gets from rotating streams (the MFMA-bound backbone convs of
one GOP beside the VALU-bound warp + CReFF kernel of another).

Capture rules the step must obey (it does): no host synchronisation, no ``.item()`` / ``.cpu()``, inputs read from fixed tensors
(``GopGraph`` replays on the SAME input tensors; refill them in place with ``copy_`` for new data)."""
from __future__ import annotations

from typing import Callable, List, Sequence

import torch


class GopGraph:
    def __init__(self, step_fns: Sequence[Callable[[], torch.Tensor]], warmup: int = 2, independent: bool = False):
        """step_fns: one closure per lane; each enqueues one full GOP step on the current stream and returns its output tensor.
        independent: one graph PER LANE, each replayed on its own stream -- no join between the lanes, so a lane's next step starts when its own
        previous step is done instead of when the slowest lane of the replay is (what the eager path gets from rotating streams; round 3:
        six lanes 1868-1889 frames/s joined, 1900-1945 without the join)."""
        self.lanes = len(step_fns)
        self.independent = independent
        dev = torch.cuda.current_device()
        for _ in range(max(1, warmup)):                 # eager: autotune plans, size workspaces, prime the allocator
            for fn in step_fns:
                fn()
        torch.cuda.synchronize()
        if independent:
            self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.lanes)]
            self.graphs, self.outputs = [], []
            for fn, st in zip(step_fns, self._streams):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=st):
                    self.outputs.append(fn())
                self.graphs.append(g)
            torch.cuda.synchronize()
            return
        self.graph = torch.cuda.CUDAGraph()
        self._streams = [torch.cuda.Stream(device=dev) for _ in range(self.lanes - 1)]
        self.outputs: List[torch.Tensor] = []
        with torch.cuda.graph(self.graph):
            main = torch.cuda.current_stream()
            outs = [None] * self.lanes
            for i, st in enumerate(self._streams):      # fork
                st.wait_stream(main)
                with torch.cuda.stream(st):
                    outs[i + 1] = step_fns[i + 1]()
            outs[0] = step_fns[0]()
            for st in self._streams:                    # join
                main.wait_stream(st)
            self.outputs = outs
        torch.cuda.synchronize()

    def replay(self, join: bool = True) -> List[torch.Tensor]:
        """Enqueue ``lanes`` GOP steps; returns their (static) output tensors.  join (independent mode): the caller's stream waits for every
        lane, so the outputs may be read -- and the static inputs refilled with ``copy_`` -- in stream order right after the call.
        ``join=False`` leaves the lanes running on their private streams (back-to-back replays keep the cross-step overlap: what the
        throughput benchmark wants); the caller then orders itself with ``synchronize()`` before it touches inputs or outputs."""
        if self.independent:
            cur = torch.cuda.current_stream()
            for g, st in zip(self.graphs, self._streams):
                st.wait_stream(cur)                     # (work enqueued before this call, e.g. an input refill on the caller's stream)
                with torch.cuda.stream(st):
                    g.replay()
            if join:
                self.synchronize()
            return self.outputs
        self.graph.replay()
        return self.outputs

    def synchronize(self) -> None:
        """Make the caller's stream wait for every lane (independent mode; the single graph joins by itself)."""
        if self.independent:
            cur = torch.cuda.current_stream()
            for st in self._streams:
                cur.wait_stream(st)
