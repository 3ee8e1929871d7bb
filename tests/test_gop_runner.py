"""Host logic of the multi-GPU GOP runner, on CPU: the frame plan, and a world_size-2 gloo run whose
outputs must equal the single-process run bit for bit (sharding is by frame, no arithmetic changes)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from arseg_amd.gop import GopRunner, frame_plan, keyframe_owner, neighbor_plan


def test_frame_plan_covers_every_frame_once():
    for world in (1, 2, 4, 8):
        plan = frame_plan(world, 12, world)
        flat = [f for r in plan for f in r]
        assert len(flat) == len(set(flat)) == world * 11
        assert all(len(r) == 11 for r in plan)                      # balanced: every rank gets gop-1 frames
        assert {g for g, _ in flat} == set(range(world)) and {d for _, d in flat} == set(range(1, 12))
    assert keyframe_owner(5, 4) == 1


def test_neighbor_plan_spans_two_gops_per_rank():
    """The contiguous-run deal: every frame once, gop-1 frames per owned GOP per rank, and a rank's frames come from the GOPs
    it owns plus the GOPs of rank+1 only -- one foreign keyframe feature per owned GOP instead of world-1."""
    for world in (1, 2, 4, 8):
        for n_gops in (world, 2 * world):
            plan = neighbor_plan(n_gops, 12, world)
            flat = [f for r in plan for f in r]
            assert len(flat) == len(set(flat)) == n_gops * 11
            assert all(len(r) == 11 * n_gops // world for r in plan)
            for r, fr in enumerate(plan):
                owners = {keyframe_owner(g, world) for g, _ in fr}
                assert owners <= {r, (r + 1) % world}
                for g in {g for g, _ in fr}:                     # a GOP is cut once: its first half to the previous owner, the rest to its owner
                    ds = sorted(d for gg, d in fr if gg == g)
                    assert ds == (list(range(6, 12)) if keyframe_owner(g, world) == r and world > 1 else
                                  list(range(1, 6)) if world > 1 else list(range(1, 12)))


def _key_fn(k):
    return torch.tanh(k * 1.5) + 0.25


def _nonkey_fn(ref, frame, mv):
    return ref * frame.mean() + mv.float().sum() * 1e-3 + frame


def _data(n_gops, gop=12):
    g = torch.Generator().manual_seed(1)
    keys = {i: torch.randn(3, 8, 8, generator=g) for i in range(n_gops)}
    frames = {(i, d): torch.randn(3, 8, 8, generator=g) for i in range(n_gops) for d in range(1, gop)}
    mvs = {(i, d): torch.randint(-8, 8, (8, 8, 2), generator=g).to(torch.int16) for i in range(n_gops) for d in range(1, gop)}
    return keys, frames, mvs


def _phase1(frames):
    return frames * 2.0                  # (exactly invertible in floating point)


def _phase2(feat, refs, mvs):
    return torch.stack([_nonkey_fn(r, f / 2.0, m) for r, f, m in zip(refs, feat, mvs)])


def _worker(rank, world, port, q, mode="batched"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_gops = 1 if mode in ("single", "loopback-broadcast") else world
    keys, frames, mvs = _data(n_gops)
    loop = {"loopback": "all_gather", "loopback-overlapped": "all_gather", "loopback-broadcast": "broadcast"}.get(mode, False)
    runner = GopRunner(_key_fn, _nonkey_fn, n_gops=n_gops, local=mode.startswith("local"), loopback=loop,
                       deal="neighbor" if mode.startswith("neighbor") else "round_robin")
    if mode.startswith("neighbor") and world > 1:
        assert runner.neighbor and {keyframe_owner(g, world) for g, _ in runner.plan} == {rank, (rank + 1) % world}
    if loop:
        assert world == 1 and runner.loopback and runner.single_gop == (loop == "broadcast")
        calls = []
        for name in ("all_gather_into_tensor", "broadcast"):
            def spy(*a, _f=getattr(dist, name), _n=name, **k):
                calls.append(_n)
                return _f(*a, **k)
            setattr(dist, name, spy)
    like = torch.empty(3, 8, 8)
    if mode.startswith("local"):
        assert runner.plan == [(rank, d) for d in range(1, 12)]       # whole GOP `rank`, nothing from the other ranks' GOPs
    if mode in ("overlapped", "local-overlapped", "loopback-overlapped", "neighbor-overlapped"):        # HR forward -> exchange || phase 1 -> phase 2, the rank's frames as one batch
        fs = torch.stack([frames[f] for f in runner.plan])
        ms = torch.stack([mvs[f] for f in runner.plan])
        res = runner.run_overlapped({g: keys[g] for g in runner.my_gops}, fs, ms, _phase1, _phase2)
        out = {f: res[i] for i, f in enumerate(runner.plan)}
    else:
        out = runner.run({g: keys[g] for g in runner.my_gops}, {f: frames[f] for f in runner.plan}, {f: mvs[f] for f in runner.plan}, like=like)
    if loop:          # the one-rank group did issue its collective (it is not short-cut as without a process group)
        assert calls == ["broadcast" if loop == "broadcast" else "all_gather_into_tensor"], calls
    hist = torch.tensor([float(len(out))])
    dist.all_reduce(hist)                                            # the confusion-matrix reduction pattern
    q.put((rank, {k: v.numpy().copy() for k, v in out.items()}, float(hist)))      # by value: a shared-memory tensor handle dies with the worker
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world,mode", [(2, "batched"), (2, "overlapped"), (2, "single"), (4, "batched"), (4, "single"),
                                        (8, "overlapped"), (8, "single"), (2, "local"), (4, "local-overlapped"),
                                        (1, "loopback"), (1, "loopback-overlapped"), (1, "loopback-broadcast"),
                                        (2, "neighbor"), (4, "neighbor-overlapped"), (8, "neighbor")])
def test_multi_rank_gloo_matches_single_process(world, mode):
    """world 2 / 4 / 8 over gloo == the single-process run, bit for bit: the batched plan (all-gather), the overlapped schedule (exchange
    concurrent with phase 1) and the single-GOP plan (owner broadcasts ref_p, 11 frames dealt over the ranks -- at world 8 three ranks
    get two frames and five get one: the literal north-star configuration, BASELINE configs[3] is the batched plan at world 8); and the
    zero-communication comparison plan of SURVEY 8e ("local": rank g keeps GOP g whole, no collective on the data path).  world 1
    "loopback": a one-rank group that still issues its collective (how the exchange code meets RCCL on a 1-GPU box).  "neighbor": the
    contiguous-run deal of round 6 -- one grouped send / receive of the keyframe feature between ring neighbours instead of the all-gather."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    n_gops = 1 if mode in ("single", "loopback-broadcast") else world
    keys, frames, mvs = _data(n_gops)
    single = GopRunner(_key_fn, _nonkey_fn, n_gops=n_gops).run(keys, frames, mvs)     # no process group -> world 1
    merged = {}
    for _, out, total in results:
        assert total == n_gops * 11
        merged.update(out)
    if mode == "single":
        sizes = sorted(len(out) for _, out, _ in results)
        assert sizes == sorted([11 // world + (r < 11 % world) for r in range(world)])      # 11 frames dealt over the ranks
    assert set(merged) == set(single)
    for k in single:
        assert torch.equal(torch.from_numpy(merged[k]), single[k])


def _range_worker(rank, world, port, q):
    """Rank 0's pass trips the operand-range word, rank 1's does not: both must repeat (same collectives on every rank) and the histogram
    is reduced once, from the final pass."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from arseg_amd import evaluation as ev
    from arseg_amd import ops

    state = {"math": "f16x3", "runs": 0, "resets": 0}
    ops.range_tripped = lambda: state["math"] == "f16x3" and rank == 0 and state["runs"] > 0

    def set_math(name):
        prev, state["math"] = state["math"], name
        return prev

    ops.set_conv_math = set_math

    def run():
        state["runs"] += 1
        h = torch.zeros(3, 3, dtype=torch.int64)
        h[rank, rank] = 10 * state["runs"] + (1 if state["math"] == "f32" else 0)      # the pass and its arithmetic are visible in the result
        h[rank, 2] = state["runs"]
        h[2, 2] = 5
        return h

    def reset():
        state["resets"] += 1

    miou = ev._range_safe(run, [1, 2, 3], 3, reset)
    q.put((rank, state["runs"], state["resets"], state["math"], miou))
    dist.barrier()
    dist.destroy_process_group()


def test_range_fallback_is_collective():
    """the decision to repeat an evaluation pass on the fp32 back end is taken by all ranks together."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_range_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=240) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, runs, resets, math, miou in results:
        assert runs == 2 and resets == 1 and math == "f16x3"            # both ranks repeated once, counters reset, back end restored
    # the histogram that is reduced is the sum of the SECOND passes only
    from arseg_amd import evaluation as ev
    want = ev._miou(torch.tensor([[21, 0, 2], [0, 21, 2], [0, 0, 10]]), 3)
    assert results[0][4] == results[1][4] == want
