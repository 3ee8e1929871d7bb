#!/usr/bin/env python3
"""Times the two Winograd transforms (split-row input transform, output transform with residual + ReLU) on the layer3 / layer4 shapes
of the headline network (GPU only; ARSEG_HIP_LIB selects a variant build of the library).

    python tools/bench_wino.py [--reps 30]"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from arseg_amd import _lib  # noqa: E402

SHAPES = [(11, 32, 64, 512, 1), (11, 32, 64, 512, 4), (11, 32, 64, 256, 2), (11, 32, 64, 256, 1), (11, 32, 64, 128, 1), (1, 64, 128, 512, 4), (1, 64, 128, 256, 2)]


def P(t):
    return ctypes.c_void_p(t.data_ptr() if t is not None else None)


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) / reps * 1e3


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--sets", type=int, default=1, help="buffer sets used in rotation (8: cold, every launch reads from HBM)")
    args = ap.parse_args()
    lib = _lib.load()
    dev = torch.device("cuda:0")
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    tot_i = tot_o = 0.0
    for N, H, W, C, dil in SHAPES:
        T = lib.arseg_wino43_tiles(N, H, W, dil)
        # --sets buffer sets in rotation: with one set the tensors stay in the 256 MB Infinity Cache (5-6 TB/s); in a GOP step they come from HBM
        S = args.sets
        x = [torch.randn(N, H, W, C, device=dev) for _ in range(S)]
        res = [torch.randn(N, H, W, C, device=dev) for _ in range(S)]
        out = [torch.empty(N, H, W, C, device=dev) for _ in range(S)]
        V = [torch.empty(36, T, C, device=dev) for _ in range(S)]
        M = [torch.randn(36, T, C, device=dev) for _ in range(S)]
        sc, bi = torch.ones(C, device=dev), torch.zeros(C, device=dev)
        it = [0]

        def run_in():
            k = it[0] = (it[0] + 1) % S
            _lib.check(lib.arseg_wino43_input_split_fwd(P(x[k]), C, P(V[k]), N, H, W, C, dil, 0, 0.0625, None, 0.0, st), "in")

        def run_out():
            k = it[0] = (it[0] + 1) % S
            _lib.check(lib.arseg_wino43_output_fwd(P(M[k]), P(sc), P(bi), P(res[k]), C, P(out[k]), C, N, H, W, C, dil, _lib.ACT_RELU, 0.0, 16.0, st), "out")

        ti = timeit(run_in, args.reps)
        to = timeit(run_out, args.reps)
        bi_ = (N * H * W * C * 4 + 36 * T * C * 4) / 1e3
        bo_ = (2 * N * H * W * C * 4 + 36 * T * C * 4 + N * H * W * C * 4 * 0) / 1e3
        print(f"N{N} {H}x{W} C{C} d{dil}: input {ti:7.1f} us ({bi_ / ti / 1e3:5.2f} TB/s)   output {to:7.1f} us ({(bo_ + N * H * W * C * 4 / 1e3) / to / 1e3:5.2f} TB/s)", flush=True)
        tot_i += ti
        tot_o += to
    print(f"sum input {tot_i:.1f} us, output {tot_o:.1f} us")


if __name__ == "__main__":
    main()
