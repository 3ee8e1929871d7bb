#!/usr/bin/env python3
"""bench.py -- AR-Seg LR-branch hot path on MI355X: non-keyframe frames/s on synthetic GOP-12 clips.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1], the configuration the metric is quoted on): PSPNet-18, keyframe HR branch at
512x1024, 11 non-keyframes per GOP through the LR branch (0.5x -> 256x512 backbone) + CReFF at 512x1024, fp32.

A "step" is one GOP per rank, i.e. one pass of the hot path over one batch of synthetic input:
    keyframe HR forward -> (N > 1: RCCL all-gather of ref_p) -> 11 x [frame downscale + NHWC ingest, LR backbone,
    MV resize + warp, fused CReFF + classifier + log-softmax]
All inputs (frames, int16 MV maps, weights) are resident in HBM before the timed region.  `value` counts the
non-keyframes only, while the keyframe's HR forward and the exchange are inside the timed region (nothing skipped).

Arithmetic: fp32 tensors end to end.  The convolution GEMMs are evaluated on the fp16 matrix cores with every fp32 operand
split into hi + lo fp16 (22 significant bits) and three MFMAs per product, fp32 accumulation (`--conv-math f16x3`, default;
measured error vs an fp64 reference 1.5e-6 relative, the fp32 MFMA's is 2.3e-6 -- tests/test_gpu_ops.py); `--conv-math f32`
runs them on v_mfma_f32_32x32x2_f32 instead.

Extra objects on the JSON line: `roofline` (dominant kernel = the implicit-GEMM conv on the matrix cores), `roofline_creff`
(warp + CReFF stage against the HBM roofline, algorithmic bytes of SURVEY.md section 8d), `cpu_baseline` (the oracle,
i.e. a port, timed on the host cores for one non-keyframe of the same clip) and `parity` (max-abs error and argmax
agreement of that frame against the oracle).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

GOP, SCALE = 12, 0.5
# headline workload = BASELINE.json configs[1]; "psp2k" = the same network with the 512x1024 *non-key* reading of the metric
# (SURVEY.md section 8 preamble); "bise" = configs[2] (BiSeNet-18, Cityscapes sizes) measured in fp32 --
# the bf16 MFMA conv path that config names is not built yet, so it is an extra, not the headline.
CONFIGS = {
    "psp": dict(kind="psp", H=512, W=1024, n_cls=12, C=64, feat_div=1, ref_lr_gflop=116.9, ref_hr_gflop=468.2,
                label="PSPNet-18 HR keyframe 512x1024 + 11 non-keyframes LR 0.5x (256x512) + CReFF 7x7 @512x1024"),
    "psp2k": dict(kind="psp", H=1024, W=2048, n_cls=12, C=64, feat_div=1, ref_lr_gflop=467.5, ref_hr_gflop=1872.8,
                  label="PSPNet-18 HR keyframe 1024x2048 + 11 non-keyframes LR 0.5x (512x1024) + CReFF 7x7 @1024x2048"),
    "semseg": dict(kind="semseg", H=1024, W=2048, n_cls=19, C=512, feat_div=8, ref_lr_gflop=0.0, ref_hr_gflop=0.0,
                   label="Cityscapes PSPNet-18 (model/pspnet_semseg.py) HR keyframe 1024x2048 + 11 non-keyframes LR 0.5x (512x1024) + CReFF 7x7 C=512 @128x256"),
    "bise03": dict(kind="bise", H=1024, W=2048, n_cls=19, C=256, feat_div=8, ref_lr_gflop=0.0, ref_hr_gflop=242.8, scale=0.3,
                   label="BiSeNet-18 HR keyframe 1024x2048 + 11 non-keyframes LR 0.3x (307x614) + CReFF 7x7 @128x256 (BASELINE configs[4] shapes, fp32 tensors)"),
    "bise": dict(kind="bise", H=1024, W=2048, n_cls=19, C=256, feat_div=8, ref_lr_gflop=60.6, ref_hr_gflop=242.8,
                 label="BiSeNet-18 HR keyframe 1024x2048 + 11 non-keyframes LR 0.5x (512x1024) + CReFF 7x7 @128x256"),
}
PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD
PEAK_F16_MFMA_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense fp16/bf16 MFMA (v_mfma_f32_32x32x16_f16)
PEAK_HBM_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s (spec; ~6.3 TB/s achievable)


def build_nets(dev, cfg):
    from arseg_amd import synth
    from arseg_amd.model import BiSeNetV1, BiSeNetV1WithFuse, PSPNet, PSPNetWithFuse, pspnet_semseg

    N_CLS = cfg["n_cls"]
    if cfg["kind"] == "semseg":
        hr = pspnet_semseg.PSPNetWithFuse(bins=(1, 2, 3, 6), classes=N_CLS, feat_dim=512, layers=18)      # evaluation.py:27,34: both
        lr = pspnet_semseg.PSPNetWithFuse(bins=(1, 2, 3, 6), classes=N_CLS, feat_dim=512, layers=18)      # branches use this class
    elif cfg["kind"] == "psp":
        hr = PSPNet(sizes=(1, 2, 3, 6), n_classes=N_CLS, psp_size=512, deep_features_size=256, backend="resnet18")
        lr = PSPNetWithFuse(sizes=(1, 2, 3, 6), n_classes=N_CLS, psp_size=512, deep_features_size=256, backend="resnet18", atten_k=7)
    else:
        hr = BiSeNetV1(n_classes=N_CLS, backend="resnet18")
        lr = BiSeNetV1WithFuse(n_classes=N_CLS, backend="resnet18")
    synth.load_synth_weights(hr, 0)
    synth.load_synth_weights(lr, 1)
    sd_hr = {k: v.clone() for k, v in hr.state_dict().items()}
    sd_lr = {k: v.clone() for k, v in lr.state_dict().items()}
    return hr.to(dev).eval(), lr.to(dev).eval(), sd_hr, sd_lr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the (slow) CPU oracle leg")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-kernel event pass")
    ap.add_argument("--config", choices=sorted(CONFIGS), default="psp", help="psp = the headline workload (BASELINE configs[1])")
    ap.add_argument("--streams", type=int, default=3, help="HIP streams the GOP steps are rotated over (independent GOPs overlap)")
    ap.add_argument("--conv-math", choices=["f16x3", "f32", "f16"], default="f16x3",
                    help="MFMA back end of the fp32 conv GEMMs: f16x3 = split-fp16 emulation (3 fp16 MFMAs, fp32 accumulate), f32 = fp32 MFMA")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from arseg_amd import _lib, evaluation as ev, ops, synth
    from arseg_amd.gop import GopRunner

    _lib.load()
    ops.set_conv_math(args.conv_math)
    cfg = CONFIGS[args.config]
    global SCALE
    SCALE = cfg.get("scale", SCALE)
    H, W, N_CLS = cfg["H"], cfg["W"], cfg["n_cls"]
    mean, std = (synth.CAMVID_MEAN, synth.CAMVID_STD) if cfg["kind"] == "psp" else (synth.CITY_BISE_MEAN, synth.CITY_BISE_STD)
    hr, lr, sd_hr, sd_lr = build_nets(dev, cfg)

    # ---- synthetic batch: `world` GOPs; this rank owns keyframe `rank` and 11 round-robin non-keyframes
    def key_fn(key_img):
        return ops.to_nhwc(hr(key_img)[-1])[0]                        # ref_p, NHWC [Hp,Wp,C]

    def nonkey_fn(ref_p, img, mvq):
        out, p_c8 = ev.alter_res_step_fast(lr, ref_p.unsqueeze(0), img, mvq, SCALE)
        return out

    runner = GopRunner(key_fn, nonkey_fn, n_gops=world, gop=GOP)
    clips = {}
    needed = set(runner.my_gops) | {g for g, _ in runner.plan}
    for g in sorted(needed):
        clips[g] = synth.make_clip(g, H, W, gop=GOP, mean=mean, std=std)
    keyframes = {g: torch.from_numpy(clips[g]["frames"][0:1]).to(dev) for g in runner.my_gops}
    frames = {(g, d): torch.from_numpy(clips[g]["frames"][d:d + 1]).to(dev) for g, d in runner.plan}
    mvs = {(g, d): torch.from_numpy(clips[g]["mv"][d:d + 1]).to(dev) for g, d in runner.plan}

    frames_b = torch.cat([frames[f] for f in runner.plan])            # this rank's 11 non-keyframes, plan order
    mvs_b = torch.cat([mvs[f] for f in runner.plan])

    def batch_fn(refs, imgs, mvq):
        return ev.alter_res_batch_fast(lr, refs, imgs, mvq, SCALE)[0]

    def step():
        with torch.no_grad():
            return runner.run_batched(keyframes, frames_b, mvs_b, batch_fn)

    # Consecutive GOPs are independent: rotating them over a few HIP streams lets the MFMA-bound backbone convs of one GOP
    # run beside the VALU/LDS-bound warp + CReFF kernels of another.  Every step is fully executed; the timed region is
    # closed by a device-wide synchronize.
    streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, args.streams))]

    def run_steps(k):
        out = None
        for i in range(k):
            with torch.cuda.stream(streams[i % len(streams)]):
                out = step()
        return out

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = run_steps(args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)

    nonkey_per_step = world * (GOP - 1)
    result = {
        "metric": {"psp": "non-keyframe frames/sec (backbone+CReFF) at 512x1024",
                   "psp2k": "non-keyframe frames/sec (backbone+CReFF), PSPNet-18 1024x2048 / LR 512x1024",
                   "semseg": "non-keyframe frames/sec (backbone+CReFF), Cityscapes PSPNet-18 1024x2048 / LR 512x1024",
                   "bise03": "non-keyframe frames/sec (backbone+CReFF), BiSeNet-18 1024x2048 / LR 0.3x 307x614",
                   "bise": "non-keyframe frames/sec (backbone+CReFF), BiSeNet-18 1024x2048 / LR 512x1024"}[args.config],
        "value": nonkey_per_step * args.steps / elapsed,
        "unit": "frames/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f16" if args.conv_math == "f16" else "f32", "data": "synthetic",
        "conv_math": args.conv_math + {"f16x3": " (fp32 operands split into hi+lo fp16, 3 fp16 MFMAs per product, fp32 accumulate)",
                                       "f32": " (fp32 MFMA)", "f16": " (REDUCED PRECISION: plain fp16 operands, fp32 accumulate; not the headline)"}[args.conv_math],
        "streams": len(streams),
        "config": {"workload": cfg["label"] + ", GOP-12 synthetic clip per GPU, random-init (seeded) weights, fp32 tensors",
                   "gop": GOP, "frame": [H, W], "lr_scale": SCALE, "n_classes": N_CLS,
                   "parallelism": f"dp{world} (frames sharded round-robin, all-gather of keyframe features)"},
        "all_frames_per_s": world * GOP * args.steps / elapsed,
    }

    # ---- per-kernel timing with HIP events on the launch stream (one extra, instrumented step)
    if rank == 0 and not args.no_profile:
        g0, d0 = runner.plan[0]
        with torch.no_grad():
            ref_p = key_fn(keyframes[runner.my_gops[0]])
            with ops.profile() as prof_key:
                key_fn(keyframes[runner.my_gops[0]])
            with ops.profile() as prof_nk:
                for _ in range(3):
                    batch_fn([ref_p] * len(runner.plan), frames_b, mvs_b)
        nk = prof_nk.summary()
        ky = prof_key.summary()
        conv = nk["conv2d"]
        conv_k = ky["conv2d"]
        tf = lambda r: r["flops"] / (r["ms"] * 1e-3) / 1e12
        # dominant kernel of the step: conv_igemm_f32 (one batched LR pass of 11 frames + one HR frame).
        # `achieved` counts the FLOPs the MFMA kernel actually executes (Winograd F(4x4,3x3) and the folded pyramid
        # execute fewer than the reference's direct convs) over its own time = MFMA utilisation.
        tot_flops = conv["flops"] / 3 + conv_k["flops"]
        tot_ms = conv["ms"] / 3 + conv_k["ms"]
        n_launch = conv["launches"] / 3 + conv_k["launches"]
        wino_ms = sum(nk.get(k, {"ms": 0.0})["ms"] for k in ("wino_input", "wino_output")) / 3 + \
            sum(ky.get(k, {"ms": 0.0})["ms"] for k in ("wino_input", "wino_output"))
        ref_flops = ((GOP - 1) * cfg["ref_lr_gflop"] + cfg["ref_hr_gflop"]) * 1e9          # 2*MACs of every conv/linear of the reference (hook-counted)
        # f16x3: every GEMM MAC is three fp16 MFMA MACs -> executed matrix-core FLOPs = 3 x the GEMM FLOPs, against the fp16 peak
        mfma_mult, peak = {"f16x3": (3.0, PEAK_F16_MFMA_TFLOPS), "f16": (1.0, PEAK_F16_MFMA_TFLOPS), "f32": (1.0, PEAK_FP32_MFMA_TFLOPS)}[args.conv_math]
        gemm_tf = tot_flops / (tot_ms * 1e-3) / 1e12
        traffic = None
        tfile = os.path.join(ROOT, "profiles", "r01_conv_traffic.json")      # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of this command
        if os.path.exists(tfile) and args.config == "psp":
            with open(tfile) as f:
                traffic = json.load(f).get(args.conv_math, {}).get("hbm_bytes_per_launch")
        result["roofline"] = {
            "kernel": "conv_igemm_kernel<BM,BN,BK,NBUF,MATH> (implicit GEMM / batched Winograd GEMM; " +
                      {"f16x3": "3 x v_mfma_f32_32x32x16_f16 on hi/lo-split fp32 operands)", "f16": "v_mfma_f32_32x32x16_f16 on fp16-rounded operands)",
                       "f32": "v_mfma_f32_32x32x2_f32)"}[args.conv_math],
            "bound": "mfma", "achieved": mfma_mult * gemm_tf, "peak": peak, "unit": "TFLOP/s",
            "frac": mfma_mult * gemm_tf / peak,
            "traffic": traffic,
            "fp32_gemm_tflops": gemm_tf, "fp32_gemm_vs_fp32_mfma_peak": gemm_tf / PEAK_FP32_MFMA_TFLOPS,
            "per_launch": {"avg_gemm_flops": tot_flops / n_launch, "avg_ms": tot_ms / n_launch, "launches_per_step": n_launch},
            "lr_batch_gemm_tflops": tf(conv), "hr_frame_gemm_tflops": tf(conv_k),
            "reference_direct_conv_tflops": ref_flops / ((tot_ms + wino_ms) * 1e-3) / 1e12,
            "note": "achieved = matrix-core FLOPs executed by the conv kernel (GEMM FLOPs x3 under f16x3) / its time, against the "
                    "dense MFMA peak of the instruction used; fp32_gemm_tflops = the fp32 GEMM FLOPs it delivers (Winograd and the "
                    "folded pyramid execute fewer than the reference's direct convs); reference_direct_conv_tflops = the "
                    "reference's direct-convolution FLOP count (SURVEY 8d) / (conv kernel + Winograd transform time)",
        }
        # B = ref_p read + lr read + p write + int16x2 MV read + logits write, per non-keyframe
        C, fd = cfg["C"], cfg["feat_div"]
        Hp, Wp = H // fd, W // fd
        logit_px = Hp * Wp if cfg["kind"] == "semseg" else H * W              # pspnet_semseg phase 2 returns logits at feature resolution
        stage_bytes = 4 * C * Hp * Wp + 4 * C * (Hp // 2) * (Wp // 2) + 4 * C * Hp * Wp + 4 * H * W + 4 * N_CLS * logit_px
        nb = 3 * len(runner.plan)                                         # frames covered by the profiled launches
        zero = {"ms": 0.0, "flops": 0}
        fused = "creff_warp" in nk                                        # C == 64: one kernel (warp fused into the tile staging)
        cre, wrp = nk.get("creff_warp", nk.get("creff", zero)), nk.get("warp_mvq", zero)
        stage_ms = (cre["ms"] + wrp["ms"]) / nb
        result["roofline_creff"] = {
            "kernel": ("creff_rr_kernel<NB> (MV warp + CReFF + classifier, one kernel)" if fused else
                       "warp_mvq_nhwc_kernel + " + ("creff_mfma_kernel<NB>" if C >= 128 else "creff_kernel<7,NC,TH>") + " (MV warp, then fused CReFF + classifier)"),
            "bound": "hbm", "achieved": stage_bytes / (stage_ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": stage_bytes / (stage_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, "traffic": None,
            "algorithmic_bytes_per_frame": stage_bytes, "warp_ms_per_frame": wrp["ms"] / nb, "creff_ms_per_frame": cre["ms"] / nb,
            "creff_kernel_gflops": cre["flops"] / (cre["ms"] * 1e-3) / 1e9,
        }
        result["per_frame_ms"] = {"lr_frame_by_op": {k: v["ms"] / nb for k, v in sorted(nk.items())},
                                  "hr_keyframe_by_op": {k: v["ms"] for k, v in sorted(ky.items())}}

    # ---- CPU baseline (rank 0, N=1 only): the oracle (a port) on one non-keyframe of the same clip; also the parity check
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import cpu_ref

        # many-core hosts: the oracle's small strip-wise ops stop scaling (and collapse) far below 256 threads
        ncores = min(16, os.cpu_count() or 1)
        torch.set_num_threads(ncores)
        g0, d0 = runner.plan[0]
        img = torch.from_numpy(clips[g0]["frames"][d0:d0 + 1])
        key = torch.from_numpy(clips[g0]["frames"][0:1])
        mvq = torch.from_numpy(clips[g0]["mv"][d0:d0 + 1])
        with torch.no_grad():
            from arseg_amd.synth import resolve_aliases
            sd_hr, sd_lr = resolve_aliases(sd_hr), resolve_aliases(sd_lr)
            fwd = {"psp": cpu_ref.pspnet_forward, "bise": cpu_ref.bisenet_forward, "semseg": cpu_ref.semseg_forward}[cfg["kind"]]
            ref_cpu = fwd(sd_hr, key)[-1]                                         # outside the timed sample
            t1 = time.perf_counter()
            o_out, o_p, _, _ = cpu_ref.alter_res_step(cfg["kind"], sd_hr, sd_lr, img, key, cpu_ref.mv_from_int16(mvq), SCALE, ref_p=ref_cpu)
            cpu_s = time.perf_counter() - t1
        got = outs[0:1].cpu()                                              # plan[0] is the first frame of the batch
        ref_gpu = ops.as_nchw(key_fn(keyframes[g0]).unsqueeze(0)).cpu()
        result["cpu_baseline"] = {"value": 1.0 / cpu_s, "unit": "frames/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": "1 non-keyframe (downscale + LR backbone + MV resize + warp + CReFF + head) of the same "
                                            f"{H}x{W} clip with the PyTorch-CPU oracle; keyframe feature precomputed outside the sample",
                                  "seconds": cpu_s}
        result["parity"] = {"max_abs_err_logprobs": float((got - o_out).abs().max()),
                            "max_abs_err_keyframe_feature": float((ref_gpu - ref_cpu).abs().max()),
                            "argmax_agreement": float((got.argmax(1) == o_out.argmax(1)).float().mean()),
                            "tolerance": 1e-3}
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
