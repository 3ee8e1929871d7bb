"""Hot-path pieces of the reference's ``evaluation.py`` on libarseg_hip.so.

* ``warpFeature``      -- evaluation.py:61-87
* ``resize_flow``      -- the MV resize block, evaluation.py:176-180
* ``EvalConstRes`` / ``EvalAlterRes`` -- evaluation.py:90-144 / 148-215, same call signatures
  (``dl`` is any iterable of the reference's sample tuples).  Networks may be bare modules or
  wrapped in ``nn.DataParallel`` (the reference addresses ``net.module`` on the hot path).
* ``alter_res_step_fast`` -- the same non-keyframe step on the kernel-native layouts (int16 MVs in,
  MV resize + warp + CReFF + head fused), used by the GOP runner and bench.py.

The CLI / dataset walking of the reference (evaluation.py:218-439) needs the datasets and
checkpoints and is out of scope.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist

from . import _lib, ops


def _unwrap(net):
    return net.module if hasattr(net, "module") else net


def warpFeature(feature, flow):
    """feature [B,C,H,W] (NCHW-contiguous or channels_last), flow [B,H,W,2] float32/float64 in feature pixels."""
    if ops.is16(feature):                  # 16-bit keyframe feature through the NCHW interface: warp in fp32
        feature = ops.cast(feature, torch.float32)
    if ops.is_nhwc_view(feature) and not feature.is_contiguous():
        return ops.as_nchw(ops.warp(ops.to_nhwc(feature), flow, _lib.NHWC))
    return ops.warp(feature.contiguous(), flow, _lib.NCHW)


def resize_flow(flow, Hp, Wp):
    """evaluation.py:176-180 for a flow [B,H,W,2]: both components are scaled by Hp/H, then bilinear(align_corners=True), in fp64.
    int16 input = the on-disk quarter-pel representation (``flow = int16 / 4``, dataset/camvid.py:625); float32 / float64 input =
    pixels, any values (what the reference's DataLoader hands over).  No host synchronisation."""
    if flow.dtype == torch.int16:
        return ops.mv_resize(flow, Hp, Wp)
    return ops.flow_resize(flow, Hp, Wp)


def _downscale_hw(H, W, scale):
    return int(H * scale), int(W * scale)


class EvalConstRes(object):
    """evaluation.py:90-144."""

    def __init__(self, scale=0.5, ignore_label=255):
        self.ignore_label = ignore_label
        self.scale = scale

    def __call__(self, net, dl, n_classes):
        return _range_safe(lambda: self._run(net, dl, n_classes), dl, n_classes)

    def _run(self, net, dl, n_classes):
        hist = torch.zeros((n_classes, n_classes), dtype=torch.int64, device="cuda")
        for imgs, label, *_ in dl:
            label = label.cuda()
            imgs = imgs.cuda()
            N, C, H, W = imgs.shape
            h, w = _downscale_hw(H, W, self.scale)
            if (h, w) != (H, W):
                imgs = _resize_frames(imgs, h, w)
            logits = net(imgs)[0]
            _, hist = ops.argmax_confusion(logits, label, label.shape[-2], label.shape[-1], hist, self.ignore_label, want_pred=False)
        return hist


def _range_safe(run, dl, n_classes, reset=None):
    """Runs an evaluation pass (``run()`` returns this rank's confusion matrix); if the split-fp16 convs met an activation outside their
    operand range (the sticky device word of ops.range_tripped -- read once, after the pass, which ends in a host read anyway), the pass is
    repeated on the fp32 matrix-core back end.  With torch.distributed initialised the decision is COLLECTIVE (max of the ranks' flags: a
    rank that did not trip repeats as well, so every rank issues the same collectives) and the histogram is all-reduced once, on the final
    pass only (evaluation.py:134-135).  A one-shot iterator cannot be replayed: that raises instead of returning a possibly clamped result."""
    ops.range_tripped()                      # clear what earlier launches left
    hist = run()
    tripped = bool(ops.range_tripped())
    multi = dist.is_available() and dist.is_initialized()
    if multi:
        flag = torch.tensor([1.0 if tripped else 0.0], device=hist.device)
        dist.all_reduce(flag, dist.ReduceOp.MAX)
        tripped = bool(flag.item() > 0)
    if tripped:
        if hasattr(dl, "__next__") or not hasattr(dl, "__iter__"):          # an iterator, not a re-iterable loader
            raise _lib.ArsegError("an activation left the split-fp16 operand range (|x| > 65504) and the data iterator cannot be replayed: "
                                  "evaluate with ops.set_conv_math('f32')")
        if reset is not None:
            reset()                          # per-pass counters start again
        prev = ops.set_conv_math("f32")
        try:
            hist = run()
        finally:
            ops.set_conv_math(prev)
    return _miou(hist, n_classes)


def _resize_frames(imgs, h, w):
    """F.interpolate(imgs, (h,w), bilinear, align_corners=True) on NCHW frames (evaluation.py:115-117)."""
    return ops.resize_nchw(imgs, h, w, _lib.BILINEAR, True)


def _miou(hist, n_classes):
    hist = hist.float()
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(hist, dist.ReduceOp.SUM)                     # evaluation.py:134-135
    ious = hist.diag() / (hist.sum(dim=0) + hist.sum(dim=1) - hist.diag())
    return ious.mean().item()


class EvalAlterRes(object):
    """evaluation.py:148-215: keyframe through the HR net, non-keyframe through the LR net + CReFF.

    ``cache_keyframe`` (extension): the reference recomputes the HR forward of the keyframe for
    every sample (evaluation.py:173) although the 11 non-keyframes of a GOP share it; with the flag set the keyframe feature is
    kept while consecutive samples carry the same reference frame (compared on the GPU).  The result is unchanged -- the HR
    forward is deterministic -- only ~10/11 of the HR forwards disappear when the loader is GOP ordered.  ``hr_forwards``
    counts the ones executed."""

    def __init__(self, scale=0.5, ignore_label=255, cache_keyframe=False):
        self.ignore_label = ignore_label
        self.scale = scale
        self.cache_keyframe = cache_keyframe
        self.hr_forwards = 0

    def __call__(self, highres_net, net, dl, n_classes):
        first = self.hr_forwards

        def reset():
            self.hr_forwards = first

        return _range_safe(lambda: self._run(highres_net, net, dl, n_classes), dl, n_classes, reset)

    def _run(self, highres_net, net, dl, n_classes):
        hist = torch.zeros((n_classes, n_classes), dtype=torch.int64, device="cuda")
        lr_net = _unwrap(net)
        last_ref, last_p = None, None
        for imgs, label, _, ref_imgs, flow in dl:
            label = label.cuda()
            imgs = imgs.cuda()
            flow = flow.cuda()
            ref_imgs = ref_imgs.cuda()
            if self.cache_keyframe and last_ref is not None and last_ref.shape == ref_imgs.shape and torch.equal(last_ref, ref_imgs):
                highres_ref_p = last_p
            else:
                highres_ref_p = highres_net(ref_imgs)[-1]                                     # :173-174
                self.hr_forwards += 1
                last_ref, last_p = ref_imgs, highres_ref_p
            flow = resize_flow(flow, highres_ref_p.shape[-2], highres_ref_p.shape[-1])       # :177-180
            highres_ref_p = warpFeature(highres_ref_p, flow)                                 # :183
            N, C, H, W = imgs.shape
            h, w = _downscale_hw(H, W, self.scale)
            imgs = _resize_frames(imgs, h, w)                                                # :186-188
            out_p = lr_net.forward_phase1(imgs)[-1]                                          # :190-191
            out, _ = lr_net.forward_phase2(out_p, highres_ref_p)                             # :193
            _, hist = ops.argmax_confusion(out, label, label.shape[-2], label.shape[-1], hist, self.ignore_label, want_pred=False)
        return hist


def alter_res_step_fast(lr_net, ref_p_nhwc, img, mv_q, scale=0.5):
    """One non-keyframe on the kernel-native layouts.

    ref_p_nhwc: keyframe feature, NHWC [N,Hp,Wp,C] (unwarped); img: NCHW frame; mv_q: int16 quarter-pel [N,H,W,2].
    Returns (logits NCHW, p in C8 layout).  Same arithmetic as EvalAlterRes' loop body; the frame downscale is fused
    into the NHWC4 ingest, the MV resize into the warp, and everything after the backbone into one CReFF kernel.
    """
    lr_net = _unwrap(lr_net)
    N, C, H, W = img.shape
    h, w = _downscale_hw(H, W, scale)
    feat = lr_net.phase1_nhwc4(ops.frame_ingest(img, h, w, lr_net.storage_dtype), aux=ops.config.aux_outputs)[-1]     # a3 + phase 1
    return lr_net.phase2_warp(feat, [ref_p_nhwc[i] for i in range(N)], mv_q)      # a2 + a1 + CReFF + head


def alter_res_batch_fast(lr_net, ref_ps, imgs, mv_qs, scale=0.5):
    """B non-keyframes in one pass (they have no dependence on each other, evaluation.py:161-193, so the LR backbone
    and CReFF run batched: large GEMM M, one launch sequence for the whole batch).

    ref_ps: sequence of B un-warped keyframe features, NHWC [Hp,Wp,C] each (frames of one GOP share theirs);
    imgs: [B,3,H,W]; mv_qs: int16 [B,H,W,2].  Returns (logits [B,n_cls,H',W'], p C8 [B,C/8,Hp,Wp,8]).
    """
    lr_net = _unwrap(lr_net)
    B, _, H, W = imgs.shape
    sub = ops.config.lr_subbatch
    if 0 < sub < B:                       # optional: bound the working set (Winograd V / M tensors) per pass
        outs = [alter_res_batch_fast(lr_net, ref_ps[i:i + sub], imgs[i:i + sub], mv_qs[i:i + sub], scale) for i in range(0, B, sub)]
        return torch.cat([o[0] for o in outs]), torch.cat([o[1] for o in outs])
    h, w = _downscale_hw(H, W, scale)
    feat = lr_net.phase1_nhwc4(ops.frame_ingest(imgs, h, w, lr_net.storage_dtype), aux=ops.config.aux_outputs)[-1]     # a3 + phase 1, batched
    return lr_net.phase2_warp(feat, list(ref_ps), mv_qs)               # a2 + a1 (each frame has its own MV map) + CReFF + head


def alter_res_phase1(lr_net, imgs, scale=0.5):
    """First half of ``alter_res_batch_fast``: frame downscale + ingest + LR backbone (evaluation.py:186-191).  Independent of the
    keyframe feature -- the multi-GPU runner overlaps it with the exchange of ``ref_p`` (arseg_amd/gop.py)."""
    lr_net = _unwrap(lr_net)
    B, _, H, W = imgs.shape
    h, w = _downscale_hw(H, W, scale)
    return lr_net.phase1_nhwc4(ops.frame_ingest(imgs, h, w, lr_net.storage_dtype), aux=ops.config.aux_outputs)[-1]


def alter_res_phase2(lr_net, feat, ref_ps, mv_qs):
    """Second half: MV resize + warp + CReFF + head on the phase-1 feature (evaluation.py:176-183,193) -> logits."""
    return _unwrap(lr_net).phase2_warp(feat, list(ref_ps), mv_qs)[0]


def alter_res_batch_pred(lr_net, ref_ps, imgs, mv_qs, scale=0.5, labels=None, hist=None, ignore_label=255):
    """B non-keyframes through backbone + warp + CReFF + head and the evaluator tail (evaluation.py:201-209) in one go:
    -> (pred int32 [B,H,W], hist int64 [n_cls,n_cls] | None).  For BiSeNet the head's 1/8-resolution logits go straight into
    the argmax (x8 upsample fused); the other networks' logits are resized (align_corners=True,
    the identity for PSPNet) inside the same argmax kernel."""
    lr_net = _unwrap(lr_net)
    B, _, H, W = imgs.shape
    h, w = _downscale_hw(H, W, scale)
    feat = lr_net.phase1_nhwc4(ops.frame_ingest(imgs, h, w, lr_net.storage_dtype), aux=ops.config.aux_outputs)[-1]
    fused_up = hasattr(lr_net, "out_upsample")                       # BiSeNetOutput: head -> nn.Upsample(x8, align_corners=False)
    if fused_up:
        lo, _ = lr_net.phase2_warp(feat, list(ref_ps), mv_qs, upsample=False)
        if (8 * lo.shape[-2], 8 * lo.shape[-1]) != (H, W):               # label size differs from 8x the head: two resizes, not fusable
            lo, fused_up = ops.resize_nchw(lo, 8 * lo.shape[-2], 8 * lo.shape[-1], _lib.BILINEAR, False), False
    else:
        lo, _ = lr_net.phase2_warp(feat, list(ref_ps), mv_qs)
    return ops.argmax_confusion(lo, labels, H, W, hist, ignore_label, align_corners=not fused_up)
