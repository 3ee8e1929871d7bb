"""Deterministic synthetic weights and GOP-12 clips for the AR-Seg LR-branch hot path.

There is no network in the build or GPU environment, so neither the CamVid /
Cityscapes datasets nor the trained checkpoints (reference README.md:52,60) are
available.  Everything the tests, the golden generator and bench.py feed to the
path comes from here, seeded, so the same tensors can be regenerated anywhere:

* ``synth_state_dict``  -- weights for a given ordered ``(key, shape)`` list.  Every
  tensor depends only on ``(seed, key, shape)`` (not on key order), BN statistics are
  deliberately non-trivial so BN-folding bugs are visible.
* ``make_clip``         -- one GOP: keyframe + 11 non-key frames + int16 quarter-pel
  motion-vector maps in the on-disk format of the reference datasets
  (``dataset/camvid.py:624-626``: little-endian int16 ``[H,W,2]``, ``/4`` -> pixels;
  integer-pel values, block constant, as ``pre-process/generate_compressed_dataset_camvid.py:53-54``
  emits them).
"""
from __future__ import annotations

import zlib
from collections import OrderedDict
from typing import Dict, Iterable, Tuple

import numpy as np

# per-model input normalisation of the reference datasets
CAMVID_MEAN = (0.39068785, 0.40521392, 0.41434407)      # dataset/camvid.py:505
CAMVID_STD = (0.29652068, 0.30514979, 0.30080369)
CITY_BISE_MEAN = (0.3257, 0.3690, 0.3223)                # dataset/cityscapes.py:211-212
CITY_BISE_STD = (0.2112, 0.2148, 0.2115)


def _rng(seed: int, key: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(np.random.SeedSequence([int(seed), zlib.crc32(key.encode())])))


def synth_tensor(seed: int, key: str, shape: Tuple[int, ...], is_bn: bool, attn_gain: float = 0.12, res_gain: float = 0.3) -> np.ndarray:
    """One tensor of a synthetic state_dict (float32; int64 zeros for BN counters)."""
    g = _rng(seed, key)
    leaf = key.rsplit(".", 1)[-1]
    shape = tuple(int(s) for s in shape)
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "running_var":
        return g.uniform(0.5, 1.5, shape).astype(np.float32)
    if is_bn and leaf == "weight":
        return g.uniform(0.75, 1.25, shape).astype(np.float32)
    if is_bn and leaf == "bias":
        return (0.1 * g.standard_normal(shape)).astype(np.float32)
    if leaf == "weight" and len(shape) == 1:           # PReLU slope (single shared parameter)
        return g.uniform(0.1, 0.4, shape).astype(np.float32)
    if leaf == "weight":
        fan_in = int(np.prod(shape[1:]))
        std = float(np.sqrt(2.0 / max(fan_in, 1)))
        if "fuse_attention" in key or key.startswith(("lr_query_conv", "hr_key_conv", "hr_value_conv")):
            std *= attn_gain
        elif key.endswith("conv2.weight") and len(shape) == 4 and shape[0] == shape[1]:
            # last conv of a residual block: the synthetic BN statistics do not re-normalise, so an unscaled
            # branch would double the variance at every block (x16 in std through ResNet-18) and drive the CReFF
            # scores into a saturated softmax that amplifies fp32 rounding.  Trained nets have O(1) features.
            std *= res_gain
        return (std * g.standard_normal(shape)).astype(np.float32)
    if leaf == "bias":
        return (0.05 * g.standard_normal(shape)).astype(np.float32)
    raise ValueError(f"unrecognised state_dict key {key!r}")


def synth_state_dict(spec: Iterable[Tuple[str, Tuple[int, ...]]], seed: int = 0, attn_gain: float = 0.12, res_gain: float = 0.3) -> "OrderedDict[str, np.ndarray]":
    """``spec`` = ordered (key, shape) pairs, e.g. from ``module.state_dict()``."""
    spec = [(k, tuple(s)) for k, s in spec]
    keys = {k for k, _ in spec}
    out: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for k, s in spec:
        parent = k.rsplit(".", 1)[0] if "." in k else ""
        is_bn = (parent + ".running_mean" if parent else "running_mean") in keys
        out[k] = synth_tensor(seed, k, s, is_bn, attn_gain, res_gain)
    return out


def load_synth_weights(module, seed: int = 0, attn_gain: float = 0.12, res_gain: float = 0.3):
    """Fill a torch module (reference or ours) in place with the synthetic state_dict.  ``attn_gain`` scales the CReFF depthwise
    convs, ``res_gain`` the last conv of every residual block; (1.0, 1.0) = plain He initialisation everywhere ("un-damped")."""
    import torch

    sd = module.state_dict()
    syn = synth_state_dict([(k, tuple(v.shape)) for k, v in sd.items()], seed, attn_gain, res_gain)
    module.load_state_dict(OrderedDict((k, torch.from_numpy(v)) for k, v in syn.items()))
    return module


# ----------------------------------------------------------------------------------------------
# synthetic GOP clips
# ----------------------------------------------------------------------------------------------

def _texture(g: np.random.Generator, H: int, W: int) -> np.ndarray:
    """Smooth random RGB texture in [0,1]: sum of low-frequency sinusoids + a little noise."""
    yy, xx = np.meshgrid(np.arange(H, dtype=np.float32), np.arange(W, dtype=np.float32), indexing="ij")
    img = np.zeros((H, W, 3), dtype=np.float32)
    for c in range(3):
        acc = np.zeros((H, W), dtype=np.float32)
        for _ in range(6):
            fy, fx = g.uniform(0.5, 6.0, 2) * 2 * np.pi / np.array([H, W])
            acc += g.uniform(0.3, 1.0) * np.sin(fy * yy + fx * xx + g.uniform(0, 2 * np.pi)).astype(np.float32)
        img[..., c] = acc
    img = (img - img.min()) / (img.max() - img.min() + 1e-6)
    img += 0.03 * g.standard_normal(img.shape).astype(np.float32)
    return np.clip(img, 0.0, 1.0)


def _block_mv(g: np.random.Generator, H: int, W: int, pan: np.ndarray, d: int) -> np.ndarray:
    """int16 [H,W,2] quarter-pel MVs (dx,dy), integer-pel, constant on a random 8..64-px block tiling."""
    mv = np.zeros((H, W, 2), dtype=np.int16)
    y = 0
    while y < H:
        bh = int(g.choice([8, 16, 32, 64]))
        x = 0
        while x < W:
            bw = int(g.choice([8, 16, 32, 64]))
            if g.uniform() < 0.10:                              # "intra" block: zero motion
                v = np.zeros(2)
            else:
                v = np.round(pan * d + g.normal(0.0, 2.0, 2))
            v = np.clip(v, -150, 150)                           # dataset/camvid.py:665 clamp
            mv[y:y + bh, x:x + bw, 0] = np.int16(v[0] * 4)
            mv[y:y + bh, x:x + bw, 1] = np.int16(v[1] * 4)
            x += bw
        y += bh
    return mv


def make_clip(seed: int, H: int, W: int, gop: int = 12, mean=CAMVID_MEAN, std=CAMVID_STD) -> Dict[str, np.ndarray]:
    """One synthetic GOP.

    Returns ``frames`` float32 [gop,3,H,W] (normalised like ToTensor+Normalize; frame 0 is the
    keyframe) and ``mv`` int16 [gop,H,W,2] quarter-pel motion of frame d back to the keyframe
    (``mv[0]`` is all zero).
    """
    g = _rng(seed, f"clip{H}x{W}")
    key = _texture(g, H, W)
    pan = g.uniform(-3.0, 3.0, 2)
    frames = np.empty((gop, 3, H, W), dtype=np.float32)
    mvs = np.zeros((gop, H, W, 2), dtype=np.int16)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    yy, xx = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    for d in range(gop):
        if d == 0:
            img = key
        else:
            mv = _block_mv(g, H, W, pan, d)
            mvs[d] = mv
            sy = np.clip(yy + mv[..., 1] // 4, 0, H - 1)
            sx = np.clip(xx + mv[..., 0] // 4, 0, W - 1)
            img = key[sy, sx] + 0.02 * g.standard_normal((H, W, 3)).astype(np.float32)
            img = np.clip(img, 0.0, 1.0)
        q = np.round(img * 255.0) / 255.0                        # decoded frames are uint8
        frames[d] = (np.transpose(q, (2, 0, 1)) - m) / s
    return {"frames": frames, "mv": mvs}


# BiSeNetV1 registers the sub-modules of ``conv_out`` a second time under other names
# (model/bisenet.py:428-430, 490-492): ``feat_conv_out`` is ``conv_out.conv`` and ``final_conv`` is
# ``conv_out.conv_out``.  A state_dict therefore lists every such tensor under two keys;
# ``load_state_dict`` copies in key order, so the value under the LATER key (the alias) wins.
BISENET_ALIASES = (("feat_conv_out.", "conv_out.conv."), ("final_conv.", "conv_out.conv_out."),
                   ("final_conv.", "cls.4."))          # last pair: pspnet_semseg.PSPNetWithFuse.final_conv = cls[-1]


def resolve_aliases(sd):
    """Return a copy of ``sd`` in which aliased keys (BiSeNet, Cityscapes PSPNet) hold the value load_state_dict would leave."""
    out = type(sd)(sd)
    for alias, canon in BISENET_ALIASES:
        for k in list(sd.keys()):
            if k.startswith(alias):
                ck = canon + k[len(alias):]
                if ck in out:
                    out[ck] = sd[k]
    return out


def make_mv_chain(seed: int, H: int, W: int, n_frames: int) -> np.ndarray:
    """Per-frame codec motion fields as mergeMotion reads them (pre-process/generate_compressed_dataset_camvid.py:16-17):
    int16 [n_frames+1, H, W, 3] = (mv_x, mv_y in quarter-pel, reference index), entry 0 unused.  Block-constant on 8/16-px
    blocks, odd quarter-pel values (rounding, incl. exact halves), reference indices 0..2 plus intra markers (-1, 5)."""
    g = np.random.Generator(np.random.PCG64(seed))
    out = np.zeros((n_frames + 1, H, W, 3), dtype=np.int16)
    for f in range(1, n_frames + 1):
        bs = 8 if f % 2 else 16
        hb, wb = (H + bs - 1) // bs, (W + bs - 1) // bs
        mv = g.integers(-70, 71, (hb, wb, 2))                      # quarter-pel, ~ +-17 px, all residues mod 4
        ref = g.choice(np.array([0, 0, 0, 1, 2, -1, 5]), (hb, wb))
        blk = np.concatenate([mv, ref[..., None]], axis=-1).astype(np.int16)
        out[f] = np.repeat(np.repeat(blk, bs, axis=0), bs, axis=1)[:H, :W]
    return out
