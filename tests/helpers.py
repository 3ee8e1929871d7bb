"""Shared helpers for tests: synthetic state_dicts from the committed manifest."""
import numpy as np
import torch

from arseg_amd import synth


def sd_from_manifest(manifest, name, seed, attn_gain=0.12, res_gain=0.3):
    spec = [(k, tuple(s)) for k, s in manifest[name]["keys"]]
    sd = synth.resolve_aliases(synth.synth_state_dict(spec, seed, attn_gain, res_gain))
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def t(a):
    return torch.from_numpy(np.asarray(a))


def maxdiff(a, b):
    a = a.detach().cpu().double() if torch.is_tensor(a) else torch.from_numpy(np.asarray(a)).double()
    b = b.detach().cpu().double() if torch.is_tensor(b) else torch.from_numpy(np.asarray(b)).double()
    return float((a - b).abs().max())
