"""Shared helpers for the tests: synthetic state dicts keyed by the reference schema, inputs, metrics."""
import json
import os
from functools import lru_cache

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
SEED = 1234


def schema(which: str, enc: str):
    with open(os.path.join(GOLD, f"schema_{which.rstrip('h')}_{enc}.json")) as f:   # "Ah" / "Bh": the same model, heavy weights
        return json.load(f)


@lru_cache(maxsize=4)
def synth_sd(which: str, enc: str):
    """Oracle-side state dict (CPU f32) with the synthetic weights the fixtures were made with."""
    from vdn import synth
    from oracle import ref_cpu as O
    sch = schema(which, enc)
    if enc == "vitg":   # 1.1 G parameters: torch's CPU generator (same torch build as the fixture generator), not the portable hash
        sd = dict(synth.fast_state_dict([(k, tuple(s)) for k, s in sch["params"]], SEED))
    else:
        sd = {k: torch.from_numpy(v) for k, v in synth.synth_state_dict([(k, tuple(s)) for k, s in sch["params"]], SEED).items()}
    for k, s in sch["buffers"]:
        b = synth.synth_buffer(SEED, k, tuple(s))   # BatchNorm statistics of the use_bn head ("Af" / "Bf" schemas)
        sd[k] = torch.from_numpy(np.asarray(b)) if b is not None else O.temporal_pe(s[-1], s[1])
    if which.endswith("h"):   # "Ah" / "Bh": checkpoint-like outliers on top (vdn/synth.heavy_overlay; fixtures *_heavy)
        synth.heavy_overlay(sd)
    return sd


def inputs(n, h, w):
    from vdn import synth
    return torch.from_numpy(synth.normalize_frames(synth.frames_u8(SEED, n, h, w)))


def rel_l2(a, b):
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return float((a - b).norm() / (b.norm() + 1e-30))


def worst_px(a, b):
    """max |a - b| relative to max |b|: a handful of badly wrong pixels (a tile tail, a halo) that a whole-map rel-L2 hides."""
    a = torch.as_tensor(a).double().reshape(-1)
    b = torch.as_tensor(b).double().reshape(-1)
    return float((a - b).abs().max() / (b.abs().max() + 1e-30))


def sample_idx(numel, n=256):
    return (np.arange(n, dtype=np.int64) * 2654435761 + 12345) % numel


def stats(t):
    t = torch.as_tensor(t).float()
    return np.array([t.mean().item(), t.std().item(), t.abs().max().item()])
