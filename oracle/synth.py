"""Deterministic synthetic parameters and inputs, keyed by state_dict name.

TEST INFRASTRUCTURE ONLY.  The reference's released checkpoints are unreachable (no network) and
a 41 M-parameter state_dict cannot be committed, so parity tests synthesise every tensor from
``(seed, key name)``.  The same function loads the reference model (gen_golden.py), the CPU
oracle and the HIP model, so all three see bit-identical parameters regardless of module
construction order ("copy the state_dict rather than rely on RNG order").

Distributions keep activations O(1) through 70 layers and make eval-mode BN non-trivial:
  conv weight  ~ N(0, gain/fan_in)      BN weight ~ U(0.5,1.5)   BN bias ~ N(0,0.1²)
  running_mean ~ N(0,0.1²)              running_var ~ U(0.5,1.5)  conv bias ~ N(0,0.1²)
"""
import zlib

import torch


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


def synth_tensor(key, shape, dtype=torch.float32, seed=0):
    g = _gen(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=torch.long)
    if leaf == "running_var":
        return torch.rand(shape, generator=g) + 0.5
    if leaf == "running_mean":
        return torch.randn(shape, generator=g) * 0.1
    if len(shape) == 4:  # conv weight [O, C/groups, kh, kw]
        fan_in = shape[1] * shape[2] * shape[3]
        return torch.randn(shape, generator=g) * (2.0 / fan_in) ** 0.5
    if len(shape) == 2:  # linear weight
        return torch.randn(shape, generator=g) * (1.0 / shape[1]) ** 0.5
    if leaf == "weight":  # BN gamma
        return torch.rand(shape, generator=g) + 0.5
    return torch.randn(shape, generator=g) * 0.1  # BN beta / conv bias


def synth_state_dict(keys_and_shapes, seed=0):
    """keys_and_shapes: iterable of (key, shape) — e.g. from ``model.state_dict()``."""
    return {k: synth_tensor(k, s, seed=seed) for k, s in keys_and_shapes}


def synth_like(state_dict, seed=0):
    return synth_state_dict([(k, tuple(v.shape)) for k, v in state_dict.items()], seed)


def synth_images(batch, height, width, seed=0):
    """`randn(B,3,H,W)` — Cityscapes after Normalize(mean .5, std .5)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(1000 + seed)
    return torch.randn(batch, 3, height, width, generator=g)


def synth_targets(batch, height, width, nclass=19, ignore_frac=0.05, seed=0):
    """int64 labels in [0,nclass) with ~5 % set to -1 (= cfg.DATASET.IGNORE_INDEX)."""
    g = torch.Generator(device="cpu")
    g.manual_seed(2000 + seed)
    t = torch.randint(0, nclass, (batch, height, width), generator=g)
    m = torch.rand(batch, height, width, generator=g) < ignore_frac
    t[m] = -1
    return t
