"""Deterministic synthetic parameters and inputs, keyed by state_dict name.

TEST INFRASTRUCTURE ONLY.  The reference's released checkpoints are unreachable (no network) and
a 41 M-parameter state_dict cannot be committed, so parity tests synthesise every tensor from
``(seed, key name)``.  The same function loads the reference model (gen_golden.py), the CPU
oracle and the HIP model, so all three see bit-identical parameters regardless of module
construction order ("copy the state_dict rather than rely on RNG order").

Distributions keep activations O(1) through 70 layers and make eval-mode BN non-trivial:
  conv weight  ~ N(0, gain/fan_in)      BN weight ~ U(0.5,1.5)   BN bias ~ N(0,0.1²)
  running_mean ~ N(0,0.1²)              running_var ~ U(0.5,1.5)  conv bias ~ N(0,0.1²)

``conditioned=True`` (r04) — a WELL-CONDITIONED state for end-to-end checks with fixed bars.
The default state is a chaotic map: a random ReLU+BatchNorm chain of 70 layers amplifies a
perturbation by ~1.2x per layer (the mean that BatchNorm removes is half of a post-ReLU signal),
and gradients decorrelate through ReLU-mask flips (error ~ sqrt(flipped fraction) per layer):
rounding only the conv weights to bf16 moves the fp64 oracle's logits by L2-rel 0.5 and its
gradients by 1.0 (cosine ~0), CPU fp32 gradients sit 1.7e-2 from fp64.  Two changes, both on
BatchNorm affine parameters only (every tensor keeps its shape, BN statistics stay non-trivial):
  * gamma ~ U(0.1,0.2) on the LAST BatchNorm of every residual branch (xception
    `sep_conv3.block.bn_point` of blocks 1-20, xception.py:27-42; ResNet / HRNet `bn3` resp.
    BasicBlock `bn2`; MobileNetV2's projection BN) — each block is x + eps*F(x);
  * beta = +2*gamma*U(0.8,1.2) on every other BatchNorm — ~2 % of the units behind it are
    clipped by the ReLU instead of ~50 %: mask flips under perturbation become rare while every
    mask path still carries gradient.
Measured on C3 @65x129 (oracle/gen_golden_cond.py asserts it before writing fixtures): CPU
fp32-vs-fp64 logits 2e-6 / gradients 1e-4 (was 1.5e-4 / 1.7e-2); bf16-rounded weights: logits
4e-3, gradients 2e-2 (was 0.5 / 1.0).
"""
import re
import zlib

import torch


def _gen(seed, key):
    g = torch.Generator(device="cpu")
    g.manual_seed((zlib.crc32(key.encode()) ^ (seed * 0x9E3779B1)) & 0x7FFFFFFF)
    return g


BETA_FACTOR = 2.0
_BRANCH_LAST = re.compile(
    r"(encoder\.block([1-9]|1[0-9]|20)\.sep_conv3\.block\.bn_point"      # xception.py:27-42
    r"|\.(layer\d+|branches\.\d+)\.\d+\.bn3"                             # Bottleneck
    r"|encoder\.block\d+\.\d+\.conv\.\d+)\.(weight|bias)$")              # InvertedResidual


def branch_last_bn(key, all_keys=None):
    """Is `key` the affine parameter of the last BatchNorm of a residual branch?  BasicBlock's
    `bn2` qualifies only when the block has no `bn3` (needs the key set)."""
    if _BRANCH_LAST.search(key):
        return True
    m = re.search(r"^(.*\.(layer\d+|branches\.\d+)\.\d+)\.bn2\.(weight|bias)$", key)
    return bool(m and all_keys is not None and (m.group(1) + ".bn3.weight") not in all_keys)


def synth_tensor(key, shape, dtype=torch.float32, seed=0, conditioned=False, all_keys=None):
    g = _gen(seed, key)
    shape = tuple(shape)
    leaf = key.rsplit(".", 1)[-1]
    if conditioned and len(shape) == 1 and leaf in ("weight", "bias") \
            and all_keys is not None and (key[:-len(leaf)] + "running_mean") in all_keys:
        if branch_last_bn(key, all_keys):
            if leaf == "weight":
                return 0.1 + 0.1 * torch.rand(shape, generator=g)
        elif leaf == "bias":
            w = synth_tensor(key[:-4] + "weight", shape, dtype, seed)
            return BETA_FACTOR * w * (0.8 + 0.4 * torch.rand(shape, generator=g))
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


def synth_state_dict(keys_and_shapes, seed=0, conditioned=False):
    """keys_and_shapes: iterable of (key, shape) — e.g. from ``model.state_dict()``."""
    keys_and_shapes = list(keys_and_shapes)
    names = set(k for k, _ in keys_and_shapes) if conditioned else None
    return {k: synth_tensor(k, s, seed=seed, conditioned=conditioned, all_keys=names)
            for k, s in keys_and_shapes}


def synth_like(state_dict, seed=0, conditioned=False):
    return synth_state_dict([(k, tuple(v.shape)) for k, v in state_dict.items()], seed,
                            conditioned)


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
