"""Seeded synthetic weights / inputs shared by the golden generator and the tests.

TEST INFRASTRUCTURE ONLY (see oracle/crop2seg_oracle.py header).

Weights are regenerated from (key list, shapes, seed, flavour) instead of being stored in the
fixtures (a U-TAE state_dict is 4.3 MB); each fixture keeps a per-tensor checksum so RNG drift
between torch builds is detected instead of silently producing a mismatch.

Flavours:
  "wi"   - same distributions as the reference's ``weight_init`` (src/learning/weight_init.py:4-46):
           Conv2d/ConvTranspose2d/Linear xavier-normal weight + N(0,1) bias, Conv1d N(0,1) both,
           BatchNorm weight N(0,1) / bias 0, GroupNorm and Q left at their constructor init.
  "tame" - well-conditioned: norm gains 1+0.25*N(0,1), small biases.
BatchNorm running stats are randomised in both (mean 0.2*N, var U(0.5,1.5)) so eval mode
exercises them.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import torch

KeyShapes = List[Tuple[str, Tuple[int, ...]]]


def _is_bn(key: str, names: set) -> bool:
    return key.rsplit(".", 1)[0] + ".running_mean" in names


def make_state(key_shapes: KeyShapes, seed: int, flavour: str = "wi", dtype=torch.float32) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    names = {k for k, _ in key_shapes}
    sd: Dict[str, torch.Tensor] = {}
    for key, shape in key_shapes:
        shape = tuple(shape)
        leaf = key.rsplit(".", 1)[1]
        if leaf == "num_batches_tracked":
            sd[key] = torch.zeros((), dtype=torch.int64)
            continue
        if leaf == "running_mean":
            t = 0.2 * torch.randn(shape, generator=g)
        elif leaf == "running_var":
            t = 0.5 + torch.rand(shape, generator=g)
        elif leaf == "Q":
            t = torch.randn(shape, generator=g) * math.sqrt(2.0 / shape[-1])
        elif len(shape) == 1 and leaf == "weight":            # norm gain
            if flavour == "wi" and _is_bn(key, names):
                t = torch.randn(shape, generator=g)
            elif flavour == "wi":
                t = torch.ones(shape)
            else:
                t = 1.0 + 0.25 * torch.randn(shape, generator=g)
        elif len(shape) == 1:                                   # bias
            prefix = key.rsplit(".", 1)[0]
            is_norm = (prefix + ".weight") in names and len(dict(key_shapes)[prefix + ".weight"]) == 1
            if is_norm:
                t = torch.zeros(shape) if flavour == "wi" else 0.1 * torch.randn(shape, generator=g)
            else:
                t = torch.randn(shape, generator=g) * (1.0 if flavour == "wi" else 0.1)
        elif "inconv" in key and flavour == "wi":               # Conv1d: N(0,1)
            t = torch.randn(shape, generator=g)
        else:                                                   # xavier-normal
            rf = 1
            for s in shape[2:]:
                rf *= s
            std = math.sqrt(2.0 / ((shape[0] + shape[1]) * rf))
            t = torch.randn(shape, generator=g) * std
        sd[key] = t.to(dtype)
    return sd


def checksum(sd: Dict[str, torch.Tensor]) -> torch.Tensor:
    """[n_tensors, 2] (sum, abs-sum) in float64, key order."""
    rows = [[float(v.double().sum()), float(v.double().abs().sum())] for v in sd.values()]
    return torch.tensor(rows, dtype=torch.float64)


def make_inputs(B: int, T: int, C: int, H: int, W: int, seed: int, lengths: Sequence[int] | None = None,
                n_classes: int = 15) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """Synthetic batch in the shape pad_collate produces (reference src/utils.py:14-32):
    x ~ N(0,1) f32 [B,T,C,H,W], dates = 5*t (+b) int64, frames t >= lengths[b] are exactly 0
    in x and dates; y ~ U{0..n_classes-1} int64 [B,H,W]."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, C, H, W, generator=g)
    dates = (5 * torch.arange(T)[None, :] + torch.arange(B)[:, None]).to(torch.int64)
    if lengths is not None:
        for b, L in enumerate(lengths):
            x[b, L:] = 0
            dates[b, L:] = 0
    y = torch.randint(0, n_classes, (B, H, W), generator=g)
    return x, dates, y


def dates_for(dates: torch.Tensor, ctor: dict) -> torch.Tensor:
    """batch_positions as the positional-encoder flags of a fixture expect them: use_abs_rel_enc takes [B,T,2] = (relative
    date, day of year); the day of year is a fixed function of the relative date here (padded frames keep 0)."""
    if ctor and ctor.get("use_abs_rel_enc"):
        doy = torch.where(dates > 0, (3 * dates + 17) % 365, torch.zeros_like(dates))
        return torch.stack([dates, doy], dim=-1)
    return dates
