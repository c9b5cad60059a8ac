"""Synthetic batches in the shape the reference's collate function hands to the model.

`pad_collate` (reference src/utils.py:20-66 with pad_tensor :14-17) right-pads every series of a batch with zeros to
the longest one, in the data AND in the acquisition dates, so a batch of irregular series is `x [B,T,10,H,W]` with
frames `t >= T_b` exactly 0 and `dates[b, t >= T_b] = 0`.  The regular workload (BASELINE configs[1]) has T_b = T.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import torch


def series_lengths(B: int, T: int, seed: int, t_min: int = 27) -> List[int]:
    """T_b ~ U{t_min..T} with one series forced to T (reference README.md:91-92: 27..61 acquisitions per patch)."""
    g = torch.Generator().manual_seed(seed)
    lengths = torch.randint(min(t_min, T), T + 1, (B,), generator=g).tolist()
    lengths[int(torch.randint(0, B, (1,), generator=g))] = T
    return lengths


def synthetic_batch(B: int, T: int, H: int, W: int, seed: int, device, n_classes: int = 15, irregular: bool = False,
                    lengths: Optional[Sequence[int]] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, List[int]]:
    """x ~ N(0,1) f32 [B,T,10,H,W] (the dataset is z-scored, s2_ts_cz_crop.py:393-398); dates = 5*t int64 (5-day
    revisit); y ~ U{0..n_classes-1}; generator seed = 1 + rank.  irregular=True applies the pad_collate padding for
    series lengths T_b ~ U{27..T}.  Returns (x, dates, y, lengths)."""
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B, T, 10, H, W, generator=g)
    dates = (5 * torch.arange(T))[None, :].repeat(B, 1).to(torch.int64)
    y = torch.randint(0, n_classes, (B, H, W), generator=g)
    if lengths is None:
        lengths = series_lengths(B, T, seed) if irregular else [T] * B
    for b, tb in enumerate(lengths):
        x[b, tb:] = 0
        dates[b, tb:] = 0
    return x.to(device), dates.to(device), y.to(device), list(lengths)
