"""Tiled inference: the prediction loop of the reference's web app on the device.

Reference: src/webapp/prediction.py:253-355 -- a Sentinel-2 tile is cut into 10 x 10 patches of 128 x 128; the loop runs
the model on ONE patch at a time (batch_size 1), moves the logits to the host, applies Softmax(dim=1), keeps the top-1
class, stacks the 100 results, re-tiles them with einops '(h w) ... h1 w1 -> ... (h h1) (w w1)' and crops to
1098 x 1098.

Here the patches of a tile go through the backbone in batches (eval mode: BatchNorm uses running statistics, so a patch's
logits do not depend on its batch -- bit for bit, tests/test_configs_gpu.py), and one kernel (c2s_softmax_stitch) turns a
batch of logits into its part of the tile rasters: softmax, top-1, tiling and crop in a single pass, nothing leaves HBM
until the caller asks for the rasters.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from . import engine as E
from ._lib import check, lib

Tensor = torch.Tensor


@torch.no_grad()
def predict_tile(model, x: Tensor, dates: Tensor, grid: int = 10, crop: int = 1098, batch_size: int = 10) -> Tuple[Tensor, Tensor]:
    """x [grid*grid, T, C, h1, w1] (patches in row-major tile order, as the reference's test loader yields them), dates
    [grid*grid, T] -> (proba [K, crop, crop] f32, top1 [crop, crop] int64) on the device."""
    if not x.is_cuda:
        raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback)")
    n = x.shape[0]
    assert n == grid * grid and dates.shape[0] == n
    h1, w1 = x.shape[-2:]
    out_h, out_w = min(crop, grid * h1), min(crop, grid * w1)
    was_training = model.training
    model.eval()
    proba: Optional[Tensor] = None
    top1: Optional[Tensor] = None
    try:
        for first in range(0, n, batch_size):
            xb = x[first:first + batch_size].contiguous()
            db = dates[first:first + batch_size].contiguous()
            logits = model(xb, batch_positions=db)
            if isinstance(logits, tuple):
                logits = logits[0]
            K = logits.shape[1]
            if proba is None:
                proba = torch.empty(K, out_h, out_w, device=x.device, dtype=torch.float32)
                top1 = torch.empty(out_h, out_w, device=x.device, dtype=torch.int64)
            check(lib().c2s_softmax_stitch(logits.data_ptr(), proba.data_ptr(), top1.data_ptr(), first, logits.shape[0], K, h1, w1,
                                           grid, out_h, out_w, E._stream()), "softmax_stitch")
    finally:
        model.train(was_training)
    model.check_health()                 # one host synchronisation per tile: a failed normalisation wait must not reach a map
    return proba, top1
