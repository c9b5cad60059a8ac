"""Boundary-loss pieces of the reference's training loop as HIP kernels.

Reference: src/learning/utils.py:198-222 (get_dilated), :283-285 (y_b), :318-324 (loss = CE + focal on the boundary
head), src/learning/focal_loss.py:7-44 (FocalCELoss).  No autograd graph: `focal_ce` returns the loss and, on request,
dL/dlogits -- the train step feeds that into the engine's tape.
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch

from .. import engine as E
from .._lib import check, lib

Tensor = torch.Tensor


def boundary_target(y: Tensor) -> Tensor:
    """y_b = where(get_dilated(y, K, connectivity=4).sum(1) > 1, 1, 0) (utils.py:283-285): 1 on pixels with a 4-neighbour
    of another class (image borders: zero padding, i.e. outside pixels never count), computed straight from the label map
    -- no one-hot tensor, no depthwise convolution."""
    if not y.is_cuda:
        raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback)")
    y = y.to(torch.int64).contiguous()
    B, H, W = y.shape
    yb = torch.empty_like(y)
    check(lib().c2s_boundary_target(y.data_ptr(), yb.data_ptr(), B, H, W, E._stream()), "boundary_target")
    return yb


def focal_ce(logits: Tensor, target: Tensor, gamma: float = 1.0, ignore_index: int = -100, want_grad: bool = False,
             ws: Optional[E.Workspace] = None, loss_out: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor]]:
    """FocalCELoss(gamma, size_average=True, weight=None)(logits [B,K,H,W], target [B,H,W]) (focal_loss.py:19-44).
    Returns (loss[1], dlogits | None).  With `loss_out` the value is ADDED to that tensor (utils.py:324)."""
    if not logits.is_cuda:
        raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback)")
    logits = logits.contiguous()
    target = target.to(torch.int64).contiguous()
    B, K = logits.shape[:2]
    HW = logits[0, 0].numel()
    ws = ws or E.Workspace(logits.device)
    w = ws.get("focal", lib().c2s_focal_ce_workspace_floats())
    loss = loss_out if loss_out is not None else torch.empty(1, device=logits.device, dtype=torch.float32)
    gl = torch.empty_like(logits) if want_grad else None
    check(lib().c2s_focal_ce(logits.data_ptr(), target.data_ptr(), loss.data_ptr(), gl.data_ptr() if want_grad else None, B, K,
                             HW, float(gamma), int(ignore_index), 1 if loss_out is not None else 0, w.data_ptr(), w.numel(),
                             E._stream()), "focal_ce")
    return loss, gl


class FocalCELoss:
    """Call-compatible with the reference's module for the configuration iterate() uses (gamma=2.0, no class weights,
    mean reduction); the value carries no autograd graph -- see TrainStep(add_boundary_loss=True) for training."""

    def __init__(self, gamma: float = 1.0, size_average: bool = True, ignore_index: int = -100, weight=None):
        if weight is not None or not size_average:
            raise NotImplementedError("FocalCELoss: class weights / sum reduction are not built")
        self.gamma, self.ignore_index = gamma, ignore_index

    def __call__(self, preds: Tensor, target: Tensor) -> Tensor:
        return focal_ce(preds, target, self.gamma, self.ignore_index)[0][0]
