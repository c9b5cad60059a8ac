"""Boundary-loss pieces of the reference's training loop as HIP kernels.

Reference: src/learning/utils.py:198-222 (get_dilated), :283-285 (y_b), :318-324 (loss = CE + focal on the boundary
head), src/learning/focal_loss.py:7-44 (FocalCELoss), src/learning/smooth_loss.py:18-84 (SmoothCrossEntropy2D).  No autograd graph: `focal_ce` returns the loss and, on request,
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
             ws: Optional[E.Workspace] = None, loss_out: Optional[Tensor] = None, class_w: Optional[Tensor] = None,
             size_average: bool = True) -> Tuple[Tensor, Optional[Tensor]]:
    """FocalCELoss(gamma, size_average, ignore_index, weight)(logits [B,K,H,W], target [B,H,W]) (focal_loss.py:12-45).
    Returns (loss[1], dlogits | None).  With `loss_out` the value is ADDED to that tensor (utils.py:324).  With class weights the
    value is the reference's: mean(w[target]) * mean(focal terms) (sums for size_average=False) -- its [N,1] x [N] broadcast, see
    csrc/metrics.hip."""
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
    if class_w is not None:
        class_w = class_w.detach().to(logits.device, torch.float32).contiguous()
        if class_w.numel() != K:
            raise ValueError(f"FocalCELoss: {class_w.numel()} class weights for {K} classes")
    check(lib().c2s_focal_ce_ex(logits.data_ptr(), target.data_ptr(), E._ptr(class_w), loss.data_ptr(),
                                gl.data_ptr() if want_grad else None, B, K, HW, float(gamma), int(ignore_index),
                                1 if size_average else 0, 1 if loss_out is not None else 0, w.data_ptr(), w.numel(),
                                E._stream()), "focal_ce")
    return loss, gl


class FocalCELoss:
    """Call-compatible with the reference's module (focal_loss.py:7-45: gamma, size_average, ignore_index, weight); the value
    carries no autograd graph -- `grad()` returns dL/dlogits of the last call with want_grad=True, and
    TrainStep(add_boundary_loss=True) is the training route."""

    def __init__(self, gamma: float = 1.0, size_average: bool = True, ignore_index: int = -100, weight: Optional[Tensor] = None):
        self.gamma, self.size_average, self.ignore_index, self.weight = gamma, bool(size_average), ignore_index, weight
        self._grad: Optional[Tensor] = None

    def __call__(self, preds: Tensor, target: Tensor, want_grad: bool = False) -> Tensor:
        if preds.dim() == 2:                        # (N, C) logits with (N,) targets
            preds, target = preds.t().reshape(1, preds.shape[1], preds.shape[0], 1), target.reshape(1, -1, 1)
        loss, self._grad = focal_ce(preds, target, self.gamma, self.ignore_index, want_grad, class_w=self.weight,
                                    size_average=self.size_average)
        return loss[0]

    forward = __call__

    def grad(self) -> Optional[Tensor]:
        return self._grad


DEFAULT_CLASS_PROPORTIONS = (0.3111, 0.0193, 0.0809, 0.2809, 0.1084, 0.0892, 0.0350, 0.0170, 0.0007,
                             0.0047, 0.0015, 0.0044, 0.0394, 0.0074)          # smooth_loss.py:28-29 (S2TSCZCrop)


def smooth_ce(logits: Tensor, target: Tensor, label_smoothing: float = 0.1, class_w: Optional[Tensor] = None,
              bg_distrib: Optional[Tensor] = None, bg_index: int = 0, want_grad: bool = False,
              ws: Optional[E.Workspace] = None, loss_out: Optional[Tensor] = None, reduction: str = "mean",
              pixel_loss: Optional[Tensor] = None) -> Tuple[Tensor, Optional[Tensor], Tensor]:
    """One pass of c2s_smooth_ce_ex: returns (loss[1], dlogits | None, counters[2]) -- counters[1] is the number of labels
    outside [0, K) the pass met (a device value: reading it synchronises).  reduction 'mean' | 'sum' | 'none' (the per-pixel
    terms go to `pixel_loss` [B,H,W]; loss[0] is then their sum and dlogits the gradient of that sum)."""
    if not logits.is_cuda:
        raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback)")
    logits = logits.contiguous()
    target = target.to(torch.int64).contiguous()
    B, K, H, W = logits.shape
    ws = ws or E.Workspace(logits.device)
    w = ws.get("smooth_ce", lib().c2s_smooth_ce_workspace_floats())
    loss = loss_out if loss_out is not None else torch.empty(1, device=logits.device, dtype=torch.float32)
    gl = torch.empty_like(logits) if want_grad else None
    red = {"mean": 0, "sum": 1, "none": 2}[reduction]
    if red == 2 and pixel_loss is None:
        raise ValueError("smooth_ce: reduction='none' needs a pixel_loss tensor")
    check(lib().c2s_smooth_ce_ex(logits.data_ptr(), target.data_ptr(), E._ptr(class_w), E._ptr(bg_distrib), loss.data_ptr(),
                                 E._ptr(gl), E._ptr(pixel_loss), B, K, H, W, float(label_smoothing), int(bg_index), red,
                                 1 if loss_out is not None else 0, w.data_ptr(), w.numel(), E._stream()), "smooth_ce")
    return loss, gl, w[w.numel() - 2:]


class SmoothCrossEntropy2D:
    """Call-compatible with the reference's criterion (smooth_loss.py:18-84; constructor arguments in the same order):
    label smoothing that follows the field borders -- classes present in the 4-neighbourhood of a pixel share the target
    mass -- with the fixed distribution for background pixels, then CE with probability targets (reduction 'mean' / 'sum' /
    'none', or the legacy size_average / reduce flags; with 'none' the call returns the [B,H,W] terms and grad() is the gradient of
    their sum).
    The value carries no autograd graph; `grad()` returns dL/dlogits of the last call with want_grad=True (TrainStep feeds
    it into the tape).  `check_targets()` raises if the last call met labels outside [0, K) -- the reference's one_hot
    raises there; the kernel skips and counts them instead of faulting."""

    def __init__(self, weight: Optional[Tensor] = None, size_average=None, ignore_index: int = -100, reduce=None,
                 reduction: str = "mean", label_smoothing: float = 0.1, background_treatment: bool = True,
                 background_index: int = 0, background_label_value: float = 0.6,
                 class_proportions=DEFAULT_CLASS_PROPORTIONS):
        if size_average is not None or reduce is not None:       # torch.nn._reduction.legacy_get_string
            size_average = True if size_average is None else size_average
            reduce = True if reduce is None else reduce
            reduction = "mean" if (size_average and reduce) else ("sum" if reduce else "none")
        if reduction not in ("mean", "sum", "none"):
            raise ValueError(f"{reduction} is not a valid value for reduction")
        self.reduction = reduction
        self.weight, self.ls = weight, float(label_smoothing)
        self.background_treatment, self.background_index = bool(background_treatment), int(background_index)
        bd = torch.tensor([background_label_value] + list(class_proportions), dtype=torch.float32)
        bd[1:] *= 1 - background_label_value                     # smooth_loss.py:78-80
        self._bg_host = bd
        self._bg_dev: Optional[Tensor] = None
        self._w_dev: Optional[Tensor] = None
        self._ws: Optional[E.Workspace] = None
        self._last = None

    def __call__(self, input: Tensor, target: Tensor, want_grad: bool = False) -> Tensor:
        assert input.dim() == 4, f"`input` is expected to have 4 dimensions (B x N_CLASSES x H x W) but is of shape {input.shape}"
        assert target.dim() == 3, f"`target` is expected to have 3 dimensions (B x H x W) but is of shape {target.shape}"
        bg = None
        if self.background_treatment:
            if self._bg_host.numel() != input.shape[1]:
                raise ValueError(f"background distribution has {self._bg_host.numel()} entries, the logits {input.shape[1]} classes")
            if self._bg_dev is None or self._bg_dev.device != input.device:
                self._bg_dev = self._bg_host.to(input.device)
            bg = self._bg_dev
        cw = None
        if self.weight is not None:
            if self._w_dev is None or self._w_dev.device != input.device:
                self._w_dev = self.weight.detach().to(input.device, torch.float32).contiguous()
            cw = self._w_dev
        if self._ws is None or self._ws.device != input.device:
            self._ws = E.Workspace(input.device)
        pl = torch.empty(target.shape, device=input.device, dtype=torch.float32) if self.reduction == "none" else None
        loss, gl, counters = smooth_ce(input, target, self.ls, cw, bg, self.background_index, want_grad, self._ws,
                                       reduction=self.reduction, pixel_loss=pl)
        self._last = (gl, counters)
        return pl if pl is not None else loss[0]

    forward = __call__

    def grad(self) -> Optional[Tensor]:
        return None if self._last is None else self._last[0]

    def check_targets(self) -> None:
        if self._last is not None and float(self._last[1][1]) != 0.0:
            raise ValueError(f"SmoothCrossEntropy2D: {int(float(self._last[1][1]))} target labels outside [0, n_classes)")
