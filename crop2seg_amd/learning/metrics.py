"""Metrics tail of the reference's iterate() loop on the device.

Reference: src/learning/utils.py:332-336,377-385 (argmax, top-2 rule, meters) and src/learning/miou.py:24-230
(ConfusionMatrix / IoU).  `IoU` keeps the reference's class interface (`add`, `value`, `get_miou_acc`, `reset`) with the
confusion matrix as an int64 [K,K] device tensor filled by one HIP kernel -- logits are read once, nothing synchronises
until the host asks for a value.  `StepMeters` is the fused form of the three meters of iterate(): one pass over the
logits feeds the top-1 and top-2 confusion matrices, and the running loss is summed on the device instead of
`loss.item()` every step.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from .._lib import check, lib

Tensor = torch.Tensor


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _require_hip(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(f"crop2seg_amd metrics run on MI355X only (no CPU fallback): {what} must be a 'cuda' tensor")


def _miou_acc(conf: np.ndarray, ignore_index) -> Tuple[float, float]:
    """IoU.get_miou_acc (miou.py:213-230) on a host copy of the matrix."""
    conf = conf.copy()
    if ignore_index is not None:
        conf[:, ignore_index] = 0
        conf[ignore_index, :] = 0
    tp = np.diag(conf)
    fp = np.sum(conf, 0) - tp
    fn = np.sum(conf, 1) - tp
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (tp + fp + fn)
    return float(np.nanmean(iou) * 100), float(np.diag(conf).sum() / conf.sum() * 100)


class ConfusionMatrix:
    """miou.py:24-128 with the matrix held on the device (rows = target, columns = prediction)."""

    def __init__(self, num_classes: int, normalized: bool = False, device="cuda", lazy: bool = True):
        if not 1 <= num_classes <= 32:
            raise ValueError("1 <= num_classes <= 32")
        self.num_classes, self.normalized, self.device, self.lazy = num_classes, normalized, torch.device(device), lazy
        self.conf = torch.zeros(num_classes, num_classes, dtype=torch.int64, device=self.device)

    def reset(self) -> None:
        self.conf.zero_()

    def add(self, predicted: Tensor, target: Tensor) -> None:
        """predicted: [N,K] scores (arg-maxed by the kernel) or [N] class indices; target: [N] class indices."""
        _require_hip(predicted, "predicted")
        K = self.num_classes
        target = target.to(torch.int64).contiguous().view(-1)
        if predicted.dim() != 1:
            assert predicted.shape[1] == K, "number of predictions does not match size of confusion matrix"
            scores = predicted.to(torch.float32).t().contiguous()          # [K, N]: class-major like NCHW logits
            check(lib().c2s_metrics_update(scores.data_ptr(), target.data_ptr(), self.conf.data_ptr(), None, None, None, 1, K,
                                           target.numel(), _stream()), "metrics_update")
            return
        predicted = predicted.to(torch.int64).contiguous()
        assert predicted.shape[0] == target.shape[0], "number of targets and predicted outputs do not match"
        check(lib().c2s_confusion_add(predicted.data_ptr(), target.data_ptr(), self.conf.data_ptr(), target.numel(), K, _stream()),
              "confusion_add")

    def value(self):
        conf = self.conf.cpu().numpy()
        if self.normalized:
            conf = conf.astype(np.float32)
            return conf / conf.sum(1).clip(min=1e-12)[:, None]
        return conf


class IoU:
    """miou.py:131-230: per-class IoU / mIoU / accuracy from the accumulated confusion matrix."""

    def __init__(self, num_classes: int, normalized: bool = False, ignore_index=None, cm_device="cuda", lazy: bool = True):
        self.conf_metric = ConfusionMatrix(num_classes, normalized, device=cm_device, lazy=lazy)
        self.lazy = lazy
        if ignore_index is None:
            self.ignore_index = None
        elif isinstance(ignore_index, int):
            self.ignore_index = (ignore_index,)
        else:
            try:
                self.ignore_index = tuple(ignore_index)
            except TypeError:
                raise ValueError("'ignore_index' must be an int or iterable")

    def reset(self) -> None:
        self.conf_metric.reset()

    def add(self, predicted: Tensor, target: Tensor) -> None:
        """predicted: (N,K,H,W) scores or (N,H,W) class indices; target: (N,H,W) indices (or (N,K,H,W) one-hot scores)."""
        assert predicted.size(0) == target.size(0), "number of targets and predicted outputs do not match"
        assert predicted.dim() in (3, 4), "predictions must be of dimension (N, H, W) or (N, K, H, W)"
        assert target.dim() in (3, 4), "targets must be of dimension (N, H, W) or (N, K, H, W)"
        _require_hip(predicted, "predicted")
        if target.dim() == 4:
            _, target = target.max(1)
        target = target.to(torch.int64).contiguous()
        cm = self.conf_metric
        if predicted.dim() == 4:
            B, K = predicted.shape[:2]
            assert K == cm.num_classes
            logits = predicted.to(torch.float32).contiguous()
            check(lib().c2s_metrics_update(logits.data_ptr(), target.data_ptr(), cm.conf.data_ptr(), None, None, None, B, K,
                                           logits[0, 0].numel(), _stream()), "metrics_update")
        else:
            cm.add(predicted.reshape(-1), target.view(-1))

    def value(self):
        conf = self.conf_metric.value()
        if self.ignore_index is not None:
            conf[:, self.ignore_index] = 0
            conf[self.ignore_index, :] = 0
        tp = np.diag(conf)
        fp = np.sum(conf, 0) - tp
        fn = np.sum(conf, 1) - tp
        with np.errstate(divide="ignore", invalid="ignore"):
            iou = tp / (tp + fp + fn)
        return iou, np.nanmean(iou)

    def get_miou_acc(self) -> Tuple[float, float]:
        return _miou_acc(self.conf_metric.conf.cpu().numpy(), self.ignore_index)


class StepMeters:
    """The three meters of iterate() (utils.py:247-259,377-380) fed by ONE kernel per step:

        pred      = out.argmax(1)                                   -> iou_meter
        pred_top2 = where(y == second, second, pred)                -> iou_meter_top2
        loss_meter.add(loss.item())                                 -> device-side (sum, count), no host sync

    `update` only enqueues work on the current stream; `get_miou_acc`, `get_miou_acc_top2` and `loss_mean` synchronise
    (the reference reads them every display_step iterations).  torch.topk leaves the order of tied logits unspecified;
    ties resolve to the lower class index here, so the second class is the reference's wherever the three largest logits
    of a pixel are distinct."""

    def __init__(self, num_classes: int, ignore_index=None, device="cuda", add_boundary_loss: bool = False,
                 test_region: str = "all"):
        """add_boundary_loss: a third meter for the 2-class boundary head (utils.py:258-259,383-384).
        test_region: 'all' | 'boundary' | 'interior' (utils.py:362-373): the other region's pixels are relabelled to the
        ignore label [0..K)[ignore_index] before the meters see them (needs an int ignore_index, as in the reference)."""
        assert test_region in ("all", "boundary", "interior"), test_region
        self.iou = IoU(num_classes, ignore_index=ignore_index, cm_device=device)
        self.iou_top2 = IoU(num_classes, ignore_index=ignore_index, cm_device=device)
        self.iou_boundary = IoU(2, cm_device=device) if add_boundary_loss else None      # IoU(num_classes=2), utils.py:259
        self.loss_acc = torch.zeros(2, dtype=torch.float64, device=device)
        self.num_classes = num_classes
        self.test_region = test_region
        if test_region != "all":
            if not isinstance(ignore_index, int):
                raise ValueError("test_region needs an int ignore_index (the reference indexes range(num_classes) with it)")
            self.ignore_label = list(range(num_classes))[ignore_index]
        self._watched = []

    def watch(self, step) -> "StepMeters":
        """Register a TrainStep (anything with `check_health()`): the host-synchronising readers below then also surface a
        failed one-pass normalisation wait of that step's workspace (they raise; see TrainStep.check_health)."""
        self._watched.append(step)
        return self

    def _health(self) -> None:
        for s in self._watched:
            s.check_health()

    def reset(self) -> None:
        self.iou.reset()
        self.iou_top2.reset()
        if self.iou_boundary is not None:
            self.iou_boundary.reset()
        self.loss_acc.zero_()

    def region_target(self, y: Tensor) -> Tensor:
        """y as the meters see it under `test_region` (utils.py:362-373)."""
        if self.test_region == "all":
            return y
        y = y.to(torch.int64).contiguous()
        B, H, W = y.shape
        out = torch.empty_like(y)
        check(lib().c2s_region_relabel(y.data_ptr(), out.data_ptr(), B, H, W, 1 if self.test_region == "boundary" else 0,
                                       int(self.ignore_label), _stream()), "region_relabel")
        return out

    def update_boundary(self, out_b: Tensor, y_b: Tensor) -> None:
        """iou_meter_boundary.add(out_b.argmax(1), y_b) (utils.py:342,383-384); the arg-max runs inside the kernel."""
        assert self.iou_boundary is not None, "StepMeters(add_boundary_loss=True)"
        self.iou_boundary.add(out_b, y_b)

    def get_miou_acc_boundary(self) -> Tuple[float, float]:
        return self.iou_boundary.get_miou_acc()

    def update(self, out: Tensor, y: Tensor, loss: Optional[Tensor] = None, want_pred: bool = False):
        """out [B,K,H,W] f32 logits, y [B,H,W] int64, loss: 1-element device tensor.  Returns (pred, pred_top2) int64
        [B,H,W] when want_pred, else None."""
        _require_hip(out, "out")
        B, K = out.shape[:2]
        assert K == self.num_classes and out.dtype == torch.float32
        out = out.contiguous()
        y = self.region_target(y.to(torch.int64).contiguous())
        HW = out[0, 0].numel()
        pred = torch.empty_like(y) if want_pred else None
        pred2 = torch.empty_like(y) if want_pred else None
        check(lib().c2s_metrics_update(out.data_ptr(), y.data_ptr(), self.iou.conf_metric.conf.data_ptr(),
                                       self.iou_top2.conf_metric.conf.data_ptr(), pred.data_ptr() if want_pred else None,
                                       pred2.data_ptr() if want_pred else None, B, K, HW, _stream()), "metrics_update")
        if loss is not None:
            _require_hip(loss, "loss")
            lf = loss.detach().to(torch.float32).reshape(-1)
            check(lib().c2s_loss_meter_add(lf.data_ptr(), self.loss_acc.data_ptr(), _stream()), "loss_meter_add")
        return (pred, pred2) if want_pred else None

    def get_miou_acc(self) -> Tuple[float, float]:
        self._health()
        return self.iou.get_miou_acc()

    def get_miou_acc_top2(self) -> Tuple[float, float]:
        return self.iou_top2.get_miou_acc()

    def loss_mean(self) -> float:
        self._health()
        s, n = self.loss_acc.tolist()
        return s / n if n else float("nan")
