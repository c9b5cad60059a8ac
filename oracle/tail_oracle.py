"""CPU restatements of the steps on either side of the backbone hot path.

TEST INFRASTRUCTURE ONLY (same rules as oracle/crop2seg_oracle.py: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg only; the product package never imports it).

Parity status per function (paths under the reference root):

  PINNED by fixtures generated from the imported reference (oracle/make_golden_tail.py -> tests/golden/tail_*.npz):
    confusion_matrix / miou_acc        src/learning/miou.py:55-117, 213-230  (ConfusionMatrix.add, IoU.get_miou_acc)
    focal_ce                           src/learning/focal_loss.py:7-44
  PARITY UNPINNED -- the defining modules do not import in this container (ordinary ModuleNotFoundError: torchnet /
  torchvision / rasterio / geopandas are absent) and the reference ships no fixtures for them; the restatements below call
  the same torch primitives the reference calls, line for line:
    metrics_tail (argmax / top-2 rule)  src/learning/utils.py:332-336, 377-380
    get_dilated / boundary_target       src/learning/utils.py:198-222, 283-285
    collate_series                      src/datasets/s2_ts_cz_crop.py:366-374, 393-398; src/utils.py:14-32; train.py:291
    softmax_stitch                      src/webapp/prediction.py:310-333
    smooth_targets                      src/learning/smooth_loss.py:58-84

  torch.topk does not define the order of tied values (its CPU and CUDA kernels disagree with each other), so the top-2
  rule is restated with the stable order (lowest class index first); it equals out.topk(2, dim=1).indices wherever the
  three largest logits of a pixel are distinct -- `top2_defined` returns that mask.
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F


# ------------------------------------------------------------------------------------------------ N2: metrics tail
def confusion_matrix(pred: np.ndarray, target: np.ndarray, num_classes: int) -> np.ndarray:
    """ConfusionMatrix.add (miou.py:98-112): bincount of pred + K * target -> [K,K], rows = target."""
    x = pred.astype(np.int64).reshape(-1) + num_classes * target.astype(np.int64).reshape(-1)
    return np.bincount(x, minlength=num_classes ** 2).reshape(num_classes, num_classes)


def miou_acc(conf: np.ndarray, ignore_index=None) -> Tuple[float, float]:
    """IoU.get_miou_acc (miou.py:213-230)."""
    conf = conf.copy()
    if ignore_index is not None:
        ign = (ignore_index,) if isinstance(ignore_index, int) else tuple(ignore_index)
        conf[:, ign] = 0
        conf[ign, :] = 0
    tp = np.diag(conf)
    fp = conf.sum(0) - tp
    fn = conf.sum(1) - tp
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = tp / (tp + fp + fn)
    return float(np.nanmean(iou) * 100), float(tp.sum() / conf.sum() * 100)


def stable_top2(logits: torch.Tensor) -> torch.Tensor:
    """[B,K,H,W] -> indices [B,2,H,W] of the two largest logits, ties resolved towards the lower class index."""
    order = torch.sort(logits, dim=1, descending=True, stable=True).indices
    return order[:, :2]


def top2_defined(logits: torch.Tensor) -> torch.Tensor:
    """[B,H,W] mask of pixels whose three largest logits are pairwise distinct (torch.topk is then unambiguous)."""
    v = torch.sort(logits, dim=1, descending=True).values
    ok = v[:, 0] != v[:, 1]
    if logits.shape[1] > 2:
        ok &= v[:, 1] != v[:, 2]
    return ok


def metrics_tail(logits: torch.Tensor, y: torch.Tensor, num_classes: int):
    """utils.py:332-336,377-380: returns (pred, pred_top2, conf, conf_top2)."""
    pred = logits.argmax(dim=1)
    pred_ = stable_top2(logits)
    pred_top2 = torch.where(y == pred_[:, 1], pred_[:, 1], pred_[:, 0])
    conf = confusion_matrix(pred.numpy(), y.numpy(), num_classes)
    conf2 = confusion_matrix(pred_top2.numpy(), y.numpy(), num_classes)
    return pred, pred_top2, conf, conf2


# ------------------------------------------------------------------------------------------------ N4: boundary loss
def get_dilated(target: torch.Tensor, n_classes: int, connectivity: int = 4) -> torch.Tensor:
    """utils.py:198-222."""
    if connectivity == 8:
        weights = torch.ones((n_classes, 1, 3, 3))
    else:
        weights = torch.tensor([[0., 1., 0.], [1., 1., 1.], [0., 1., 0.]]).view(1, 1, 3, 3).repeat(n_classes, 1, 1, 1)
    one_hot = F.one_hot(target.long(), num_classes=n_classes).permute(0, 3, 1, 2)
    return F.conv2d(one_hot.float(), weights, groups=n_classes, padding=(1, 1)).bool().long()


def boundary_target(y: torch.Tensor, n_classes: int) -> torch.Tensor:
    """utils.py:283-285: 0 background, 1 boundary."""
    return torch.where(get_dilated(y, n_classes, 4).sum(1) > 1, 1, 0)


def focal_ce(preds: torch.Tensor, target: torch.Tensor, gamma: float = 2.0, ignore_index: int = -100,
             weight: Optional[torch.Tensor] = None, size_average: bool = True) -> torch.Tensor:
    """FocalCELoss.forward (focal_loss.py:19-45).  With class weights the reference gathers them as an [N,1] column and
    multiplies by the [N] row of focal terms (focal_loss.py:36-39): an N x N outer product whose mean / sum is the product of
    the means / sums -- restated as that product (the fixtures tail_focal_*weighted* pin it to the imported module)."""
    target = target.reshape(-1, 1)
    if preds.ndim > 2:
        preds = preds.permute(0, 2, 3, 1).flatten(0, 2)
    keep = target[:, 0] != ignore_index
    preds, target = preds[keep, :], target[keep, :]
    logpt = F.log_softmax(preds, dim=1).gather(1, target).view(-1)
    pt = logpt.exp()
    loss = -1 * (1 - pt) ** gamma * logpt
    if weight is not None:
        w = weight[target[:, 0]]
        return w.mean() * loss.mean() if size_average else w.sum() * loss.sum()
    return loss.mean() if size_average else loss.sum()


def smooth_targets(target: torch.Tensor, n_classes: int, label_smoothing: float = 0.1) -> torch.Tensor:
    """SmoothCrossEntropy2D soft targets with background_treatment=False (smooth_loss.py:66-73), [B,K,H,W]."""
    dilated = get_dilated(target, n_classes, 4)
    eps = label_smoothing / n_classes
    exp_small = eps * (n_classes - dilated.sum(1))
    exp_large = (1 - exp_small) / dilated.sum(1)
    return torch.where(dilated.permute(1, 0, 2, 3) == 1, exp_large, eps).permute(1, 0, 2, 3)


def region_target(y: torch.Tensor, n_classes: int, test_region: str, ignore_index: int) -> torch.Tensor:
    """test_region relabelling of iterate() (learning/utils.py:362-373)."""
    if test_region == "all":
        return y
    dilated = get_dilated(y, n_classes, 4)
    ignore_label = [i for i in range(n_classes)][ignore_index]
    if test_region == "boundary":
        return torch.where(dilated.sum(1) == 1, ignore_label, y)
    return torch.where(dilated.sum(1) > 1, ignore_label, y)


DEFAULT_CLASS_PROPORTIONS = (0.3111, 0.0193, 0.0809, 0.2809, 0.1084, 0.0892, 0.0350, 0.0170, 0.0007,
                             0.0047, 0.0015, 0.0044, 0.0394, 0.0074)          # smooth_loss.py:28-29


def smooth_cross_entropy_2d(logits: torch.Tensor, target: torch.Tensor, weight: Optional[torch.Tensor] = None,
                            label_smoothing: float = 0.1, background_treatment: bool = True, background_index: int = 0,
                            background_label_value: float = 0.6, class_proportions=DEFAULT_CLASS_PROPORTIONS,
                            reduction: str = "mean") -> torch.Tensor:
    """SmoothCrossEntropy2D.forward (smooth_loss.py:58-84): dilation soft targets, the background distribution, then
    torch's own CrossEntropyLoss (the reference's superclass) with probability targets.  Parity unpinned: the module
    imports src.learning.utils -> torchnet, which this image lacks; the CE itself is torch's."""
    K = logits.shape[1]
    target_out = smooth_targets(target, K, label_smoothing)
    if background_treatment:
        bd = torch.tensor([background_label_value] + list(class_proportions), dtype=torch.float32)
        bd[1:] *= 1 - background_label_value
        target_out = torch.where(target[:, None, ...] == background_index, bd[:, None, None], target_out)
    return torch.nn.CrossEntropyLoss(weight=weight, reduction=reduction)(logits, target_out)


# ------------------------------------------------------------------------------------------------ N1: collate
CHANNELS_LIKE_PASTIS = [2, 1, 0, 4, 5, 6, 3, 7, 8, 9]          # s2_ts_cz_crop.py:248, train.py:291


def pad_tensor(x: torch.Tensor, l: int, pad_value=0) -> torch.Tensor:
    """src/utils.py:14-17."""
    padlen = l - x.shape[0]
    pad = [0 for _ in range(2 * len(x.shape[1:]))] + [0, padlen]
    return F.pad(x, pad=pad, value=pad_value)


def collate_series(series: Sequence[np.ndarray], dates: Sequence[np.ndarray], channels_order: Sequence[int],
                   mean: Optional[np.ndarray], std: Optional[np.ndarray], pad_value=0, add_ndvi: bool = False,
                   ndvi_bands=(6, 2)):
    """__getitem__ tail (astype(float32) -> channel re-order -> (d - mean) / std in fp32) for every series, then
    pad_collate of the (data, dates) pairs: returns x [B,T,C,H,W] f32, dates [B,T] int64."""
    xs, ds = [], []
    for a, d in zip(series, dates):
        t = torch.from_numpy(a.astype(np.float32))[:, list(channels_order), ...]
        if add_ndvi:                       # s2_ts_cz_crop.py:376-391: on the re-ordered RAW bands, before the normalisation
            na, nb = ndvi_bands
            ndvi = torch.where(t[:, na] + t[:, nb] == 0, 0., (t[:, na] - t[:, nb]) / (t[:, na] + t[:, nb]))
            ndvi = torch.where((ndvi < -1) | (ndvi > 1), 0, ndvi)
        if mean is not None:
            m = torch.from_numpy(np.asarray(mean)).float()
            s = torch.from_numpy(np.asarray(std)).float()
            t = (t - m[None, :, None, None]) / s[None, :, None, None]
        if add_ndvi:                       # :401-402: appended after the normalisation, itself not normalised
            t = torch.cat([t, ndvi[:, None]], dim=1)
        xs.append(t)
        ds.append(torch.from_numpy(np.asarray(d)).long())
    m_ = max(t.shape[0] for t in xs)
    x = torch.stack([pad_tensor(t, m_, pad_value) for t in xs], 0)
    dd = torch.stack([pad_tensor(d, m_, pad_value) for d in ds], 0)
    return x, dd


# ------------------------------------------------------------------------------------------------ N3: tiled inference
def softmax_stitch(patch_logits: Sequence[torch.Tensor], grid: int = 10, crop: int = 1098):
    """prediction.py:310-333: per patch (B = 1) Softmax(dim=1) and top-1, then '(h w) ... h1 w1 -> ... (h h1) (w w1)' and the
    crop.  patch_logits: grid*grid tensors [1,K,h1,w1].  Returns (proba [K,crop,crop] f32, t1 [crop,crop] int64)."""
    proba, t1 = [], []
    for out in patch_logits:
        pred_ = torch.nn.Softmax(dim=1)(out)
        proba.append(pred_)
        t1.append(pred_.max(dim=1)[1][0])
    t1 = torch.stack(t1)                       # [(h w), h1, w1]
    proba = torch.stack(proba)                 # [(h w), 1, K, h1, w1]
    n, h1, w1 = t1.shape
    t1 = t1.view(grid, grid, h1, w1).permute(0, 2, 1, 3).reshape(grid * h1, grid * w1)
    K = proba.shape[2]
    proba = proba.view(grid, grid, 1, K, h1, w1).permute(2, 3, 0, 4, 1, 5).reshape(1, K, grid * h1, grid * w1)[0]
    return proba[..., :crop, :crop], t1[..., :crop, :crop]
