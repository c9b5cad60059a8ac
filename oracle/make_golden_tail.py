"""Generate tests/golden/tail_*.npz from the IMPORTED reference modules that import in the build container:
src/learning/miou.py (ConfusionMatrix / IoU) and src/learning/focal_loss.py (FocalCELoss).

    PYTHONPATH=/root/reference python oracle/make_golden_tail.py

TEST INFRASTRUCTURE ONLY; the reference never travels to the GPU box -- only these small input/output vectors do.
(src/learning/utils.py, src/utils.py, the dataset and the web app do not import here: torchnet / torchvision / rasterio
are absent -- see oracle/tail_oracle.py for what that leaves unpinned.)
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from src.learning.miou import IoU  # noqa: E402  (reference)
from src.learning.focal_loss import FocalCELoss  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def miou_case(name, seed, K, B, H, ignore_index, n_batches, zero_frac):
    g = torch.Generator().manual_seed(seed)
    meter = IoU(num_classes=K, ignore_index=ignore_index, cm_device="cpu")
    logits_all, y_all = [], []
    for _ in range(n_batches):
        # BN + ReLU head: logits >= 0 with a large share of exact zeros (utae.py:191) -> argmax ties
        logits = torch.relu(torch.randn(B, K, H, H, generator=g))
        logits = logits * (torch.rand(B, K, H, H, generator=g) >= zero_frac)
        y = torch.randint(0, K, (B, H, H), generator=g)
        meter.add(logits.argmax(dim=1), y)                 # iterate(): pred = out.argmax(dim=1); iou_meter.add(pred, y)
        logits_all.append(logits)
        y_all.append(y)
    conf = meter.conf_metric.value().copy()
    miou, acc = meter.get_miou_acc()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), logits=torch.stack(logits_all).numpy(), y=torch.stack(y_all).numpy(),
                        conf=conf.astype(np.int64), miou=np.float64(miou), acc=np.float64(acc), K=np.int64(K),
                        ignore_index=np.int64(ignore_index))
    print(name, "miou", miou, "acc", acc, "conf sum", int(conf.sum()))


def focal_case(name, seed, B, K, H, gamma, ignore_frac, weighted=False, size_average=True):
    g = torch.Generator().manual_seed(seed)
    logits = (2 * torch.randn(B, K, H, H, generator=g)).requires_grad_(True)
    y = torch.randint(0, K, (B, H, H), generator=g)
    if ignore_frac > 0:
        y[torch.rand(B, H, H, generator=g) < ignore_frac] = -100
    extra = {}
    if weighted or not size_average:               # round 4: the module's other constructor arguments (focal_loss.py:12-45)
        w = (torch.rand(K, generator=g) + 0.5) if weighted else None
        loss = FocalCELoss(gamma=gamma, size_average=size_average, weight=w)(logits, y)
        extra = dict(size_average=np.int64(size_average), weight=(w.numpy() if weighted else np.zeros(0, np.float32)))
    else:
        loss = FocalCELoss(gamma=gamma)(logits, y)
    loss.backward()
    np.savez_compressed(os.path.join(OUT, name + ".npz"), logits=logits.detach().numpy(), y=y.numpy(),
                        loss=np.float64(loss.item()), grad=logits.grad.numpy(), gamma=np.float64(gamma), **extra)
    print(name, "loss", loss.item())


def round4():
    focal_case("tail_focal_g2_weighted", 203, 2, 5, 16, 2.0, 0.15, weighted=True)             # [N,1] x [N] broadcast of the reference
    focal_case("tail_focal_g1_sum", 204, 2, 4, 12, 1.0, 0.1, size_average=False)
    focal_case("tail_focal_g2_weighted_sum", 205, 1, 3, 12, 2.0, 0.0, weighted=True, size_average=False)


def main():
    os.makedirs(OUT, exist_ok=True)
    if "--round4" in sys.argv:
        return round4()
    miou_case("tail_miou_k15", 101, 15, 3, 32, -1, 3, 0.4)
    miou_case("tail_miou_k4", 103, 4, 2, 16, -1, 2, 0.7)
    focal_case("tail_focal_g2", 201, 3, 2, 32, 2.0, 0.0)          # iterate(): FocalCELoss(gamma=2.0) on the 2-class boundary head
    focal_case("tail_focal_g1_ignore", 202, 2, 5, 16, 1.0, 0.2)


if __name__ == "__main__":
    main()
