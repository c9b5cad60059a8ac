#!/usr/bin/env python
"""Diagnostic (GPU box): which torch ops inside one U-TAE train step launch device copies / fills (torch profiler, CPU side)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import crop2seg_amd as C2S  # noqa: E402
from crop2seg_amd.learning.utils import TrainStep, default_config, get_model  # noqa: E402
from crop2seg_amd.learning.synthetic import synthetic_batch  # noqa: E402

dev = torch.device("cuda")
torch.manual_seed(1)
net = get_model(default_config("utae")).to(dev)
net.apply(C2S.weight_init)
net.train()
step = TrainStep(net, num_classes=15)
x, dates, y, lengths = synthetic_batch(4, 32, 128, 128, 1, dev, irregular=False)
for _ in range(3):
    step(x, dates, y)
torch.cuda.synchronize()
from torch.profiler import ProfilerActivity, profile  # noqa: E402
with profile(activities=[ProfilerActivity.CPU], with_stack=True) as prof:
    step(x, dates, y)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages(group_by_stack_n=6) if e.key in ("aten::copy_", "aten::clone", "aten::fill_", "aten::zero_", "aten::_to_copy",
                                                                       "aten::contiguous", "aten::cat", "aten::zeros_like", "aten::index", "aten::uniform_",
                                                                       "aten::random_", "aten::bernoulli_", "aten::rand", "aten::randint")]
rows.sort(key=lambda e: -e.count)
for e in rows[:40]:
    st = [s for s in e.stack if "crop2seg_amd" in s or "bench" in s][:3]
    print(f"{e.key:18s} x{e.count:3d}  {' <- '.join(s.strip().split('/')[-1] for s in st)}")
