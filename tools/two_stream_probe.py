#!/usr/bin/env python
"""Diagnostic (GPU box): does running two half-batches on two HIP streams overlap the MFMA-bound convolutions of one with
the HBM-bound norm passes of the other?  Two independent U-TAE train steps at B=2 on two streams vs one step at B=4.
(Not a product path: BatchNorm statistics of the decoder would be per half-batch.)"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import crop2seg_amd as C2S  # noqa: E402
from bench import synthetic_batch  # noqa: E402
from crop2seg_amd.learning.utils import TrainStep, default_config, get_model  # noqa: E402

dev = torch.device("cuda")


def make(B):
    torch.manual_seed(1)
    net = get_model(default_config("utae")).to(dev)
    net.apply(C2S.weight_init)
    net.train()
    return TrainStep(net, num_classes=15, distributed=False), synthetic_batch(B, 32, 128, 128, 1, dev)


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


s4, b4 = make(4)
print(f"one stream, B=4: {timeit(lambda: s4(*b4)):.2f} ms/step")
del s4
sa, ba = make(2)
sb, bb = make(2)
print(f"one stream, B=2: {timeit(lambda: sa(*ba)):.2f} ms/step")
st = [torch.cuda.Stream(), torch.cuda.Stream()]


def both():
    with torch.cuda.stream(st[0]):
        sa(*ba)
    with torch.cuda.stream(st[1]):
        sb(*bb)


ms = timeit(both)
print(f"two streams, B=2 + B=2: {ms:.2f} ms per pair = {4 / ms * 1e3:.1f} patches/s")
