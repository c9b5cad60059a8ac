#!/usr/bin/env python
"""Micro-benchmark of the kernels either side of the hot path at the sizes the reference uses:
HIP-event time per call and algorithmic bytes / time against the 8 TB/s HBM peak.
Usage (GPU box): python tools/tail_bench.py [--reps 20]"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crop2seg_amd.inference import predict_tile  # noqa: E402,F401
from crop2seg_amd.learning.losses import boundary_target, focal_ce  # noqa: E402
from crop2seg_amd.learning.metrics import StepMeters  # noqa: E402
from crop2seg_amd.utils import CHANNELS_LIKE_PASTIS, SeriesCollator  # noqa: E402
from crop2seg_amd import _lib  # noqa: E402


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


def line(name, ms, nbytes, note=""):
    tb = nbytes / (ms * 1e-3) / 1e12
    print(f"{name:44s} {ms * 1e3:9.1f} us  {nbytes / 1e6:9.1f} MB  {tb:5.2f} TB/s ({100 * tb / 8:4.1f} % of 8 TB/s) {note}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda")
    rng = np.random.default_rng(0)
    K, H = 15, 128

    # N1: B = 8 int16 series of irregular length (BASELINE.json configs[2]) -> normalised, reordered, padded fp32 batch
    lengths = [27, 51, 31, 30, 30, 40, 48, 61]
    series = [rng.integers(0, 10000, size=(t, 10, H, H), dtype=np.int16) for t in lengths]
    dates = [np.sort(rng.integers(0, 365, size=t)).astype(np.int64) for t in lengths]
    mean = rng.uniform(500, 3000, 10).astype(np.float32)
    std = rng.uniform(300, 2000, 10).astype(np.float32)
    for mode in ("staged", "zero_copy"):
        col = SeriesCollator(CHANNELS_LIKE_PASTIS, mean, std, device="cuda", mode=mode)
        ms = timed(lambda: col(series, dates), max(3, a.reps // 4))
        raw = sum(lengths) * 10 * H * H * 2
        out = 8 * 61 * 10 * H * H * 4
        line(f"N1 SeriesCollator int16 B=8 T<=61 ({mode})", ms, raw + out,
             "(whole call: host staging memcpy + " + ("H2D copy + kernel)" if mode == "staged" else "kernel reading pinned memory)"))
    # the kernel alone, raw int16 series and the (offsets | dates) table already on the device
    import ctypes as C
    L = _lib.lib()
    total = sum(lengths)
    raw_dev = torch.from_numpy(np.concatenate(series, 0)).to(dev)
    meta = torch.tensor(np.concatenate([[0], np.cumsum(lengths), np.concatenate(dates)]), dtype=torch.int64, device=dev)
    x = torch.empty(8, 61, 10, H, H, device=dev)
    dd = torch.empty(8, 61, dtype=torch.int64, device=dev)
    valid = torch.empty(8 * 61, dtype=torch.int32, device=dev)
    order_a = (C.c_int * 10)(*CHANNELS_LIKE_PASTIS)
    mean_a, std_a = mean.ctypes.data_as(C.POINTER(C.c_float)), std.ctypes.data_as(C.POINTER(C.c_float))

    def collate_kernel():
        _lib.check(L.c2s_collate_series(raw_dev.data_ptr(), _lib.SRC_I16, meta.data_ptr(), meta.data_ptr() + 8 * 9, x.data_ptr(),
                                        dd.data_ptr(), valid.data_ptr(), 8, 61, 10, 10, H * H, order_a, mean_a, std_a, 0.0,
                                        torch.cuda.current_stream().cuda_stream), "collate_series")
    ms = timed(collate_kernel, a.reps)
    line("N1 c2s_collate_series alone (int16 in HBM)", ms, total * 10 * H * H * 2 + 8 * 61 * 10 * H * H * 4)
    del raw_dev, x

    # N2: metrics tail on the logits of one step (B = 8) and of a validation batch (B = 64)
    for B in (8, 64):
        logits = torch.randn(B, K, H, H, device=dev)
        y = torch.randint(0, K, (B, H, H), device=dev)
        loss = torch.tensor(1.0, device=dev)
        meters = StepMeters(K, device="cuda")
        ms = timed(lambda: meters.update(logits, y, loss), a.reps)
        line(f"N2 StepMeters.update B={B} (argmax, top-2, 2 cms)", ms, logits.numel() * 4 + y.numel() * 8)

    # N4: boundary target and focal CE (forward + gradient) at B = 8
    y = torch.randint(0, K, (8, H, H), device=dev)
    ms = timed(lambda: boundary_target(y), a.reps)
    line("N4 boundary_target B=8", ms, y.numel() * 8 * 2)
    logits2 = torch.randn(8, 2, H, H, device=dev)
    yb = torch.randint(0, 2, (8, H, H), device=dev)
    ms = timed(lambda: focal_ce(logits2, yb, 2.0, want_grad=True), a.reps)
    line("N4 focal_ce (+gradient) B=8, 2 classes", ms, logits2.numel() * 4 * 2 + yb.numel() * 8)

    # N3: softmax + top-1 + stitch of a 10 x 10 grid of 128 x 128 patches into a 1098 x 1098 tile (one call per batch of 10)
    lg = torch.randn(10, K, H, H, device=dev)
    probs = torch.empty(K, 1098, 1098, device=dev)
    top1 = torch.empty(1098, 1098, dtype=torch.int64, device=dev)

    def stitch():
        for first in range(0, 100, 10):
            _lib.check(L.c2s_softmax_stitch(lg.data_ptr(), probs.data_ptr(), top1.data_ptr(), first, 10, K, H, H, 10, 1098, 1098,
                                            torch.cuda.current_stream().cuda_stream), "softmax_stitch")
    ms = timed(stitch, a.reps)
    line("N3 c2s_softmax_stitch, 100 patches -> 1098^2", ms, 100 * K * H * H * 4 + K * 1098 * 1098 * 4 + 1098 * 1098 * 8)


if __name__ == "__main__":
    main()
