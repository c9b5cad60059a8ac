#!/usr/bin/env python
"""Micro-benchmark of the convolution kernels at the U-TAE B=4,T=32 layer shapes (HIP-event timing per launch).
Usage (GPU box): python tools/kbench.py [--reps 5]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crop2seg_amd import _lib  # noqa: E402
if os.environ.get("C2S_KBENCH_LIB"):          # A/B runs against an older build of the library
    _lib.LIB_PATH = os.environ["C2S_KBENCH_LIB"]
from crop2seg_amd import engine as E  # noqa: E402

L = _lib


def time_fn(fn, reps):
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
    return min(ts), sum(ts) / len(ts)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--only", default="")
    ap.add_argument("--mode", default="f32", choices=["f32", "bf16x3"])
    args = ap.parse_args()
    dev = torch.device("cuda")
    E.CONV_MODE = args.mode
    E.REDUCE_BATCH = False          # time each weight gradient with its own slice sum
    shapes = [  # name, N, Cin, Cout, H, K, S, pad, mode
        ("in_conv.3   64->64  3x3 @128", 128, 64, 64, 128, 3, 1, 1, L.PAD_REFLECT),
        ("in_conv.0   10->64  3x3 @128", 128, 10, 64, 128, 3, 1, 1, L.PAD_REFLECT),
        ("down0.down  64->64  4x4s2 @128", 128, 64, 64, 128, 4, 2, 1, L.PAD_REFLECT),
        ("down0.conv1 64->64  3x3 @64", 128, 64, 64, 64, 3, 1, 1, L.PAD_REFLECT),
        ("down1.down  64->64  4x4s2 @64", 128, 64, 64, 64, 4, 2, 1, L.PAD_REFLECT),
        ("down1.conv1 64->64  3x3 @32", 128, 64, 64, 32, 3, 1, 1, L.PAD_REFLECT),
        ("down2.conv1 64->128 3x3 @16", 128, 64, 128, 16, 3, 1, 1, L.PAD_REFLECT),
        ("down2.down  64->128 4x4s2 @32", 128, 64, 128, 32, 4, 2, 1, L.PAD_REFLECT),
        ("down2.conv2 128->128 3x3 @16", 128, 128, 128, 16, 3, 1, 1, L.PAD_REFLECT),
        ("up2.conv1   96->32  3x3 @128 (B=4)", 4, 96, 32, 128, 3, 1, 1, L.PAD_REFLECT),
        ("wtae pointwise 64->64 1x1 @128", 128, 64, 64, 128, 1, 1, 0, L.PAD_ZEROS),
    ]
    print(f"{'layer':36s} {'GFLOP':>7s} | {'fwd ms':>7s} {'TF':>6s} | {'dgrad ms':>8s} {'TF':>6s} | {'wgrad ms':>8s} {'TF':>6s}")
    for name, N, Cin, Cout, H, K, S, pad, mode in shapes:
        if args.only and args.only not in name:
            continue
        Ho = (H + 2 * pad - K) // S + 1
        gflop = 2.0 * N * Cin * Cout * K * K * Ho * Ho / 1e9
        w = torch.randn(Cout, Cin, K, K, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        x = torch.randn(N, Cin, H, H, device=dev)
        params = {"w": w, "b": b}
        grads = {"w": torch.empty_like(w), "b": torch.empty_like(b)}
        ws = E.Workspace(dev)
        res = {}

        def fwd():
            ctx = E.Ctx(params, {}, grads, ws, True, None)
            res["y"] = E.conv2d(ctx, [x], "w", "b", K, S, pad, mode, None)
        t_f = time_fn(fwd, args.reps)
        gy = torch.randn_like(res["y"])
        # backward pieces timed separately: build a tape, then call the two halves
        ctx = E.Ctx(params, {}, grads, ws, True, E.Tape())
        y = E.conv2d(ctx, [x], "w", "b", K, S, pad, mode, None)
        KK = K * K

        vflags = torch.ones(N, dtype=torch.int32, device=dev)      # frame flags as in a training step (all real)

        def wgrad():
            E._wgrad_launch(ctx, [x], gy, Cout, Ho, Ho, K, S, pad, mode, grads["w"], Cin * KK, KK, list(range(KK)), 0, vflags)
        t_w = time_fn(wgrad, args.reps)

        # backward (wgrad + dgrad) timed alone: the forward that builds the tape runs outside the events
        ts = []
        for _ in range(args.reps + 1):
            c2 = E.Ctx(params, {}, {"w": torch.empty_like(w), "b": torch.empty_like(b)}, ws, True, E.Tape())
            yy = E.conv2d(c2, [x], "w", "b", K, S, pad, mode, None)
            c2.tape.grads[yy.data_ptr()] = gy
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            c2.tape.backward()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        t_d = max(min(ts[1:]) - t_w[0], 1e-6)
        print(f"{name:36s} {gflop:7.1f} | {t_f[0]:7.3f} {gflop / t_f[0]:6.1f} | {t_d:8.3f} {gflop / t_d:6.1f} | {t_w[0]:8.3f} {gflop / t_w[0]:6.1f}")


if __name__ == "__main__":
    main()
