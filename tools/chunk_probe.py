#!/usr/bin/env python
"""Diagnostic (GPU box): does running the per-frame layer chain of the encoder on frame CHUNKS keep producer -> consumer
traffic in the 256 MB Infinity Cache?  conv3x3 64->64 @128^2 -> GroupNorm+ReLU -> conv3x3 -> GroupNorm+ReLU on N = 128 frames at
once, against four passes over 32 frames each (forward only, same kernels).  Not part of the product."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crop2seg_amd import _lib, engine as E  # noqa: E402

L = _lib
dev = torch.device("cuda")
N, Cc, H = 128, 64, 128
g = torch.Generator(device="cpu").manual_seed(0)
p = {"w1": (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05).to(dev), "b1": torch.randn(Cc, generator=g).to(dev),
     "w2": (torch.randn(Cc, Cc, 3, 3, generator=g) * 0.05).to(dev), "b2": torch.randn(Cc, generator=g).to(dev),
     "n1.weight": torch.ones(Cc, device=dev), "n1.bias": torch.zeros(Cc, device=dev),
     "n2.weight": torch.ones(Cc, device=dev), "n2.bias": torch.zeros(Cc, device=dev)}
x = torch.randn(N, Cc, H, H, device=dev)
ws = E.Workspace(dev)


def chain(xc, ctx):
    h = E.conv2d(ctx, [xc], "w1", "b1", 3, 1, 1, L.PAD_REFLECT, None)
    h = E.norm_act(ctx, h, "n1", L.NORM_GROUP, 4, True, None, None)
    h = E.conv2d(ctx, [h], "w2", "b2", 3, 1, 1, L.PAD_REFLECT, None)
    return E.norm_act(ctx, h, "n2", L.NORM_GROUP, 4, True, None, None)


grads = {k: torch.zeros_like(v) for k, v in p.items()}
gfull = torch.randn(N, Cc, H, H, device=dev)


def run(nchunks, backward=False):
    outs = []
    step = N // nchunks
    if not backward:
        ctx = E.Ctx(p, {}, None, ws, True, None)
        for i in range(nchunks):
            outs.append(chain(x[i * step:(i + 1) * step], ctx))
        return outs
    gw = set()
    for i in range(nchunks):          # forward + backward of one chunk at a time (weight gradients accumulate over the chunks)
        ctx = E.Ctx(p, {}, grads, ws, True, E.Tape())
        ctx._gwritten = gw
        xc = x[i * step:(i + 1) * step]
        y = chain(xc, ctx)
        ctx.tape.grads[y.data_ptr()] = gfull[i * step:(i + 1) * step]
        ctx.tape.backward()
    return outs


BWD = "--bwd" in sys.argv
for nchunks in (1, 2, 4, 8):
    run(nchunks, BWD)
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(nchunks, BWD)
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print(f"{nchunks} chunk(s) of {N // nchunks:3d} frames: min {min(ts):.3f} ms   median {sorted(ts)[3]:.3f} ms", flush=True)
