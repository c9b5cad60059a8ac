"""Normalisation layers in isolation: one-pass vs two-pass kernels, forward and backward, HIP-event timed.
    python tools/norm_bench.py            (on the GPU box)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from crop2seg_amd import _lib as L
from crop2seg_amd import engine as E

dev = torch.device("cuda")
E.ONEPASS_MIN_HW = 256


def run(shape, kind, groups, onepass, valid_frac=0.0, iters=20):
    E.ONEPASS_NORM = onepass
    N, C, H, W = shape
    x = torch.randn(shape, device=dev)
    g = torch.randn(shape, device=dev)
    p = {"n.weight": torch.ones(C, device=dev), "n.bias": torch.zeros(C, device=dev)}
    b = {"n.running_mean": torch.zeros(C, device=dev), "n.running_var": torch.ones(C, device=dev),
         "n.num_batches_tracked": torch.zeros((), dtype=torch.int64, device=dev)}
    gr = {k: torch.zeros_like(v) for k, v in p.items()}
    ws = E.Workspace(dev)
    tf, tb = [], []
    for it in range(iters + 3):
        ctx = E.Ctx(p, b, gr, ws, True, E.Tape())
        e = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        e[0].record()
        y = E.norm_act(ctx, x, "n", kind, groups, True, None, None, 0.0)
        e[1].record()
        ctx.tape.grads[y.data_ptr()] = g.clone()
        e[2].record()
        old = E.SIDE_WGRAD
        E.SIDE_WGRAD = False
        ctx.tape.backward()
        E.SIDE_WGRAD = old
        e[3].record()
        torch.cuda.synchronize()
        if it >= 3:
            tf.append(e[0].elapsed_time(e[1]))
            tb.append(e[2].elapsed_time(e[3]))
    assert ws.sync_error() == 0
    mb = x.numel() * 4 / 1e6
    f, bw = sum(tf) / len(tf), sum(tb) / len(tb)
    print(f"{str(shape):22s} {'batch' if kind == L.NORM_BATCH else 'group'} onepass={int(onepass)}  fwd {f * 1e3:7.1f} us ({mb * (2 if onepass else 3) / f / 1e3:5.2f} TB/s)"
          f"  bwd {bw * 1e3:7.1f} us ({mb * (3 if onepass else 5) / bw / 1e3:5.2f} TB/s)", flush=True)


for shape, kind, groups in (((128, 64, 128, 128), L.NORM_GROUP, 4), ((128, 64, 64, 64), L.NORM_GROUP, 4), ((128, 64, 32, 32), L.NORM_GROUP, 4),
                            ((128, 128, 16, 16), L.NORM_GROUP, 4), ((4, 32, 128, 128), L.NORM_BATCH, 1), ((4, 64, 32, 32), L.NORM_BATCH, 1)):
    for onepass in (False, True):
        run(shape, kind, groups, onepass)
