#!/usr/bin/env python
"""L-TAE block alone at the TimeUNet shape (B=8, T=61, C=64, 128x128 pixels; BASELINE.json configs[2]):
HIP-event time of forward and backward and the algorithmic HBM rate
  forward : x once + attn + attn_pre + emb                       (bytes the block has to move at least)
  backward: x once + gx + attn, attn_pre, g_attn once + g_emb
Usage (GPU box): python tools/ltae_bench.py [--B 8 --T 61 --hw 128]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from crop2seg_amd import _lib  # noqa: E402
if os.environ.get("C2S_DIAG_LIB"):          # diagnostic builds (tools/_diag/libs): load-only / compute-only kernels
    _lib.LIB_PATH = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.environ["C2S_DIAG_LIB"])
from crop2seg_amd import engine as E  # noqa: E402
from oracle import seeded  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--T", type=int, default=61)
    ap.add_argument("--C", type=int, default=64)
    ap.add_argument("--hw", type=int, default=128)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-attn", action="store_true", help="TimeUNet's call: nobody reads the post-dropout weights (need_attn=False)")
    a = ap.parse_args()
    dev = torch.device("cuda")
    B, T, C, h = a.B, a.T, a.C, a.hw
    ks = [("te.inconv.weight", (256, C, 1)), ("te.inconv.bias", (256,)), ("te.attention_head.Q", (16, 1, 4)),
          ("te.attention_head.fc1_k.weight", (64, 256)), ("te.attention_head.fc1_k.bias", (64,)),
          ("te.in_norm.weight", (C,)), ("te.in_norm.bias", (C,))]
    sd = {k: v.to(dev) for k, v in seeded.make_state(ks, 21, "tame").items()}
    x = torch.randn(B, T, C, h, h, device=dev)
    dates = (5 * torch.arange(T, device=dev)[None]).repeat(B, 1)
    valid = torch.ones(B * T, dtype=torch.int32, device=dev)
    P = B * h * h
    na = 1 if a.no_attn else 2                      # attention tensors the forward writes
    fwd_bytes = 4.0 * (P * T * C + na * 16 * P * T + 256 * P)
    bwd_bytes = 4.0 * (2 * P * T * C + (1 if a.no_attn else 3) * 16 * P * T + 256 * P)
    for rep in range(a.reps):
        grads = {k: torch.empty_like(v) for k, v in sd.items()}
        ctx = E.Ctx(sd, {}, grads, E.Workspace(dev), True, E.Tape())
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        emb, attn = E.ltae_attention(ctx, x, dates, valid, "te", 16, 4, 256, 1000.0, 0.1, True, 1234, None, need_attn=not a.no_attn)
        e1.record()
        if attn is not None:
            ctx.tape.grads[attn.data_ptr()] = torch.randn_like(attn)
        ctx.tape.grads[emb.data_ptr()] = torch.randn_like(emb)
        torch.cuda.synchronize()
        e1b = torch.cuda.Event(enable_timing=True)
        e1b.record()
        ctx.tape.backward()
        e2.record()
        torch.cuda.synchronize()
        tf, tb = e0.elapsed_time(e1), e1b.elapsed_time(e2)
        print(f"rep {rep}: forward {tf:7.3f} ms  {fwd_bytes / tf / 1e9:6.2f} TB/s algorithmic ({fwd_bytes / tf / 1e9 / 8 * 100:4.1f} % of 8 TB/s) | "
              f"backward {tb:7.3f} ms  {bwd_bytes / tb / 1e9:6.2f} TB/s algorithmic ({bwd_bytes / tb / 1e9 / 8 * 100:4.1f} %)", flush=True)


if __name__ == "__main__":
    main()
