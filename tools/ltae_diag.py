#!/usr/bin/env python
"""Diagnostic (GPU box): time the register-resident L-TAE forward kernels of a diagnostic library (C2S_DIAG_LIB, e.g.
tools/_diag/libs/libc2s_loadonly.so: the kernels return after their load phase) at the TimeUNet shape."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crop2seg_amd import _lib  # noqa: E402
if os.environ.get("C2S_DIAG_LIB"):
    _lib.LIB_PATH = os.path.join(ROOT, os.environ["C2S_DIAG_LIB"])
from crop2seg_amd import engine as E  # noqa: E402
from oracle import seeded  # noqa: E402

dev = torch.device("cuda")
Bn, T, Cc, h = 8, 61, 64, 128
ks = [("te.inconv.weight", (256, Cc, 1)), ("te.inconv.bias", (256,)), ("te.attention_head.Q", (16, 1, 4)),
      ("te.attention_head.fc1_k.weight", (64, 256)), ("te.attention_head.fc1_k.bias", (64,)),
      ("te.in_norm.weight", (Cc,)), ("te.in_norm.bias", (Cc,))]
sd = {k: v.to(dev) for k, v in seeded.make_state(ks, 21, "tame").items()}
x = torch.randn(Bn, T, Cc, h, h, device=dev)
dates = (5 * torch.arange(T, device=dev)[None]).repeat(Bn, 1)
valid = torch.ones(Bn * T, dtype=torch.int32, device=dev)
for rep in range(4):
    ctx = E.Ctx(sd, {}, {k: torch.empty_like(v) for k, v in sd.items()}, E.Workspace(dev), True, None)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    emb, attn = E.ltae_attention(ctx, x, dates, valid, "te", 16, 4, 256, 1000.0, 0.1, True, 1234, None, need_attn=False)
    t1.record()
    torch.cuda.synchronize()
    print(f"rep {rep}: forward call {t0.elapsed_time(t1):.3f} ms (includes the fold / table launches)", flush=True)
