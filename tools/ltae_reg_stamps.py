#!/usr/bin/env python
"""Diagnostic (GPU box): phase times of the register-resident L-TAE forward kernel (stamped build, -DC2S_LT_STAMP)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from crop2seg_amd import build as B  # noqa: E402

out = os.path.join(ROOT, "gpurun_out", "libc2s_ltstamp.so")
os.makedirs(os.path.dirname(out), exist_ok=True)
srcs = [os.path.join(B.CSRC, s) for s in B.SOURCES]
subprocess.check_call([B.hipcc(), *B.FLAGS, "-DC2S_LT_STAMP", *os.environ.get("C2S_EXTRA_FLAGS", "").split(), "-shared", "-fPIC", *srcs, "-o", out])
from crop2seg_amd import _lib  # noqa: E402
_lib.LIB_PATH = out
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
# C2S_CU_MASK=half: run on a stream restricted to every other CU (hipExtStreamCreateWithCUMask) -- if the load phases get
# shorter per tile, they are bound by the memory system as a whole; if not, by what one CU can request
stream_ctx = None
if os.environ.get("C2S_CU_MASK") == "half":
    hip = C.CDLL("libamdhip64.so")
    st = C.c_void_p()
    mask = (C.c_uint32 * 8)(*([0x55555555] * 8))
    assert hip.hipExtStreamCreateWithCUMask(C.byref(st), 8, mask) == 0
    stream_ctx = torch.cuda.stream(torch.cuda.ExternalStream(st.value))
    stream_ctx.__enter__()
    print("running on a CU-masked stream (every other CU)")
for _ in range(2):
    ctx = E.Ctx(sd, {}, {k: torch.empty_like(v) for k, v in sd.items()}, E.Workspace(dev), True, None)
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    emb, attn = E.ltae_attention(ctx, x, dates, valid, "te", 16, 4, 256, 1000.0, 0.1, True, 1234, None)
    t1.record()
torch.cuda.synchronize()
print(f"forward call: {t0.elapsed_time(t1):.3f} ms")
if stream_ctx is not None:
    stream_ctx.__exit__(None, None, None)
lib = E.lib()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.c2s_debug_ltae_stamps.argtypes = [C.c_void_p]
assert lib.c2s_debug_ltae_stamps(buf.ctypes.data) == 0
s = buf.reshape(-1, 8).astype(np.int64)
s = s[s[:, 0] > 0]
names = ["F1+F2 load, statistics", "F3 scores (MFMA)", "F4 softmax, dropout, stores", "F5 z per pixel (MFMA)", "F5 pe + transposition", "F6 emb (MFMA) + stores"]
for k, n in enumerate(names):
    v = s[:, k + 1] - s[:, k]
    print(f"{n:28s} median {np.median(v):9.0f}  p10 {np.percentile(v, 10):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
v = s[:, 6] - s[:, 0]
print(f"{'total (wave 0)':28s} median {np.median(v):9.0f}   ({len(s)} workgroups sampled)")
span = s[:, 6].max() - s[:, 0].min()
print(f"stamped workgroups span {span} cycles of the kernel (first {len(s)} of {Bn * h * h // 16} tiles)")
