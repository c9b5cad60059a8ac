#!/usr/bin/env python
"""Diagnostic (GPU box): forward error of the small-map L-TAE kernels against the fp64 oracle, LDS-resident (C2S_LTAE_LDS=1)
vs 16-pixel kernel (C2S_LTAE_LDS=0), in child processes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from crop2seg_amd import engine as E
from oracle import crop2seg_oracle as O
sys.path.insert(0, os.path.join(%r, "tests"))
import test_ops_gpu as TO
for (B, T, C, h) in [(2, 4, 64, 16), (2, 5, 128, 4), (4, 32, 128, 16)]:
    g = torch.Generator().manual_seed(13)
    sd = TO._ltae_state(C, g)
    cfg = O.BackboneConfig()
    x = torch.randn(B, T, C, h, h, generator=g)
    dates = (5 * torch.arange(T)[None] + torch.arange(B)[:, None]).long()
    valid = torch.ones(B, T, dtype=torch.int32)
    sd64 = {k: v.double() for k, v in sd.items()}
    with torch.no_grad():
        e64, a64 = O.ltae_attention(x.double(), dates, ~valid.bool(), sd64, "te", cfg, None)
        e32, a32 = O.ltae_attention(x, dates, ~valid.bool(), sd, "te", cfg, None)
    ctx = TO.make_ctx({k: v for k, v in sd.items()}, training=True)
    e_out, a_out = E.ltae_attention(ctx, x.cuda(), dates.cuda(), valid.view(-1).cuda(), "te", 16, 4, 256, 1000.0, 0.0, True, 0, None)
    a5 = a64.view(16, B, h, h, T).permute(0, 1, 4, 2, 3)
    e4 = e64.view(B, h, h, 256).permute(0, 3, 1, 2)
    print(f"  B{B} T{T} C{C} {h}x{h}: attn max err {float((a_out.cpu().double() - a5).abs().max()):.2e} (oracle32 {float((a32.double() - a64).abs().max()):.2e}); "
          f"emb rel {float((e_out.cpu().double() - e4).norm() / e4.norm()):.2e} (oracle32 {float((e32.double() - e64).norm() / e64.norm()):.2e})")
''' % (ROOT, ROOT)
for lds in ("1", "0"):
    print("C2S_LTAE_LDS=" + lds)
    r = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, C2S_LTAE_LDS=lds), capture_output=True, text=True)
    print(r.stdout, r.stderr[-800:] if r.returncode else "")
