"""Every L-TAE kernel of csrc/ltae.hip under an oracle test: the default dispatch takes the register-resident kernels at the
TimeUNet shapes, so the three-pass streaming kernels (ltae_prep / ltae_stream_fwd, ltae_stream_bwd_heads<4>,
ltae_stream_bwd_gx<4>) are only reached through the dispatch switches or the size fallbacks.  The switches are read once
per process, hence the child processes (nothing is re-exec'ed; pattern of tests/test_dist_gpu.py)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAM_CASES = ["2,61,64,128,1,1,1", "2,5,64,128,1,1,1", "2,4,64,128,0,0,0"]
# small maps (U-TAE / W-TAE: L-TAE on 16 x 16 and smaller): B, T, C, h, with_emb, pad, drop
SMALL_CASES = ["2,6,128,4,1,1,0", "1,5,64,8,1,0,1", "2,7,128,4,0,1,1", "4,32,128,16,1,1,1", "1,61,64,16,1,1,0"]


def _run(env_extra, cases):
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ltae_env_worker.py"), *cases], env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"LTAE_ENV_OK {len(cases)}" in r.stdout
    return r.stdout


def test_streaming_forward_and_backward_kernels():
    """C2S_LTAE_REG=0: ltae_prep + ltae_stream_fwd; with it the backward falls to ltae_stream_bwd_heads<4>; C2S_LTAE_GX64=0
    adds ltae_stream_bwd_gx<4>."""
    _run({"C2S_LTAE_REG": "0", "C2S_LTAE_REG_BWD": "0", "C2S_LTAE_GX64": "0"}, STREAM_CASES)


def test_streaming_heads_with_the_64_pixel_dx_kernel():
    """C2S_LTAE_REG_BWD=0 alone: register-resident forward, ltae_stream_bwd_heads<4> feeding ltae_stream_bwd_gx64 (the
    combination the size fallback of c2s_ltae_attn_bwd selects when 16*B*T*HW >= 2^29)."""
    _run({"C2S_LTAE_REG_BWD": "0"}, STREAM_CASES[:2])


def test_plane_that_is_not_a_multiple_of_16_pixels_takes_the_streaming_path():
    """HW % 16 != 0 at C = 64 with enough tiles: the default dispatch itself (no switch) leaves the register-resident
    kernels -- 180 x 182 = 32760 pixels (multiple of 8, not of 16), B = 2: 1024 tiles of 64 pixels."""
    import ctypes as C
    from crop2seg_amd import _lib
    d = _lib.LtaeDesc(2, 5, 64, 180 * 182, 16, 256, 1e-5, 0.0, 0, None, None)
    assert _lib.lib().c2s_ltae_uses_streaming(C.byref(d)) == 1


def test_small_map_kernels_of_round_1_still_match_the_oracle():
    """The default dispatch takes the LDS-resident 4-pixel kernels on small maps (round 4); C2S_LTAE_LDS=0 / C2S_LTAE_LDS_BWD=0
    select the 16-pixel forward and the 8-pixel backward pair they replaced -- still the path for series whose tile does not
    fit in LDS -- on the same cases, the bench shape (B=4, T=32, C=128, 16 x 16) included."""
    _run({"C2S_LTAE_LDS": "0", "C2S_LTAE_LDS_BWD": "0"}, SMALL_CASES)


def test_lds_resident_kernels_at_their_size_limits():
    """Default dispatch, shapes around the LDS limits of the 4-pixel kernels: C = 256 (one time slice per thread, 16 heads per
    thread in the z phase; T <= 31 in the forward); T = 48 at C = 128 (the forward's tile fits, the backward's does not: the
    8-pixel pair runs behind the LDS-resident forward); C = 192 (neither takes it); T = 39 at C = 128, the longest series of the
    fused backward."""
    _run({}, ["1,9,256,8,1,1,1", "2,48,128,8,1,1,1", "1,6,192,8,1,0,1", "1,39,128,8,1,1,1"])


@pytest.mark.parametrize("B,fwd_path,optional", [
    (9, 2, 0),      # 16 B T HW = 5.8e8 >= 2^29: register-resident forward, but the backward leaves its 31-bit buffer descriptors
    (17, 1, 0),     # 16 B T HW = 1.09e9 >= 2^30: the forward leaves its 32-bit element offsets too (three-pass streaming kernels)
])
def test_size_fallbacks_of_the_dispatch_at_the_sizes_that_trigger_them(B, fwd_path, optional):
    """TimeUNet's L-TAE at 256 x 256 with T = 61 and a batch large enough to trip the size guards of c2s_ltae_attn_fwd_ws /
    c2s_ltae_attn_bwd -- no switch, the real dispatch at a size a 288 GB card holds (x alone: 9.2 / 17.4 GB).  The oracle cannot
    run there; the block is independent across batch elements, so the reference is the SAME block on sub-batches of four, which
    take the register-resident kernels that tests/test_ops_gpu.py pins to the oracle: outputs and d x per batch element, parameter
    gradients as the sum over the sub-batches."""
    import ctypes
    import torch
    from test_ops_gpu import _engine, _ltae_state, make_ctx, rel
    E, L = _engine()
    T, C, h = 61, 64, 256
    d = L.LtaeDesc(B, T, C, h * h, 16, 256, 1e-5, 0.0, 0, None, None)
    assert L.lib().c2s_ltae_fwd_path(ctypes.byref(d)) == fwd_path
    assert L.lib().c2s_ltae_attn_optional(ctypes.byref(d)) == optional
    d4 = L.LtaeDesc(4, T, C, h * h, 16, 256, 1e-5, 0.0, 0, None, None)
    assert L.lib().c2s_ltae_fwd_path(ctypes.byref(d4)) == 2 and L.lib().c2s_ltae_attn_optional(ctypes.byref(d4)) == 1
    torch.cuda.empty_cache()
    g = torch.Generator(device="cuda").manual_seed(41)
    sd = _ltae_state(C, torch.Generator().manual_seed(13))
    x = torch.randn(B, T, C, h, h, generator=g, device="cuda")
    dates = (5 * torch.arange(T)[None] + torch.arange(B)[:, None]).long()
    valid = torch.ones(B, T, dtype=torch.int32)
    for b in range(0, B, 3):                       # irregular series lengths (reference README.md:92: 27 .. 61 acquisitions)
        tb = 27 + 5 * (b % 7)
        valid[b, tb:] = 0
        dates[b, tb:] = 0
        x[b, tb:] = 0
    g_emb = torch.randn(B, 256, h, h, generator=g, device="cuda")
    dd, vd = dates.cuda(), valid.cuda()

    def run(lo, hi):
        ctx = make_ctx({k: v for k, v in sd.items()}, training=True)
        xs = x[lo:hi].contiguous()
        e, a = E.ltae_attention(ctx, xs, dd[lo:hi].contiguous(), vd[lo:hi].reshape(-1).contiguous(), "te", 16, 4, 256, 1000.0,
                                0.0, True, 0, None)
        ctx.tape.grads[e.data_ptr()] = g_emb[lo:hi].contiguous()
        ctx.tape.backward()
        torch.cuda.synchronize()
        return e, a, ctx.tape.grads[xs.data_ptr()], {k: v.clone() for k, v in ctx.g.items()}

    e_f, a_f, gx_f, p_f = run(0, B)
    assert bool(torch.isfinite(e_f).all()) and bool(torch.isfinite(gx_f).all())
    p_sum = None
    for lo in range(0, B, 4):
        hi = min(lo + 4, B)
        e_s, a_s, gx_s, p_s = run(lo, hi)
        assert rel(e_f[lo:hi], e_s) < 2e-5, (lo, "emb")
        assert float((a_f[:, lo:hi] - a_s).abs().max()) < 5e-6, (lo, "attn")
        assert rel(gx_f[lo:hi], gx_s) < 1e-4, (lo, "gx")
        p_sum = p_s if p_sum is None else {k: p_sum[k] + v for k, v in p_s.items()}
        del e_s, a_s, gx_s, p_s
    gmax = max(float(v.norm()) for v in p_sum.values())
    for k, v in p_sum.items():
        assert float((p_f[k] - v).norm()) <= 1e-4 * float(v.norm()) + 1e-6 * gmax, k
    del e_f, a_f, gx_f, x, g_emb
    torch.cuda.empty_cache()
