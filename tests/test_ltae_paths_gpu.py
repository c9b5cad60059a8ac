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
