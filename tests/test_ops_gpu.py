"""GPU parity tests of the individual HIP ops (through the C ABI) against the CPU oracle
(oracle/crop2seg_oracle.py restatements + torch CPU autograd on them).  Tolerances: fp32 kernels vs an fp32 (or
fp64) CPU evaluation of the same formula; the north star's bar is 1e-3 relative, the kernels are held to 1e-4..1e-5
unless stated."""
import ctypes
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

from oracle import crop2seg_oracle as O  # noqa: E402


def _engine():
    from crop2seg_amd import engine as E
    from crop2seg_amd import _lib
    return E, _lib


def rel(a, b):
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def make_ctx(params, buffers=None, training=True, tape=True):
    E, _ = _engine()
    dev = torch.device("cuda")
    p = {k: v.to(dev).contiguous() for k, v in params.items()}
    b = {k: v.to(dev).contiguous() for k, v in (buffers or {}).items()}
    g = {k: torch.full_like(v, float("nan")) for k, v in p.items()}
    return E.Ctx(p, b, g, E.Workspace(dev), training, E.Tape() if tape else None)


def seed_backward(ctx, out, gout):
    ctx.tape.grads[out.data_ptr()] = gout.to(out.device).contiguous().clone()
    ctx.tape.backward()


CONV_CASES = [
    # N, Cin, Cout, H, W, K, S, pad, mode
    (3, 10, 64, 32, 32, 3, 1, 1, "reflect"),
    (2, 64, 64, 128, 128, 3, 1, 1, "reflect"),
    (2, 64, 128, 16, 16, 3, 1, 1, "reflect"),
    (2, 32, 15, 32, 32, 3, 1, 1, "reflect"),
    (3, 64, 64, 8, 8, 3, 1, 1, "reflect"),
    (2, 128, 128, 4, 4, 3, 1, 1, "reflect"),
    (2, 64, 64, 64, 64, 4, 2, 1, "reflect"),
    (3, 64, 64, 8, 8, 4, 2, 1, "reflect"),
    (2, 64, 64, 32, 32, 1, 1, 0, "zeros"),
    (2, 256, 128, 4, 4, 1, 1, 0, "zeros"),
    (1, 24, 40, 24, 40, 3, 1, 1, "reflect"),      # ragged: sizes that are not powers of two
    (2, 4, 64, 16, 64, 3, 1, 1, "zeros"),         # few-input-channel kernel: 2 channel pairs, zero padding
    (3, 9, 128, 8, 32, 3, 1, 1, "reflect"),       # ... odd channel count, two blocks of 64 output channels, one tile per frame
    # padded frame (valid[1] = 0) on the wide full-resolution kernels real batches run: Winograd forward / data / weight
    # gradient (next_tile skip + XCD start permutation), and the 4x4 stride-2 forward + transposed-row data gradient
    (3, 64, 64, 128, 128, 3, 1, 1, "reflect"),
    (3, 64, 64, 64, 64, 4, 2, 1, "reflect"),
    (5, 64, 64, 64, 64, 3, 1, 1, "reflect"),      # two padded frames (1 and 3): tiles of several skipped frames in a row
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case):
    E, L = _engine()
    N, Cin, Cout, H, W, K, S, pad, mode = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)))
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, K, K, generator=g) / math.sqrt(Cin * K * K)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    valid = torch.ones(N, dtype=torch.int32)
    if N >= 3:
        valid[1] = 0
    if N >= 5:
        valid[3] = 0
    keep = valid.bool()
    ref = O.conv2d(x[keep], w, b, S, pad, mode)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)

    ctx = make_ctx({"w": w.detach(), "b": b.detach()})
    xd = x.detach().cuda()
    vd = valid.cuda()
    out = E.conv2d(ctx, [xd], "w", "b", K, S, pad, L.PAD_REFLECT if mode == "reflect" else L.PAD_ZEROS, vd)
    assert rel(out[keep.cuda()], ref) < 2e-6
    gfull = torch.zeros(N, *ref.shape[1:])
    gfull[keep] = gout
    seed_backward(ctx, out, gfull)
    gx = ctx.tape.grads[xd.data_ptr()]
    assert rel(gx[keep.cuda()], x.grad[keep]) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6


WIDE_WINOGRAD_CASES = [
    # N, C0, C1, Cout, H, W, mode
    (2, 64, 0, 64, 128, 128, "reflect"),
    (3, 64, 0, 64, 64, 64, "reflect"),          # padded frame 1
    (2, 64, 0, 128, 32, 32, "reflect"),         # two blocks of output channels, one tile column
    (1, 32, 0, 72, 12, 40, "reflect"),          # ragged: partial tiles in both directions, padded output channels
    (2, 40, 0, 64, 8, 32, "zeros"),             # one tile per frame, zero padding, 5 chunks
    (2, 32, 64, 64, 32, 64, "reflect"),         # two sources (the decoder's [up, skip])
    # ragged channel counts (not a multiple of the 8-channel chunk): must stay OFF the 8-wave kernel -- its chunk base travels
    # in an SGPR offset that the buffer range check ignores -- and be right on the 4-wave kernel; N = 1: an over-read of the
    # last chunk would leave the allocation
    (1, 36, 0, 64, 16, 32, "reflect"),
    (1, 100, 0, 64, 8, 32, "zeros"),
    (1, 64, 0, 36, 16, 32, "reflect"),          # ragged Cout: the data gradient's input channels
]


@pytest.mark.parametrize("case", WIDE_WINOGRAD_CASES)
def test_conv2d_wide_winograd(case):
    """The 8-wave Winograd kernel (conv_winograd16.hip; C2S_WINO16) against the oracle: forward, data gradient (reflect
    adjoint on the raw patches), and it must be the kernel that ran."""
    E, L = _engine()
    N, C0, C1, Cout, H, W, mode = case
    Cin = C0 + C1
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)))
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    valid = torch.ones(N, dtype=torch.int32)
    if N >= 3:
        valid[1] = 0
    keep = valid.bool()
    ref = O.conv2d(x[keep], w, b, 1, 1, mode)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"w": w.detach(), "b": b.detach()})
    xd = x.detach().cuda()
    srcs = [xd] if C1 == 0 else [xd[:, :C0].contiguous(), xd[:, C0:].contiguous()]
    old = E.WINO16
    E.WINO16 = True
    try:
        chans = [C0, C1] if C1 else [C0]
        ragged = any(c % 8 for c in chans)
        dsc = L.ConvDesc(N, C0, C1, H, W, Cout, (Cout + 63) // 64 * 64, H, W, H, W, 3, 3, 1, 1, 1, L.PAD_ZEROS, 1, 1, 0, 0, 0)
        assert bool(L.lib().c2s_conv3x3_winograd16_supported(ctypes.byref(dsc))) == (not ragged)
        if Cout >= 64:
            assert E._use_winograd(3, 1, 1, chans, Cout, H, W) and E._wide_winograd(H, W, chans) == (not ragged)
        assert not E._wide_winograd(H, W, [36])
        out = E.conv2d(ctx, srcs, "w", "b", 3, 1, 1, L.PAD_REFLECT if mode == "reflect" else L.PAD_ZEROS, valid.cuda())
        assert rel(out[keep.cuda()], ref) < 2e-6
        gfull = torch.zeros(N, *ref.shape[1:])
        gfull[keep] = gout
        seed_backward(ctx, out, gfull)
    finally:
        E.WINO16 = old
    gx = torch.cat([ctx.tape.grads[s.data_ptr()] for s in srcs], 1)
    assert rel(gx[keep.cuda()], x.grad[keep]) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6


S2WINO_CASES = [
    # N, Cin, Cout, Hin, Win, mode
    (2, 64, 64, 128, 128, "reflect"),
    (3, 64, 64, 64, 64, "reflect"),             # padded frame 1; one tile column
    (2, 32, 128, 64, 64, "reflect"),            # two blocks of output channels
    (1, 8, 72, 24, 80, "reflect"),              # ragged: partial tiles in both directions, padded output channels, 4 chunks
    (2, 10, 64, 16, 64, "zeros"),               # one tile per frame, zero padding
    (1, 40, 72, 16, 128, "zeros"),              # F(2x2,2x2) weight gradient with a half-empty second input block, padded output channels, zero padding
]


@pytest.mark.parametrize("case", S2WINO_CASES)
def test_conv4x4s2_winograd(case):
    """The F(2x2,2x2) kernels of the 4x4 stride-2 convolution (C2S_S2WINO) against the oracle: forward over the four input
    parities (conv_s2wino.hip) and data gradient per output parity with the reflect adjoint folded in by the variant
    3-multiply algorithm of the border blocks (conv_s2dgrad.hip); the weight gradient as F(2x2,2x2) over the four input parities
    (conv_wgrad_s2wino_kernel) where the plane tiles into 4 x 32 output pixels, and -- same inputs -- the direct kernel."""
    E, L = _engine()
    N, Cin, Cout, Hin, Win, mode = case
    g = torch.Generator().manual_seed(sum(v for v in case if isinstance(v, int)))
    x = torch.randn(N, Cin, Hin, Win, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 4, 4, generator=g) / math.sqrt(Cin * 16)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    valid = torch.ones(N, dtype=torch.int32)
    if N >= 3:
        valid[1] = 0
    keep = valid.bool()
    ref = O.conv2d(x[keep], w, b, 2, 1, mode)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"w": w.detach(), "b": b.detach()})
    xd = x.detach().cuda()
    pm = L.PAD_REFLECT if mode == "reflect" else L.PAD_ZEROS
    old = E.S2WINO
    E.S2WINO = True
    try:
        import ctypes as C
        d = L.ConvDesc(N, Cin, 0, Hin, Win, Cout, (Cout + 63) // 64 * 64, Hin // 2, Win // 2, Hin // 2, Win // 2, 4, 4, 2, 1, 1, pm, 1, 1, 0, 0, 0)
        assert E.lib().c2s_conv4x4s2_winograd_supported(C.byref(d)) == 1
        out = E.conv2d(ctx, [xd], "w", "b", 4, 2, 1, pm, valid.cuda())
        assert ("w", "fwd", "s2w") in ctx._packed
        assert rel(out[keep.cuda()], ref) < 2e-6
        gfull = torch.zeros(N, *ref.shape[1:])
        gfull[keep] = gout
        seed_backward(ctx, out, gfull)
        assert ("w", "dgrad", "s2d", 0) in ctx._packed          # the data gradient ran on conv_s2dgrad.hip as well
    finally:
        E.S2WINO = old
    assert rel(ctx.tape.grads[xd.data_ptr()][keep.cuda()], x.grad[keep]) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6
    # the other weight-gradient algorithm on the same operands
    gw1 = ctx.g["w"].clone()
    try:
        E.lib().c2s_wgrad_algorithms(-1, 0)
        ctx._gwritten.discard("w")
        E._wgrad_launch(ctx, [xd], gfull.cuda(), Cout, Hin // 2, Win // 2, 4, 2, 1, pm, ctx.g["w"], Cin * 16, 16, list(range(16)), 0, valid.cuda())
    finally:
        E.lib().c2s_wgrad_algorithms(-1, -1)
    assert rel(ctx.g["w"], w.grad) < 5e-6
    fast = Cin >= 32 and (Win // 2) % 32 == 0 and (Hin // 2) % 4 == 0
    assert torch.equal(gw1, ctx.g["w"]) == (not fast), "the F(2x2,2x2) weight gradient must be the kernel that ran where it applies"


def test_weight_gradient_slice_sums_batched():
    """C2S_REDUCE_BATCH: the split-K slabs of every layer are summed by ONE launch at the end of the backward pass
    (c2s_wgrad_reduce_batch); two stacked convolutions (3x3 Winograd path and 4x4 stride 2), two backward passes so that the
    job table is built once and reused once; same order of additions as the per-layer launches: bit-identical gradients."""
    E, L = _engine()
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 64, 32, 32, generator=g)
    w1 = torch.randn(64, 64, 3, 3, generator=g) / 24
    w2 = torch.randn(64, 64, 4, 4, generator=g) / 32
    results = {}
    for batched in (False, True):
        old = E.REDUCE_BATCH
        E.REDUCE_BATCH = batched
        try:
            ctx0 = make_ctx({"w1": w1, "w2": w2})
            xd = x.cuda()
            for _ in range(2):                          # a fresh context per step on the same workspace, as TrainStep does
                ctx = E.Ctx(ctx0.p, ctx0.b, ctx0.g, ctx0.ws, True, E.Tape())
                h = E.conv2d(ctx, [xd], "w1", None, 3, 1, 1, L.PAD_REFLECT, None, need_input_grad=False)
                y = E.conv2d(ctx, [h], "w2", None, 4, 2, 1, L.PAD_REFLECT, None)
                gout = torch.randn(y.shape, generator=torch.Generator().manual_seed(3))
                seed_backward(ctx, y, gout)
            torch.cuda.synchronize()
            results[batched] = (ctx.g["w1"].clone(), ctx.g["w2"].clone())
            if batched:
                assert ctx.ws.reduce_plan is not None and ctx.ws.reduce_plan["njobs"] == 2 and not ctx.ws.reduce_jobs
        finally:
            E.REDUCE_BATCH = old
    assert torch.equal(results[True][0], results[False][0]) and torch.equal(results[True][1], results[False][1])
    xr = x.clone().requires_grad_(True)
    w1r, w2r = w1.clone().requires_grad_(True), w2.clone().requires_grad_(True)
    yr = O.conv2d(O.conv2d(xr, w1r, None, 1, 1, "reflect"), w2r, None, 2, 1, "reflect")
    yr.backward(torch.randn(yr.shape, generator=torch.Generator().manual_seed(3)))
    assert rel(results[True][0], w1r.grad) < 5e-6 and rel(results[True][1], w2r.grad) < 5e-6


@pytest.mark.parametrize("shape", [(2, 128, 64, 4, 4), (2, 64, 32, 16, 16), (1, 32, 32, 64, 64)])
def test_conv_transpose_fwd_bwd(shape):
    E, L = _engine()
    N, Cin, Cout, H, W = shape
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cin, Cout, 4, 4, generator=g) / math.sqrt(Cin * 4)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"w": w.detach(), "b": b.detach()})
    xd = x.detach().cuda()
    out = E.conv_transpose2d(ctx, xd, "w", "b")
    assert rel(out, ref) < 2e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6


def test_conv_concat_sources():
    """conv1 of UpConvBlock reads [up, skip] without materialising torch.cat (reference conv.py:408)."""
    E, L = _engine()
    g = torch.Generator().manual_seed(7)
    a = torch.randn(2, 32, 32, 32, generator=g, requires_grad=True)
    s = torch.randn(2, 64, 32, 32, generator=g, requires_grad=True)
    w = (torch.randn(32, 96, 3, 3, generator=g) / 30).requires_grad_(True)
    b = torch.randn(32, generator=g, requires_grad=True)
    ref = O.conv2d(torch.cat([a, s], 1), w, b, 1, 1, "reflect")
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"w": w.detach(), "b": b.detach()})
    ad, sd = a.detach().cuda(), s.detach().cuda()
    out = E.conv2d(ctx, [ad, sd], "w", "b", 3, 1, 1, L.PAD_REFLECT, None)
    assert rel(out, ref) < 2e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[ad.data_ptr()], a.grad) < 5e-6
    assert rel(ctx.tape.grads[sd.data_ptr()], s.grad) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6


@pytest.mark.parametrize("hw", [16, 40, 6])          # 40: float4 interior spans; 6: one output per thread (Wo % 4 != 0)
@pytest.mark.parametrize("K,S", [(3, 1), (4, 2)])
def test_depthwise_fwd_bwd(K, S, hw):
    E, L = _engine()
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 64, hw, hw + (8 if hw == 40 else 0), generator=g, requires_grad=True)
    w = torch.randn(64, 1, K, K, generator=g, requires_grad=True)
    ref = O.conv2d(x, w, None, S, 1, "reflect", groups=64)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"w": w.detach()})
    xd = x.detach().cuda()
    out = E.depthwise_conv2d(ctx, xd, "w", K, S, 1, L.PAD_REFLECT, None)
    assert rel(out, ref) < 2e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 5e-6
    assert rel(ctx.g["w"], w.grad) < 5e-6


@pytest.mark.parametrize("kind,shape,use_res,use_valid", [
    ("group", (4, 64, 32, 32), False, True),
    ("group", (3, 128, 4, 4), True, False),
    ("group", (2, 64, 128, 128), True, True),
    ("batch", (4, 32, 16, 16), False, False),
    ("batch", (2, 15, 64, 64), True, False),
    ("group", (3, 64, 16, 32), True, True),          # 512-float segments (2 float4 per lane in the one-pass form)
    ("group", (5, 64, 64, 64), False, True),         # two full segments per row
    ("batch", (4, 32, 128, 128), True, False),       # decoder-sized BatchNorm: 32 waves per channel meet
])
@pytest.mark.parametrize("onepass", [True, False])
def test_norm_relu_fwd_bwd(kind, shape, use_res, use_valid, onepass, monkeypatch):
    E, L = _engine()
    monkeypatch.setattr(E, "ONEPASS_NORM", onepass)
    monkeypatch.setattr(E, "ONEPASS_MIN_HW", 256)         # also the small-plane instances the engine's policy leaves to two-pass
    N, C, H, W = shape
    g = torch.Generator().manual_seed(11)
    x = (torch.randn(shape, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.2 * torch.randn(C, generator=g)).requires_grad_(True)
    res = torch.randn(shape, generator=g, requires_grad=True) if use_res else None
    valid = torch.ones(N, dtype=torch.int32)
    if use_valid:
        valid[0] = 0
    keep = valid.bool()
    rm, rv = torch.zeros(C), torch.ones(C)
    if kind == "group":
        y = F.relu(F.group_norm(x[keep], 4, gamma, beta, 1e-5))
    else:
        rm2, rv2 = rm.clone(), rv.clone()
        y = F.relu(F.batch_norm(x[keep], rm2, rv2, gamma, beta, True, 0.1, 1e-5))
    if use_res:
        y = y + res[keep]
    gout = torch.randn(y.shape, generator=g)
    y.backward(gout)
    ctx = make_ctx({"n.weight": gamma.detach(), "n.bias": beta.detach(), "cb": torch.zeros(C)},
                   {"n.running_mean": rm, "n.running_var": rv, "n.num_batches_tracked": torch.zeros((), dtype=torch.int64)})
    xd = x.detach().cuda()
    rd = res.detach().cuda() if use_res else None
    vd = valid.cuda() if use_valid else None
    out = E.norm_act(ctx, xd, "n", L.NORM_GROUP if kind == "group" else L.NORM_BATCH, 4, True, rd, vd, 0.0, conv_bias="cb")
    assert rel(out[keep.cuda()], y) < 2e-6
    if use_valid:
        assert float(out[0].abs().max()) == 0.0
    if kind == "batch":
        assert rel(ctx.b["n.running_mean"], rm2) < 1e-5 and rel(ctx.b["n.running_var"], rv2) < 1e-5
        assert int(ctx.b["n.num_batches_tracked"]) == 1
    gfull = torch.zeros(shape)
    gfull[keep] = gout
    seed_backward(ctx, out, gfull)
    gx = ctx.tape.grads[xd.data_ptr()]
    tol = 2e-5
    assert rel(gx[keep.cuda()], x.grad[keep]) < tol
    assert rel(ctx.g["n.weight"], gamma.grad) < tol
    assert rel(ctx.g["n.bias"], beta.grad) < tol
    # gradient of the producing convolution's bias == per-channel sum of dx
    # (under BatchNorm the sum is structurally zero: only rounding noise, which grows with the number of summands)
    noise = 1e-6 * float(x.grad[keep].abs().sum(dim=(0, 2, 3)).max())
    assert rel(ctx.g["cb"], x.grad[keep].sum(dim=(0, 2, 3))) < 1e-4 or float(ctx.g["cb"].abs().max()) < max(1e-4, noise)
    if use_res:
        assert rel(ctx.tape.grads[rd.data_ptr()][keep.cuda()], res.grad[keep]) < 1e-6
    assert ctx.ws.sync_error() == 0, "a one-pass normalisation wait gave up"


def test_norm_onepass_failed_wait_is_loud_and_recoverable(monkeypatch):
    """A sweep that gave up leaves the error word set (hdr[3]).  From then on: incomplete groups come out NaN (never plausible
    numbers from stale partial sums), `Workspace.check_sync()` raises, re-zeroes the area and switches the process to the
    two-pass kernels, whose result equals torch's.  The error word is pre-set here (the premise is not reproduced by
    repetition): launches on a poisoned area do not wait, so a wave whose partners have not published yet must write NaN."""
    E, L = _engine()
    monkeypatch.setattr(E, "ONEPASS_NORM", True)
    monkeypatch.setattr(E, "ONEPASS_MIN_HW", 256)
    g = torch.Generator().manual_seed(3)
    N, C, H = 8, 64, 128
    x = torch.randn(N, C, H, H, generator=g)
    gamma, beta = 1 + 0.3 * torch.randn(C, generator=g), 0.2 * torch.randn(C, generator=g)
    ref = F.relu(F.group_norm(x, 4, gamma, beta, 1e-5))
    xd = x.cuda()
    ctx = make_ctx({"n.weight": gamma, "n.bias": beta}, {})
    out = E.norm_act(ctx, xd, "n", L.NORM_GROUP, 4, True, None, None, 0.0)
    assert rel(out, ref) < 2e-6 and ctx.ws.sync_error() == 0
    ctx.ws.check_sync()                                   # healthy: no raise
    ctx.ws.bufs["sync"][:16].view(torch.int32)[3] = 1     # a wait gave up
    bad = E.norm_act(ctx, xd, "n", L.NORM_GROUP, 4, True, None, None, 0.0)
    torch.cuda.synchronize()
    # every wave's (row, 2048-float segment) is either right (all partners of its group had published when its workgroup's sweep
    # looked) or entirely NaN -- never a plausible number from stale partial sums, and not a zero behind the ReLU either
    segs = bad.view(N * C, -1, 2048)
    nan_segs = torch.isnan(segs).all(dim=2)
    ok_segs = ~torch.isnan(segs).any(dim=2)
    assert bool((nan_segs | ok_segs).all())
    good = ok_segs.cpu()
    if bool(good.any()):
        assert rel(segs.cpu()[good], ref.view(N * C, -1, 2048)[good]) < 2e-6
    with pytest.raises(RuntimeError, match="one-pass normalisation wait gave up"):
        ctx.ws.check_sync()
    assert E.ONEPASS_NORM is False and ctx.ws.sync_error() == 0
    ctx.ws.check_sync()                                   # recovered: no raise
    out2 = E.norm_act(ctx, xd, "n", L.NORM_GROUP, 4, True, None, None, 0.0)       # two-pass kernels now
    assert rel(out2, ref) < 2e-6
    # the error word survives a growing area
    monkeypatch.setattr(E, "ONEPASS_NORM", True)
    ws = E.Workspace(xd.device)
    ws.sync_area(4096)[:16].view(torch.int32)[3] = 1
    ws.sync_area(1 << 20)
    assert ws.sync_error() == 1


def test_norm_onepass_under_uneven_load_matches_two_pass(monkeypatch):
    """The waves of a group meet through memory: run the one-pass kernels on the 537 MB layer shape of the U-TAE encoder
    (N=128, 64 channels, 128x128, a quarter of the frames padded) while a second stream keeps part of the chip busy, many
    times over, and compare every word with the two-pass kernels (same arithmetic in the same order)."""
    E, L = _engine()
    g = torch.Generator().manual_seed(5)
    N, C, H = 128, 64, 128
    x = (torch.randn(N, C, H, H, generator=g) * 1.5 + 0.25).cuda()
    gout = torch.randn(N, C, H, H, generator=g).cuda()
    gamma, beta = (1 + 0.3 * torch.randn(C, generator=g)), 0.2 * torch.randn(C, generator=g)
    valid = torch.ones(N, dtype=torch.int32)
    valid[::4] = 0
    vd = valid.cuda()

    def run(onepass, ctx=None):
        monkeypatch.setattr(E, "ONEPASS_NORM", onepass)
        ctx = ctx or make_ctx({"n.weight": gamma, "n.bias": beta, "cb": torch.zeros(C)}, {})
        ctx._gwritten.clear()
        out = E.norm_act(ctx, x, "n", L.NORM_GROUP, 4, True, None, vd, 0.0, conv_bias="cb")
        seed_backward(ctx, out, gout.clone())
        return ctx, out, ctx.tape.grads.pop(x.data_ptr()), [ctx.g[k].clone() for k in ("n.weight", "n.bias", "cb")]

    _, y2, gx2, p2 = run(False)
    torch.cuda.synchronize()
    # launches of other shapes share the sync area (a model's layers do): more groups and fewer groups than the layer under test
    ctx = make_ctx({"n.weight": gamma, "n.bias": beta, "cb": torch.zeros(C), "m.weight": torch.ones(32), "m.bias": torch.zeros(32)}, {})
    monkeypatch.setattr(E, "ONEPASS_NORM", True)
    monkeypatch.setattr(E, "ONEPASS_MIN_HW", 256)
    for shp, pre, groups in (((600, 64, 16, 16), "n", 4), ((2, 32, 64, 64), "m", 32)):
        xs = torch.randn(shp, generator=g).cuda()
        o1 = E.norm_act(ctx, xs, pre, L.NORM_GROUP, groups, True, None, None, 0.0)
        monkeypatch.setattr(E, "ONEPASS_NORM", False)
        o2 = E.norm_act(ctx, xs, pre, L.NORM_GROUP, groups, True, None, None, 0.0)
        monkeypatch.setattr(E, "ONEPASS_NORM", True)
        assert float((o1 - o2).abs().max()) <= 1e-6
    ctx.tape.ops.clear()
    side = torch.cuda.Stream()
    busy = torch.randn(2048, 2048, device="cuda")
    for it in range(6):
        with torch.cuda.stream(side):
            for _ in range(30):
                busy = (busy @ busy) * 1e-3
        ctx, y1, gx1, p1 = run(True, ctx)
        torch.cuda.synchronize()
        assert ctx.ws.sync_error() == 0
        assert float((y1 - y2).abs().max()) <= 1e-6 and float((gx1 - gx2).abs().max()) <= 1e-6 * float(gx2.abs().max())
        for a, b in zip(p1, p2):
            assert float((a - b).abs().max()) <= 1e-6 * max(1.0, float(b.abs().max()))
        hdr = ctx.ws.bufs["sync"][:16].view(torch.int32)
        assert int(hdr[0]) == 0 and int(hdr[3]) == 0, "finished-workgroup counter back at rest, no error"


def _ltae_state(C, g, flavour="tame"):
    from oracle import seeded
    ks = [("te.inconv.weight", (256, C, 1)), ("te.inconv.bias", (256,)), ("te.attention_head.Q", (16, 1, 4)),
          ("te.attention_head.fc1_k.weight", (64, 256)), ("te.attention_head.fc1_k.bias", (64,)),
          ("te.in_norm.weight", (C,)), ("te.in_norm.bias", (C,))]
    return seeded.make_state(ks, 21, flavour)


@pytest.mark.parametrize("B,T,C,h,with_emb,pad,drop", [
    (2, 6, 128, 4, True, True, False),
    (1, 5, 64, 8, True, False, True),
    (2, 7, 128, 4, False, True, True),
    (1, 61, 64, 16, True, True, False),     # TimeUNet-like: T = 61
    (2, 5, 64, 128, True, True, True),      # 512 tiles of 64 pixels: the streaming kernels (TimeUNet resolution)
    (2, 4, 64, 128, False, False, False),   # streaming, attention masks only
    # BASELINE configs[2] shape of the block (TimeUNet: T = 61, C = 64, 128x128 pixels per patch): the streaming kernels with
    # every T-chunk loop iterated several times and the T-dependent dynamic LDS of the backward at its maximum
    (2, 61, 64, 128, True, True, True),
    (2, 61, 64, 128, True, False, False),
    # edges of the register-resident kernels (wave w owns time steps 8w .. 8w+7): the maximum T = 64 at exactly 4 tiles per CU,
    # a partly filled second wave, and a single time step (seven of the eight waves carry no step at all)
    (1, 64, 64, 128, True, True, True),
    (2, 9, 64, 128, True, True, True),
    (2, 1, 64, 128, True, False, True),
    (4, 7, 64, 64, True, True, True),       # 64x64 planes: 256 tiles per batch element, exactly 4 tiles per CU
    # 112x112 planes: 2352 tiles of 16 pixels / 588 of 64 -- the partial sums go through ragged slices (9 x 256 + 48, 2 x 256 + 76)
    (3, 5, 64, 112, True, True, True),
])
def test_ltae_attention_fwd_bwd(B, T, C, h, with_emb, pad, drop):
    E, L = _engine()
    g = torch.Generator().manual_seed(13)
    sd = _ltae_state(C, g)
    cfg = O.BackboneConfig()
    x = torch.randn(B, T, C, h, h, generator=g)
    dates = (5 * torch.arange(T)[None] + torch.arange(B)[:, None]).long()
    valid = torch.ones(B, T, dtype=torch.int32)
    if pad:
        tb = T - 2 if T < 20 else 27        # T = 61: the shortest series of the irregular-T range (reference README.md:92)
        valid[0, tb:] = 0
        x[0, tb:] = 0
        dates[0, tb:] = 0
    P = B * h * h
    keep = (torch.rand(16, P, T, generator=g) >= 0.1).float() if drop else None
    x.requires_grad_(True)
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    emb, attn = O.ltae_attention(x, dates, ~valid.bool(), sdg, "te", cfg, keep)
    attn5 = attn.view(16, B, h, h, T).permute(0, 1, 4, 2, 3)
    emb4 = emb.view(B, h, h, 256).permute(0, 3, 1, 2)
    g_attn = torch.randn(attn5.shape, generator=g)
    g_emb = torch.randn(emb4.shape, generator=g)
    loss = (attn5 * g_attn).sum() + ((emb4 * g_emb).sum() if with_emb else 0)
    loss.backward()

    ctx = make_ctx({k: v for k, v in sd.items()}, training=True)
    xd = x.detach().cuda()
    kd = keep.cuda() if drop else None
    e_out, a_out = E.ltae_attention(ctx, xd, dates.cuda(), valid.view(-1).cuda(), "te", 16, 4, 256, 1000.0,
                                    0.1 if drop else 0.0, with_emb, 0, kd)
    # forward bar: 2e-6 absolute on the attention weights / 1e-5 relative on the embedding against the fp32 oracle, or -- where
    # the fp32 oracle itself is further than that from an fp64 evaluation (T = 61 with half of the frames padded: larger
    # normalised activations, sharper softmax) -- three times the oracle's own fp32 error (SURVEY.md 8c.4 criterion)
    with torch.no_grad():
        sd64 = {k: v.double() for k, v in sd.items()}
        emb64, attn64 = O.ltae_attention(x.detach().double(), dates, ~valid.bool(), sd64, "te", cfg,
                                         keep.double() if drop else None)
    a64 = attn64.view(16, B, h, h, T).permute(0, 1, 4, 2, 3)
    e64 = emb64.view(B, h, h, 256).permute(0, 3, 1, 2)
    a_err32 = float((attn5.detach().double() - a64).abs().max())
    assert float((a_out.cpu().double() - a64).abs().max()) <= max(3 * a_err32, 2e-6)
    if with_emb:
        e_err32 = rel(emb4, e64)
        assert rel(e_out, e64) <= max(3 * e_err32, 1e-5)
    del sd64, emb64, attn64, a64, e64
    ctx.tape.grads[a_out.data_ptr()] = g_attn.cuda().contiguous()
    if with_emb:
        ctx.tape.grads[e_out.data_ptr()] = g_emb.cuda().contiguous()
    ctx.tape.backward()
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 1e-4
    gmax = max(float(v.grad.norm()) for v in sdg.values())
    for k, v in sdg.items():
        ref = v.grad
        got = ctx.g[k].cpu()
        assert float((got - ref).norm()) <= 1e-4 * float(ref.norm()) + 1e-6 * gmax, k


@pytest.mark.parametrize("B,T,h,p,pad", [(2, 61, 128, 0.1, True), (2, 64, 128, 0.3, False), (8, 7, 64, 0.0, True), (2, 9, 128, 0.1, False)])
def test_ltae_attention_without_stored_weights_is_bit_identical(B, T, h, p, pad):
    """TimeUNet never reads the post-dropout attention weights (timeunet.py:176-178,204-205): with need_attn=False the
    register-resident forward stores attn_pre and the keep flags of the dropout as one bit per element (c2s_ltae_desc.keep_bits),
    and the backward kernels rebuild attn = attn_pre * keep from those.  Same arithmetic on the same values: embedding and
    EVERY gradient must equal the stored-weights path bit for bit (which the op test above pins to the oracle), with the RNG
    mask at p > 0 and p = 0, padded frames and T up to 64."""
    E, L = _engine()
    g = torch.Generator().manual_seed(29)
    C = 64
    sd = _ltae_state(C, g)
    x = torch.randn(B, T, C, h, h, generator=g)
    dates = (5 * torch.arange(T)[None] + torch.arange(B)[:, None]).long()
    valid = torch.ones(B, T, dtype=torch.int32)
    if pad:
        tb = max(T - 3, 1)
        valid[0, tb:] = 0
        x[0, tb:] = 0
        dates[0, tb:] = 0
    g_emb = torch.randn(B, 256, h, h, generator=g).cuda()
    xd, dd, vd = x.cuda(), dates.cuda(), valid.view(-1).cuda()
    d = L.LtaeDesc(B, T, C, h * h, 16, 256, 1e-5, p, 77, None, None)
    assert L.lib().c2s_ltae_attn_optional(ctypes.byref(d)) == 1 and L.lib().c2s_ltae_fwd_path(ctypes.byref(d)) == 2
    res = []
    for need in (True, False):
        ctx = make_ctx({k: v for k, v in sd.items()}, training=True)
        e_out, a_out = E.ltae_attention(ctx, xd, dd, vd, "te", 16, 4, 256, 1000.0, p, True, 77, None, need_attn=need)
        assert (a_out is None) == (not need)
        ctx.tape.grads[e_out.data_ptr()] = g_emb.clone()
        ctx.tape.backward()
        torch.cuda.synchronize()
        res.append((e_out.clone(), ctx.tape.grads[xd.data_ptr()].clone(), {k: v.clone() for k, v in ctx.g.items()}))
    (e1, gx1, p1), (e0, gx0, p0) = res
    assert torch.equal(e1, e0) and bool(torch.isfinite(gx1).all())
    assert torch.equal(gx1, gx0)
    for k in p1:
        assert torch.equal(p1[k], p0[k]), k
    # inference (no tape) without a reader of the masks stores neither tensor
    ctx = make_ctx({k: v for k, v in sd.items()}, training=False, tape=False)
    e_inf, a_inf = E.ltae_attention(ctx, xd, dd, vd, "te", 16, 4, 256, 1000.0, p, True, 77, None, need_attn=False)
    ctx2 = make_ctx({k: v for k, v in sd.items()}, training=False, tape=False)
    e_ref, a_ref = E.ltae_attention(ctx2, xd, dd, vd, "te", 16, 4, 256, 1000.0, p, True, 77, None)
    assert a_inf is None and a_ref is not None and torch.equal(e_inf, e_ref)


@pytest.mark.parametrize("B,T,C,H,h,pad", [(2, 5, 64, 32, 4, True), (1, 4, 64, 128, 16, False), (2, 3, 128, 16, 16, True)])
def test_temporal_aggregate_fwd_bwd(B, T, C, H, h, pad):
    E, L = _engine()
    g = torch.Generator().manual_seed(17)
    x = torch.randn(B, T, C, H, H, generator=g, requires_grad=True)
    attn = torch.softmax(torch.randn(16, B, T, h, h, generator=g), dim=2).requires_grad_(True)
    valid = torch.ones(B, T, dtype=torch.int32)
    if pad:
        valid[0, -1] = 0
    ref = O.temporal_aggregate(x, ~valid.bool(), attn)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({})
    xd, ad = x.detach().cuda(), attn.detach().cuda()
    out = E.temporal_aggregate(ctx, xd, ad, valid.view(-1).cuda(), 16)
    assert rel(out, ref) < 2e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 2e-6
    assert rel(ctx.tape.grads[ad.data_ptr()], attn.grad) < 1e-5


def test_pixel_group_norm_and_dropout():
    E, L = _engine()
    g = torch.Generator().manual_seed(19)
    B, C, h = 2, 128, 8
    x = torch.randn(B, C, h, h, generator=g, requires_grad=True)
    gamma = (1 + 0.3 * torch.randn(C, generator=g)).requires_grad_(True)
    beta = (0.1 * torch.randn(C, generator=g)).requires_grad_(True)
    keep = (torch.rand(B * h * h, C, generator=g) >= 0.2).float()
    xp = x.permute(0, 2, 3, 1).reshape(B * h * h, C)
    yp = F.group_norm(xp * keep / 0.8, 16, gamma, beta, 1e-5)
    ref = yp.view(B, h, h, C).permute(0, 3, 1, 2)
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    ctx = make_ctx({"on.weight": gamma.detach(), "on.bias": beta.detach()})
    xd = x.detach().cuda()
    o = E.dropout_nchw(ctx, xd, 0.2, 0, keep.cuda())
    out = E.pixel_group_norm(ctx, o, "on", 16)
    assert rel(out, ref) < 2e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 2e-5
    assert rel(ctx.g["on.weight"], gamma.grad) < 2e-5
    assert rel(ctx.g["on.bias"], beta.grad) < 2e-5


def test_cross_entropy_and_adam():
    E, L = _engine()
    g = torch.Generator().manual_seed(23)
    logits = torch.randn(3, 15, 32, 32, generator=g, requires_grad=True)
    y = torch.randint(0, 15, (3, 32, 32), generator=g)
    ref = O.cross_entropy(logits, y, 15)
    ref.backward()
    cw = torch.ones(15)
    cw[-1] = 0
    loss, gl = E.cross_entropy(logits.detach().cuda(), y.cuda(), cw.cuda(), E.Workspace(torch.device("cuda")), True)
    assert abs(float(loss) - float(ref)) < 1e-5 * abs(float(ref))
    assert rel(gl, logits.grad) < 1e-5
    # Adam: three steps against the oracle's restatement of torch.optim.Adam defaults
    p = {"w": torch.randn(1000, generator=g)}
    m, v = {"w": torch.zeros(1000)}, {"w": torch.zeros(1000)}
    pd, md, vd = p["w"].clone().cuda(), torch.zeros(1000).cuda(), torch.zeros(1000).cuda()
    for step in range(1, 4):
        gr = torch.randn(1000, generator=g)
        O.adam_step(p, {"w": gr}, m, v, step)
        E.adam_flat(pd, gr.cuda(), md, vd, step)
    assert rel(pd, p["w"]) < 1e-6


def test_frame_flags():
    E, L = _engine()
    x = torch.randn(2, 5, 10, 32, 32)
    x[0, 3:] = 0
    x[1, 4, :, 5, 5] = 0     # a few zeros inside a real frame do not make it padding
    v = E.frame_flags(x.cuda(), 0.0).cpu().view(2, 5)
    assert v.tolist() == [[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]]
    # several chunks per frame (10 x 128 x 128: ten blocks each); real frames that look padded at the head of EVERY chunk -- the
    # kernel leaves a chunk after its first 256 values only when one of them differs from the pad value
    y = torch.zeros(1, 6, 10, 128, 128)
    y[0, 0] = torch.randn(10, 128, 128)
    y[0, 1, 9, 127, 127] = 1.0                     # only the very last value
    y[0, 2, 0, 2, 0] = -3.0                        # value 256 of the first chunk
    y[0, 3] = 7.0                                  # a constant frame that is not the pad value
    y[0, 4, 5, 64, 1] = 1e-30                      # a tiny value in the middle
    for pad, want in ((0.0, [1, 1, 1, 1, 1, 0]), (7.0, [1, 1, 1, 0, 1, 1])):
        assert E.frame_flags(y.cuda(), pad).cpu().view(-1).tolist() == want, pad


@pytest.mark.parametrize("training", [False, True])
@pytest.mark.parametrize("hw", [8, 32, 64])   # inputs are chosen kink-free (oracle.relu_margin), see O.RELU_PROBE
def test_up_conv_block_fwd_bwd(training, hw):
    """UpConvBlock (reference conv.py:362-413) as one unit, BatchNorm in train and eval mode."""
    from oracle import seeded
    from crop2seg_amd.backbones import functional as Fn
    E, L = _engine()
    d_in, d_out, d_skip = 64, 32, 64
    ks = [("u.skip_conv.0.weight", (d_skip, d_skip, 1, 1)), ("u.skip_conv.0.bias", (d_skip,)),
          ("u.skip_conv.1.weight", (d_skip,)), ("u.skip_conv.1.bias", (d_skip,)),
          ("u.skip_conv.1.running_mean", (d_skip,)), ("u.skip_conv.1.running_var", (d_skip,)),
          ("u.skip_conv.1.num_batches_tracked", ()),
          ("u.up.0.weight", (d_in, d_out, 4, 4)), ("u.up.0.bias", (d_out,)),
          ("u.up.1.weight", (d_out,)), ("u.up.1.bias", (d_out,)),
          ("u.up.1.running_mean", (d_out,)), ("u.up.1.running_var", (d_out,)), ("u.up.1.num_batches_tracked", ()),
          ("u.conv1.conv.0.weight", (d_out, d_out + d_skip, 3, 3)), ("u.conv1.conv.0.bias", (d_out,)),
          ("u.conv1.conv.1.weight", (d_out,)), ("u.conv1.conv.1.bias", (d_out,)),
          ("u.conv1.conv.1.running_mean", (d_out,)), ("u.conv1.conv.1.running_var", (d_out,)),
          ("u.conv1.conv.1.num_batches_tracked", ()),
          ("u.conv2.conv.0.weight", (d_out, d_out, 3, 3)), ("u.conv2.conv.0.bias", (d_out,)),
          ("u.conv2.conv.1.weight", (d_out,)), ("u.conv2.conv.1.bias", (d_out,)),
          ("u.conv2.conv.1.running_mean", (d_out,)), ("u.conv2.conv.1.running_var", (d_out,)),
          ("u.conv2.conv.1.num_batches_tracked", ())]
    sd = seeded.make_state(ks, 31, "tame")
    cfg = O.BackboneConfig()
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    for seed in range(33, 33 + 200):
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(2, d_in, hw // 2, hw // 2, generator=g)
        skip = torch.randn(2, d_skip, hw, hw, generator=g)
        if O.relu_margin(lambda: O.up_conv_block(x.double(), skip.double(), sd64, "u", cfg, training, None)) > 2e-6:
            break
    x.requires_grad_(True)
    skip.requires_grad_(True)
    pnames = [k for k in sd if sd[k].is_floating_point() and "running" not in k]
    sdg = {k: (v.clone().requires_grad_(True) if k in pnames else v.clone()) for k, v in sd.items()}
    ref = O.up_conv_block(x, skip, sdg, "u", cfg, training, O.BNState())
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout)
    params = {k: sd[k] for k in pnames}
    bufs = {k: sd[k] for k in sd if k not in pnames}
    ctx = make_ctx(params, bufs, training=training)
    xd, sk = x.detach().cuda(), skip.detach().cuda()
    out = Fn.up_conv_block(ctx, xd, sk, "u", Fn.BackboneSpec())
    assert rel(out, ref) < 5e-6
    seed_backward(ctx, out, gout)
    assert rel(ctx.tape.grads[xd.data_ptr()], x.grad) < 5e-5
    assert rel(ctx.tape.grads[sk.data_ptr()], skip.grad) < 5e-5
    gmax = max(float(sdg[k].grad.norm()) for k in pnames)
    for k in pnames:
        ref_g = sdg[k].grad
        assert float((ctx.g[k].cpu() - ref_g).norm()) <= 5e-5 * float(ref_g.norm()) + 1e-6 * gmax, k


BX_CASES = [
    # N, C0, C1, Cout, H, W
    (2, 64, 0, 64, 128, 128),
    (3, 64, 0, 128, 16, 16),
    (2, 32, 0, 15, 32, 32),
    (2, 32, 64, 32, 32, 32),      # concatenated sources
    (2, 128, 0, 128, 4, 4),
    (3, 64, 0, 64, 8, 8),
    (1, 24, 0, 40, 24, 40),       # ragged
    (2, 64, 0, 64, 2, 2),
]


@pytest.mark.parametrize("case", BX_CASES)
def test_conv3x3_bf16x3_fwd_bwd(case):
    """Opt-in split-precision mode (engine.CONV_MODE = 'bf16x3'): 3 bf16 MFMAs per fp32 product, fp32 accumulate.
    Bar: 5e-5 relative (observed ~1e-5) on forward, data gradient (reflect adjoint inside the kernel) and, through
    the exact-fp32 wgrad kernel, the weight gradient."""
    E, L = _engine()
    N, C0, C1, Cout, H, W = case
    Cin = C0 + C1
    g = torch.Generator().manual_seed(sum(case))
    a = torch.randn(N, C0, H, W, generator=g, requires_grad=True)
    s2 = torch.randn(N, C1, H, W, generator=g, requires_grad=True) if C1 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.randn(Cout, generator=g, requires_grad=True)
    valid = torch.ones(N, dtype=torch.int32)
    if N >= 3:
        valid[1] = 0
    keep = valid.bool()
    xin = torch.cat([a, s2], 1) if C1 else a
    ref = O.conv2d(xin[keep].double(), w.double(), b.double(), 1, 1, "reflect")
    gout = torch.randn(ref.shape, generator=g)
    ref.backward(gout.double())
    old = E.CONV_MODE
    E.CONV_MODE = "bf16x3"
    try:
        ctx = make_ctx({"w": w.detach(), "b": b.detach()})
        ad = a.detach().cuda()
        sd = s2.detach().cuda() if C1 else None
        srcs = [ad, sd] if C1 else [ad]
        out = E.conv2d(ctx, srcs, "w", "b", 3, 1, 1, L.PAD_REFLECT, valid.cuda() if N >= 3 else None)
        assert rel(out[keep.cuda()], ref) < 5e-5
        gfull = torch.zeros(N, *ref.shape[1:])
        gfull[keep] = gout
        seed_backward(ctx, out, gfull)
    finally:
        E.CONV_MODE = old
    assert rel(ctx.tape.grads[ad.data_ptr()][keep.cuda()], a.grad[keep]) < 5e-5
    if C1:
        assert rel(ctx.tape.grads[sd.data_ptr()][keep.cuda()], s2.grad[keep]) < 5e-5
    assert rel(ctx.g["w"], w.grad) < 5e-5


def test_pack_plan_matches_individual_packs():
    """From the second step on all weight packs (tap-major and Winograd) run as one launch from a job table built during
    the first step: the convolution results must be bit-identical to the individually packed ones."""
    E, L = _engine()
    g = torch.Generator().manual_seed(23)
    dev = torch.device("cuda")
    params = {"w3": (torch.randn(64, 64, 3, 3, generator=g) / 24).to(dev), "b3": torch.randn(64, generator=g).to(dev),
              "w4": (torch.randn(64, 64, 4, 4, generator=g) / 32).to(dev), "b4": torch.randn(64, generator=g).to(dev)}
    x = torch.randn(2, 64, 32, 32, generator=g).to(dev)
    ws = E.Workspace(dev)

    def run():
        grads = {k: torch.zeros_like(v) for k, v in params.items()}
        ctx = E.Ctx(params, {}, grads, ws, True, E.Tape())
        y3 = E.conv2d(ctx, [x], "w3", "b3", 3, 1, 1, L.PAD_REFLECT, None)          # Winograd forward + data gradient packs
        y4 = E.conv2d(ctx, [y3], "w4", "b4", 4, 2, 1, L.PAD_REFLECT, None)         # direct forward, transposed data gradient packs
        ctx.tape.grads[y4.data_ptr()] = torch.ones_like(y4)
        ctx.tape.backward()
        return y3.clone(), y4.clone(), ctx.tape.grads[x.data_ptr()].clone(), grads["w3"].clone(), grads["w4"].clone()

    first = run()
    assert ws.pack_plan is None and len(ws.pack_record) >= 4
    ws.finalize_pack_plan()
    assert ws.pack_plan is not None and ws.pack_plan["njobs"] >= 4
    second = run()
    for a, b in zip(first, second):
        assert torch.equal(a, b)


@pytest.mark.parametrize("shape,use_valid", [((4, 64, 32, 32), True), ((3, 128, 8, 8), False), ((2, 64, 128, 128), True),
                                             ((2, 256, 6, 10), False)])
def test_squeeze_excite_fwd_bwd(shape, use_valid):
    """SqueezeAndExcitation (squeeze_and_excitation.py:7-30) forward and backward against torch autograd on the formula."""
    E, L = _engine()
    N, C, H, W = shape
    g = torch.Generator().manual_seed(17)
    x = torch.randn(shape, generator=g).requires_grad_(True)
    w1 = (torch.randn(C // 16, C, generator=g) * 0.3).requires_grad_(True)
    w2 = (torch.randn(C, C // 16, generator=g) * 0.5).requires_grad_(True)
    valid = torch.ones(N, dtype=torch.int32)
    if use_valid:
        valid[1] = 0
    keep = valid.bool()
    xs = x[keep]
    s = torch.sigmoid(F.linear(F.relu(F.linear(xs.mean(dim=(2, 3)), w1)), w2))
    y = xs * s[:, :, None, None]
    gout = torch.randn(y.shape, generator=g)
    y.backward(gout)
    ctx = make_ctx({"m.sae.1.weight": w1.detach(), "m.sae.3.weight": w2.detach()})
    xd = x.detach().cuda()
    out = E.squeeze_excite(ctx, xd, "m", valid.cuda() if use_valid else None, 0.0)
    assert rel(out[keep.cuda()], y) < 2e-6
    if use_valid:
        assert float(out[1].abs().max()) == 0.0
    gfull = torch.zeros(shape)
    gfull[keep] = gout
    seed_backward(ctx, out, gfull)
    assert rel(ctx.tape.grads[xd.data_ptr()][keep.cuda()], x.grad[keep]) < 5e-6
    assert rel(ctx.g["m.sae.1.weight"], w1.grad) < 2e-5 and rel(ctx.g["m.sae.3.weight"], w2.grad) < 2e-5
