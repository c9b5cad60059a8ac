"""Whole-model GPU parity: crop2seg_amd (HIP kernels through the C ABI) against the golden vectors produced by the
imported reference (tests/golden, oracle/make_golden.py) and against the CPU oracle.

Protocol: eval forward <= 1e-3 (rel. to max |logit|) with bit-exact argmax map; gradients per tensor
<= 1e-3 relative except structurally-zero ones (absolute tolerance); train mode with dropout off or with injected keep
masks.  The kernels are fp32 end to end, so observed errors are ~1e-5; the asserted bars are the protocol's."""
import numpy as np
import pytest
import torch

from conftest import golden_names

pytestmark = pytest.mark.gpu


def build(g):
    import crop2seg_amd as C2S
    from crop2seg_amd.backbones import functional as Fn
    cls = {"utae": C2S.UTAE, "timeunet": C2S.TimeUNet_v1, "wtae": C2S.WTAE}[g.cfg.model]
    net = cls(**{**dict(input_dim=10, out_conv=[32, 15]), **g.ctor})
    got = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    assert got == g.key_shapes, "state_dict layout differs from the reference"
    net.load_state_dict(g.sd)
    net = net.cuda()
    drop = Fn.DropoutState()
    if g.attn_keep is not None:
        drop.attn_keep = g.attn_keep.cuda().contiguous()
    if g.mlp_keep is not None:
        drop.mlp_keep = g.mlp_keep.cuda().contiguous()
    return net, drop


def _abs_only(name, training):
    if name.endswith("attention_head.fc1_k.bias"):
        return True
    if not training:
        return False
    if name.startswith("up_blocks") and name.endswith(".0.bias"):
        return True
    if name.startswith("out_conv") and name.endswith(".bias") and name.split(".")[-2] in ("0", "3"):
        return True
    return name in ("temporal_encoder.mlp.0.bias", "temporal_encoder.inconv.bias", "temporal_encoder.in_norm.bias")


@pytest.fixture(params=["f32", "bf16x3"])
def conv_mode(request):
    """Both convolution arithmetic modes must meet the same parity protocol."""
    from crop2seg_amd import engine as E
    old = E.CONV_MODE
    E.CONV_MODE = request.param
    yield request.param
    E.CONV_MODE = old


@pytest.mark.parametrize("name", golden_names())
def test_model_matches_reference(goldens, name, conv_mode):
    g = goldens(name)
    net, drop = build(g)
    net.train(g.training)
    if g.training and g.attn_keep is None:
        net.spec.attn_dropout = 0.0
        net.spec.mlp_dropout = 0.0
    x, dates, y = g.x.cuda(), g.dates.cuda(), g.y.cuda()
    outs = net(x, batch_positions=dates, return_att=True, dropout_state=drop)
    logits, att = outs[0], outs[-1]
    ref_logits = torch.from_numpy(g.z["logits"])
    ref_att = torch.from_numpy(g.z["att"])
    scale = float(ref_logits.abs().max())
    err = float((logits.detach().cpu() - ref_logits).abs().max()) / scale
    assert err <= 1e-3, err
    assert float((att.detach().cpu() - ref_att).abs().max()) <= (1e-4 if conv_mode == "f32" else 1e-3)
    if not g.training:
        assert torch.equal(logits.argmax(1).cpu(), ref_logits.argmax(1)), "argmax class map must be bit-exact"
    # loss + backward through torch autograd (drop-in path: loss.backward() on the model output)
    wgt = torch.ones(g.cfg.out_conv[-1], device="cuda")
    wgt[-1] = 0
    loss = torch.nn.functional.cross_entropy(logits, y, weight=wgt)
    if g.ctor.get("add_boundary_loss"):
        # boundary head (utae.py:236-238) + focal term of iterate() (src/learning/utils.py:283-285,318-324).  The drop-in
        # path hands torch autograd the head's logits; the focal loss itself is evaluated with torch ops here (its HIP
        # kernel is covered by tests/test_tail_gpu.py and by test_train_step_boundary_matches_oracle below)
        from oracle import tail_oracle as TO
        from crop2seg_amd.learning.losses import boundary_target
        out_b = outs[1]
        ref_b = torch.from_numpy(g.z["logits_b"])
        assert float((out_b.detach().cpu() - ref_b).abs().max()) <= 1e-3 * float(ref_b.abs().max())
        y_b = boundary_target(y)
        assert torch.equal(y_b.cpu(), TO.boundary_target(g.y, 15))
        loss = loss + TO.focal_ce(out_b, y_b, 2.0)
    assert abs(float(loss) - float(g.z["loss"])) <= 1e-3 * abs(float(g.z["loss"]))
    loss.backward()
    if conv_mode != "f32" and g.grad_names():
        # Split-precision mode: the north-star criteria (logits <= 1e-3, bit-exact argmax, loss) are asserted above.
        # Its ~1e-5 operand noise exceeds the 1e-5 ReLU-kink margin the fixtures were selected for, so per-tensor
        # gradient parity is not defined on them (a flipped kink moves a gradient by O(1/sqrt(N))); gradients are
        # checked at op level (tests/test_ops_gpu.py::test_conv3x3_bf16x3_fwd_bwd, 5e-5) and here only for sanity.
        params = dict(net.named_parameters())
        big = max(g.grad_names(), key=lambda n: float(g.z[f"grad/{n}/norm"]))
        e, sc, _, _ = g.check_grad(big, params[big].grad)
        assert e <= 0.2 * sc, (big, e / sc)
        return
    names = g.grad_names()
    params = dict(net.named_parameters())
    if not names:                      # forward-only fixture (train-mode MBConv: oracle/make_golden.py): gradients finite, buffers below
        assert all(bool(torch.isfinite(p.grad).all()) for p in params.values())
    gmax = max([float(g.z[f"grad/{n}/norm"]) for n in names] + [0.0])
    # eval mode: 1e-3 per tensor against the reference's gradients.  train mode: fp64-anchored criterion
    # (conftest.Golden.fp64_anchor).
    g.soft_violations = []
    for n in names:
        if g.training:
            g.check_train_grad(n, params[n].grad, gmax, _abs_only(n, True))
            continue
        e, sc, ref_norm, got_norm = g.check_grad(n, params[n].grad)
        if _abs_only(n, False) or ref_norm < 1e-6 * gmax:
            assert e <= 2e-5 * gmax, (n, e, gmax)
        else:
            assert e <= 1e-3 * sc + 1e-6 * gmax, (n, e / max(sc, 1e-30))
    if g.training:
        assert len(g.soft_violations) <= 0.1 * len(names), g.soft_violations
        sd = net.state_dict()
        for k in g.z.files:
            if k.startswith("bn/") and not g.dummy_pass_pollutes(k[3:]):
                ref = torch.from_numpy(g.z[k])
                assert float((sd[k[3:]].cpu() - ref).abs().max()) <= 1e-4 * max(1.0, float(ref.abs().max())), k


def test_train_step_matches_oracle(goldens):
    """Fused train step (HIP CE + backward + flat Adam, no autograd) == oracle step on the same batch."""
    from oracle import crop2seg_oracle as O
    from crop2seg_amd.learning.utils import TrainStep
    g = goldens("utae_train_p0_tame")
    net, drop = build(g)
    net.train()
    net.spec.attn_dropout = 0.0
    net.spec.mlp_dropout = 0.0
    step = TrainStep(net, num_classes=15)
    loss, logits = step(g.x.cuda(), g.dates.cuda(), g.y.cuda(), dropout_state=drop)
    _, ref_loss, grads, bn = O.loss_and_grads(g.sd, g.x, g.dates, g.y, g.cfg, True)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    names = O.parameter_names(g.sd)
    p = {n: g.sd[n].clone() for n in names}
    m = {n: torch.zeros_like(p[n]) for n in names}
    v = {n: torch.zeros_like(p[n]) for n in names}
    O.adam_step(p, grads, m, v, 1)
    got = dict(net.named_parameters())
    gmax = max(float(grads[n].norm()) for n in names)
    for n in names:
        # first Adam step moves every entry by lr * sign(g) (|g| >> eps): compare where the gradient is not ~0
        mask = grads[n].abs() > 1e-4 * gmax / max(1.0, grads[n].numel() ** 0.5)
        d_ref = (p[n] - g.sd[n])[mask]
        d_got = (got[n].detach().cpu() - g.sd[n])[mask]
        if d_ref.numel():
            assert float((d_ref - d_got).abs().max()) <= 2e-4, n


def test_train_step_surfaces_a_failed_normalisation_wait_and_recovers(monkeypatch):
    """TrainStep reads the sync-area error word wherever it synchronises anyway (bad_targets(), StepMeters readers,
    capture()): a step on a poisoned area raises there, the area is reset, the process falls back to the two-pass
    normalisation kernels, and the next step equals the oracle's."""
    from oracle import crop2seg_oracle as O
    from oracle import seeded
    import crop2seg_amd as C2S
    from crop2seg_amd import engine as E
    from crop2seg_amd.learning.metrics import StepMeters
    from crop2seg_amd.learning.utils import TrainStep
    monkeypatch.setattr(E, "ONEPASS_NORM", True)
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    ks = [(k, tuple(v.shape)) for k, v in net.state_dict().items()]
    sd = seeded.make_state(ks, 3, "tame")
    net.load_state_dict(sd)
    net = net.cuda().train()
    net.spec.attn_dropout = 0.0
    net.spec.mlp_dropout = 0.0
    x, dates, y = seeded.make_inputs(2, 4, 10, 32, 32, 91, [4, 3])
    step = TrainStep(net, num_classes=15)
    meters = StepMeters(15, ignore_index=-1).watch(step)
    xd, dd, yd = x.cuda(), dates.cuda(), y.cuda()
    loss, logits = step(xd, dd, yd, apply_update=False)
    assert step.bad_targets() == 0                           # healthy: no raise
    ref_logits, ref_loss, grads, _ = O.loss_and_grads(sd, x, dates, y, O.BackboneConfig(), True)
    assert abs(float(loss) - float(ref_loss)) < 1e-3 * abs(float(ref_loss))
    step.ws.bufs["sync"][:16].view(torch.int32)[3] = 1       # a wait gave up
    loss, logits = step(xd, dd, yd, apply_update=False)
    meters.update(logits, yd, loss)
    with pytest.raises(RuntimeError, match="one-pass normalisation wait gave up"):
        meters.loss_mean()
    assert E.ONEPASS_NORM is False
    loss, logits = step(xd, dd, yd, apply_update=False)      # two-pass kernels: same numbers as the oracle again
    assert step.bad_targets() == 0
    assert abs(float(loss) - float(ref_loss)) < 1e-3 * abs(float(ref_loss))
    assert float((logits.cpu() - ref_logits).abs().max()) < 1e-3 * float(ref_logits.abs().max())
    n = "in_conv.conv.conv.0.weight"
    assert float((step.grads[n].cpu() - grads[n]).norm() / grads[n].norm()) < 1e-3


@pytest.mark.parametrize("name,smoothing", [("utae_train_mean_boundary_tame", 0.0), ("utae_train_p0_tame", 0.1)])
def test_train_step_boundary_and_smoothing_match_oracle(goldens, name, smoothing):
    """TrainStep with the boundary head (CE + focal loss through the HIP loss kernels, both heads on the tape) and with
    label smoothing (train.py:466-468) == the oracle's loss and gradients on the same batch."""
    from oracle import crop2seg_oracle as O
    from crop2seg_amd.learning.utils import TrainStep
    g = goldens(name)
    net, drop = build(g)
    net.train()
    net.spec.attn_dropout = 0.0
    net.spec.mlp_dropout = 0.0
    step = TrainStep(net, num_classes=15, label_smoothing=smoothing)
    loss, logits = step(g.x.cuda(), g.dates.cuda(), g.y.cuda(), dropout_state=drop, apply_update=False)
    _, ref_loss, grads, _ = O.loss_and_grads(g.sd, g.x, g.dates, g.y, g.cfg, True, label_smoothing=smoothing)
    assert abs(float(loss) - float(ref_loss)) <= 1e-4 * abs(float(ref_loss))
    if smoothing == 0.0:
        assert abs(float(loss) - float(g.z["loss"])) <= 1e-4 * abs(float(g.z["loss"]))       # the reference's own value
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in g.sd.items()}
    _, _, g64, _ = O.loss_and_grads(sd64, g.x.double(), g.dates, g.y, g.cfg, True, label_smoothing=smoothing)
    gmax = max(float(v.norm()) for v in g64.values())
    for n, ref in g64.items():
        got = step.grads[n].detach().double().cpu()
        err, err32, sc = float((got - ref).norm()), float((grads[n].double() - ref).norm()), float(ref.norm())
        assert err <= max(3 * err32, 1e-3 * sc) + 2e-5 * gmax, (n, err / max(sc, 1e-30), err32 / max(sc, 1e-30))


def test_encoder_and_return_maps_outputs(goldens):
    """encoder=True / return_maps=True tuple conventions (utae.py:224-252): maps = [L-TAE output, up-block outputs...];
    the logits are unchanged by asking for the maps."""
    import crop2seg_amd as C2S
    g = goldens("utae_eval_nopad_tame")
    x, dates = g.x.cuda(), g.dates.cuda()
    plain = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    plain.load_state_dict(g.sd)
    plain = plain.cuda().eval()
    withmaps = C2S.UTAE(input_dim=10, out_conv=[32, 15], return_maps=True)
    withmaps.load_state_dict(g.sd)
    withmaps = withmaps.cuda().eval()
    enc = C2S.UTAE(input_dim=10, out_conv=[32, 15], encoder=True)
    enc.load_state_dict(g.sd)
    enc = enc.cuda().eval()
    with torch.no_grad():
        logits = plain(x, batch_positions=dates)
        l2, maps = withmaps(x, batch_positions=dates)
        last, maps_e = enc(x, batch_positions=dates)
    assert torch.equal(logits, l2) and len(maps) == 4 and len(maps_e) == 4
    assert [tuple(m.shape[1:]) for m in maps] == [(128, 2, 2), (64, 4, 4), (32, 8, 8), (32, 16, 16)]
    assert torch.equal(last, maps_e[-1]) and all(torch.equal(a, b) for a, b in zip(maps, maps_e))
    # gradients flow through the maps on the drop-in path
    enc.train()
    last, maps_e = enc(x, batch_positions=dates)
    (last.square().mean() + maps_e[1].mean()).backward()
    assert float(enc.in_conv.conv.conv[0].weight.grad.abs().max()) > 0
    assert enc.out_conv.conv.conv[0].weight.grad is None or float(enc.out_conv.conv.conv[0].weight.grad.abs().max()) == 0.0


def test_full_size_properties():
    """BASELINE configs[1] size (U-TAE B=4, T=32, 128x128): size-independent properties.
    (a) attention sums to 1 over T for every pixel/head; (b) logits are >= 0 (BN+ReLU head, reference utae.py:191);
    (c) padded frames get exactly zero attention; (d) batch independence in eval mode: sample 0 alone == sample 0 in
    the batch, bit for bit."""
    import crop2seg_amd as C2S
    torch.manual_seed(1)
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15]).cuda()
    net.apply(C2S.weight_init)
    net.eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(4, 32, 10, 128, 128, generator=g)
    dates = (5 * torch.arange(32))[None].repeat(4, 1)
    x[1, 27:] = 0
    dates[1, 27:] = 0
    x, dates = x.cuda(), dates.cuda()
    with torch.no_grad():
        logits, att = net(x, batch_positions=dates, return_att=True)
        l0, a0 = net(x[:1].contiguous(), batch_positions=dates[:1].contiguous(), return_att=True)
    assert logits.shape == (4, 15, 128, 128) and att.shape == (16, 4, 32, 16, 16)
    assert float((att.sum(dim=2) - 1).abs().max()) < 1e-5
    assert float(logits.min()) >= 0.0
    assert float(att[:, 1, 27:].abs().max()) == 0.0
    assert torch.equal(l0[0], logits[0]) and torch.equal(a0[:, 0], att[:, 0])


def test_hipgraph_replay_equals_eager(goldens):
    """The captured step (two hipGraphs: fwd+CE+bwd, Adam) replays bit-identically to eager launches, advances the
    Adam step count on the device, and draws a fresh dropout mask on every replay."""
    import copy
    from crop2seg_amd.learning.utils import TrainStep
    g = goldens("utae_train_p0_tame")
    x, dates, y = g.x.cuda(), g.dates.cuda(), g.y.cuda()

    def fresh(p_drop):
        net, _ = build(g)
        net.train()
        net.spec.attn_dropout = p_drop
        net.spec.mlp_dropout = p_drop
        return net, TrainStep(net, num_classes=15)

    net_e, step_e = fresh(0.0)
    for _ in range(3):
        loss_e, _ = step_e(x, dates, y)
    net_g, step_g = fresh(0.0)
    step_g(x, dates, y)                      # step 1 eager
    step_g.capture(x, dates, y)
    for _ in range(2):                       # steps 2, 3 replayed
        loss_g, _ = step_g.replay()
    torch.cuda.synchronize()
    assert float(loss_g) == float(loss_e)
    assert torch.equal(step_g.flat_param, step_e.flat_param), "replayed parameters differ from eager ones"
    assert int(step_g.step_dev) == 3
    for k, v in net_e.state_dict().items():
        assert torch.equal(v, net_g.state_dict()[k]), k
    # dropout on: consecutive replays must not reuse the mask
    net_d, step_d = fresh(0.3)
    step_d.capture(x, dates, y)
    step_d.lr = 0.0
    l1 = float(step_d.replay()[0])
    l2 = float(step_d.replay()[0])
    assert l1 != l2
