"""Pin the CPU oracle (oracle/crop2seg_oracle.py) against outputs of the imported reference
(tests/golden/*.npz, written by oracle/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import golden_names
from oracle import crop2seg_oracle as O

# structurally-zero gradients: softmax shift invariance / bias followed by BatchNorm
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


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_forward(goldens, name):
    g = goldens(name)
    bn = O.BNState()
    with torch.no_grad():
        kw = g.dropout_kwargs()
        if g.cfg.model == "wtae":
            kw.pop("mlp_keep", None)
        logits, att = O.forward(g.sd, g.x, g.dates, g.cfg, training=g.training, bn=bn, **kw)
        loss = O.cross_entropy(logits, g.y, g.cfg.out_conv[-1])
        if g.cfg.add_boundary_loss:          # second head + focal term (utae.py:236-238; src/learning/utils.py:283-285,318-324)
            from oracle import tail_oracle as TO
            out_b = O.LAST_BOUNDARY[0]
            ref_b = torch.from_numpy(g.z["logits_b"])
            assert (out_b - ref_b).abs().max() <= 2e-5 * ref_b.abs().max()
            loss = loss + TO.focal_ce(out_b, TO.boundary_target(g.y, 15), 2.0)
    ref_logits = torch.from_numpy(g.z["logits"])
    ref_att = torch.from_numpy(g.z["att"])
    scale = ref_logits.abs().max()
    # train mode under weight_init-style BN gains is ill-conditioned: the fp32 reference differs from
    # ITSELF by 1e-4 between 1 and 8 CPU threads
    tol = 5e-4 if (g.training and g.meta["flavour"] == "wi") else 2e-5
    assert (logits - ref_logits).abs().max() <= tol * scale, (logits - ref_logits).abs().max() / scale
    assert (att - ref_att).abs().max() <= 2e-6
    assert abs(float(loss) - float(g.z["loss"])) <= 1e-5 * abs(float(g.z["loss"]))
    if not g.training:
        # argmax class map bit-exact (ties -> lowest index, like torch.argmax)
        assert torch.equal(logits.argmax(1), ref_logits.argmax(1))
    else:
        for k in g.z.files:
            if k.startswith("bn/") and not g.dummy_pass_pollutes(k[3:]):
                ref = torch.from_numpy(g.z[k])
                got = bn.updates[k[3:]]
                assert (got - ref).abs().max() <= 1e-5 * max(1.0, float(ref.abs().max())), k


@pytest.mark.parametrize("name", golden_names())
def test_oracle_matches_reference_backward(goldens, name):
    g = goldens(name)
    kw = g.dropout_kwargs()
    if g.cfg.model == "wtae":
        kw.pop("mlp_keep", None)
    _, loss, grads, _ = O.loss_and_grads(g.sd, g.x, g.dates, g.y, g.cfg, g.training, **kw)
    names = g.grad_names()
    if not names:
        pytest.skip("forward-only fixture (oracle/make_golden.py says why)")
    assert set(names) == set(grads.keys())
    gmax = max(float(g.z[f"grad/{n}/norm"]) for n in names)
    # eval mode: tight against the reference's gradients.  train mode: fp64-anchored criterion (Golden.fp64_anchor).
    for n in names:
        if g.training:
            g.check_train_grad(n, grads[n], gmax, _abs_only(n, True))
            continue
        err, scale, ref_norm, got_norm = g.check_grad(n, grads[n])
        if _abs_only(n, False) or ref_norm < 1e-7 * gmax:
            assert err <= 1e-5 * gmax, (n, err, gmax)
        else:
            assert err <= 2e-4 * scale + 1e-7 * gmax, (n, err / max(scale, 1e-30))


def test_oracle_fp64_runs(goldens):
    """The functional oracle runs in float64 (the reference cannot: temp_shared_block.py:25)."""
    g = goldens("utae_eval_pad_wi")
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in g.sd.items()}
    with torch.no_grad():
        l64, _ = O.forward(sd64, g.x.double(), g.dates, g.cfg)
    ref = torch.from_numpy(g.z["logits"]).double()
    assert l64.dtype == torch.float64
    assert (l64 - ref).abs().max() <= 1e-4 * ref.abs().max()


def test_padded_frames_semantics():
    """A padded frame leaves the encoder as exactly pad_value and gets exactly zero attention
    (SURVEY Appendix N.5)."""
    g = torch.Generator().manual_seed(0)
    cfg = O.BackboneConfig()
    x = torch.randn(1, 3, 10, 16, 16, generator=g)
    x[0, 2] = 0
    from oracle import seeded
    ks = [("in_conv.conv.conv.0.weight", (64, 10, 3, 3)), ("in_conv.conv.conv.0.bias", (64,)),
          ("in_conv.conv.conv.1.weight", (64,)), ("in_conv.conv.conv.1.bias", (64,)),
          ("in_conv.conv.conv.3.weight", (64, 64, 3, 3)), ("in_conv.conv.conv.3.bias", (64,)),
          ("in_conv.conv.conv.4.weight", (64,)), ("in_conv.conv.conv.4.bias", (64,))]
    sd = seeded.make_state(ks, 3, "tame")
    out = O.conv_block(x, sd, "in_conv", 2, "group", cfg, False, None, 0.0)
    assert torch.equal(out[0, 2], torch.zeros_like(out[0, 2]))
    assert out[0, :2].abs().sum() > 0


def test_oracle_adam_matches_torch_optim():
    """O.adam_step is the optimiser the HIP Adam kernel is judged against: pin it to torch.optim.Adam itself
    (defaults of the reference's train.py:454: lr 1e-3, betas (0.9, 0.999), eps 1e-8, no weight decay)."""
    g = torch.Generator().manual_seed(3)
    shapes = {"a": (64, 10, 3, 3), "b": (64,), "c": (1000,)}
    p_ref = {k: torch.nn.Parameter(torch.randn(s, generator=g)) for k, s in shapes.items()}
    p = {k: v.detach().clone() for k, v in p_ref.items()}
    m = {k: torch.zeros(s) for k, s in shapes.items()}
    v = {k: torch.zeros(s) for k, s in shapes.items()}
    opt = torch.optim.Adam(list(p_ref.values()), lr=1e-3)
    for step in range(1, 8):
        grads = {k: torch.randn(s, generator=g) * (10.0 ** (step % 3 - 1)) for k, s in shapes.items()}
        for k in shapes:
            p_ref[k].grad = grads[k].clone()
        opt.step()
        O.adam_step(p, grads, m, v, step)
        for k in shapes:
            assert float((p[k] - p_ref[k].detach()).abs().max()) <= 2e-7 * max(1.0, float(p_ref[k].abs().max())), (k, step)
