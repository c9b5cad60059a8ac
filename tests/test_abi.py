"""CPU checks of the drop-in boundary: the C-ABI library builds/loads and exports every symbol the header
declares; the model classes reproduce the reference's state_dict layout; the product has no CPU fallback."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "c2s_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(c2s_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from crop2seg_amd import _lib, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/c2s_hip.h but not exported"
    assert sorted(_lib.SIGNATURES) == syms, "ctypes binding and header disagree"
    assert _lib.lib().c2s_abi_version() == 4


def test_argument_validation_without_gpu():
    """Entry points validate before touching the device: bad descriptors return C2S_EINVAL with a message."""
    from crop2seg_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc(1, 8, 0, 8, 8, 8, 30, 8, 8, 8, 8, 3, 3, 1, 1, 1, 1, 1, 1, 0, 0, 0)   # CoutP not a multiple of 32
    rc = L.c2s_conv_igemm(ctypes.byref(d), 16, None, 16, None, 16, None, None)
    assert rc == -1 and b"CoutP" in L.c2s_last_error()
    nd = _lib.NormDesc(2, 10, 64, 0, 4, 1, 1e-5, 0.1)                                       # 10 % 4 != 0
    assert L.c2s_norm_fwd(ctypes.byref(nd), 16, 16, 16, None, None, None, 16, 16, None, 16, 1, 16, 1 << 20, None, 0.0, None) == -1


def test_smallcin_dispatch_predicate_without_gpu():
    """The first-layer kernel's host-side predicate: 3x3/s1/pad1, one source of <= 10 channels, Cout % 64 == 0,
    H % 8 == 0, W % 32 == 0, plain output placement."""
    from crop2seg_amd import _lib
    L = _lib.lib()

    def desc(c0=10, cout=64, h=128, w=128, k=3, s=1, pad=1, c1=0, acc=0):
        return _lib.ConvDesc(4, c0, c1, h, w, cout, cout, h, w, h, w, k, k, s, pad, pad, _lib.PAD_REFLECT, 1, 1, 0, 0, acc)

    assert L.c2s_conv3x3_smallcin_supported(ctypes.byref(desc())) == 1
    assert L.c2s_conv3x3_smallcin_supported(ctypes.byref(desc(c0=4, h=8, w=32))) == 1
    for bad in (desc(c0=11), desc(cout=32), desc(h=12), desc(w=48), desc(c1=2), desc(acc=1), desc(k=1, pad=0)):
        assert L.c2s_conv3x3_smallcin_supported(ctypes.byref(bad)) == 0
    rc = L.c2s_conv3x3_smallcin(ctypes.byref(desc(c0=11)), 16, 16, None, 16, None, None)
    assert rc == -1 and b"smallcin" in L.c2s_last_error()


def test_winograd_family_predicates_without_gpu():
    """Host-side predicates and argument checks of the round-3 Winograd kernels (8-wave F(2x2,3x3); F(2x2,2x2) forward and
    data gradient of the 4x4 stride-2 convolution): what they take, what the engine must route elsewhere."""
    from crop2seg_amd import _lib
    L = _lib.lib()

    def d3(n=4, c0=64, c1=0, cout=64, h=128, w=128):
        return _lib.ConvDesc(n, c0, c1, h, w, cout, (cout + 63) // 64 * 64, h, w, h, w, 3, 3, 1, 1, 1, _lib.PAD_REFLECT, 1, 1, 0, 0, 0)

    assert L.c2s_conv3x3_winograd16_supported(ctypes.byref(d3())) == 1
    assert L.c2s_conv3x3_winograd16_supported(ctypes.byref(d3(c0=32, c1=64, cout=72, h=12, w=40))) == 1
    for bad in (d3(w=16), d3(h=4), d3(c0=24), d3(c0=36, c1=8), d3(n=70000)):      # narrow / low planes, < 4 chunks, ragged first source
        assert L.c2s_conv3x3_winograd16_supported(ctypes.byref(bad)) == 0
    assert L.c2s_conv3x3_winograd16(ctypes.byref(d3(w=16)), 16, None, 16, None, 16, None, None) == -1
    assert b"winograd16" in L.c2s_last_error()

    def d4(n=4, cin=64, cout=64, hin=128, win=128, c1=0):
        return _lib.ConvDesc(n, cin, c1, hin, win, cout, (cout + 63) // 64 * 64, hin // 2, win // 2, hin // 2, win // 2, 4, 4, 2, 1, 1,
                             _lib.PAD_REFLECT, 1, 1, 0, 0, 0)

    assert L.c2s_conv4x4s2_winograd_supported(ctypes.byref(d4())) == 1
    assert L.c2s_conv4x4s2_winograd_supported(ctypes.byref(d4(cin=8, cout=72, hin=24, win=80))) == 1
    for bad in (d4(cin=7), d4(cin=6), d4(win=32), d4(hin=8), d4(c1=8)):           # odd / too few channels, output plane < 32 wide or < 8 high
        assert L.c2s_conv4x4s2_winograd_supported(ctypes.byref(bad)) == 0
    assert L.c2s_conv4x4s2_winograd(ctypes.byref(d4(win=32)), 16, 16, None, 16, None, None) == -1

    def dg(n=4, kc=64, cs=64, ho=64, wo=64, radj=1):                               # the gradient launch: gy [kc, ho, wo] -> gx [cs, 2 ho, 2 wo]
        return _lib.ConvDesc(n, kc, 0, ho, wo, cs, (cs + 63) // 64 * 64, 2 * ho, 2 * wo, 2 * ho, 2 * wo, 4, 4, 2, 1, 1, _lib.PAD_ZEROS,
                             1, 1, 0, 0, 0, radj)

    assert L.c2s_conv4x4s2_dgrad_winograd_supported(ctypes.byref(dg())) == 1
    assert L.c2s_conv4x4s2_dgrad_winograd_supported(ctypes.byref(dg(kc=72, cs=10, ho=12, wo=40, radj=0))) == 1
    for bad in (dg(kc=16), dg(kc=60), dg(wo=16), dg(ho=6), dg(n=40000)):
        assert L.c2s_conv4x4s2_dgrad_winograd_supported(ctypes.byref(bad)) == 0
    assert L.c2s_conv4x4s2_dgrad_winograd(ctypes.byref(dg(kc=16)), 16, 16, 16, None, None) == -1
    assert L.c2s_s2wino_packed_floats(64, 64) == 32 * 2 * 4 * 64 * 12 and L.c2s_s2dgrad_packed_floats(64, 64) == 4 * 64 * 64 * 12
    assert L.c2s_winograd16_packed_floats(64, 64) == 8 * 8 * 64 * 16       # 16 points per (c, o): no padding since round 4


@pytest.mark.parametrize("name", ["utae_eval_pad_wi", "timeunet_eval_pad_wi", "wtae_eval_pad_wi"])
def test_state_dict_layout_matches_reference(goldens, name):
    import crop2seg_amd as C2S
    g = goldens(name)
    cls = {"utae": C2S.UTAE, "timeunet": C2S.TimeUNet_v1, "wtae": C2S.WTAE}[g.cfg.model]
    net = cls(input_dim=10, out_conv=[32, 15])
    assert [(k, tuple(v.shape)) for k, v in net.state_dict().items()] == g.key_shapes
    net.load_state_dict(g.sd)     # strict
    n_params = sum(p.numel() for p in net.parameters())
    assert n_params == {"utae": 1085805, "timeunet": 1052589, "wtae": 1114157}[g.cfg.model]


def test_weight_init_dispatch_matches_reference_types():
    import crop2seg_amd as C2S
    torch.manual_seed(0)
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    before = {k: v.clone() for k, v in net.state_dict().items()}
    net.apply(C2S.weight_init)
    after = net.state_dict()
    changed = {k for k in after if after[k].dtype.is_floating_point and not torch.equal(after[k], before[k])}
    # GroupNorm affine and Q are untouched by weight_init (reference weight_init.py has no branch for them)
    assert "temporal_encoder.attention_head.Q" not in changed
    assert "in_conv.conv.conv.1.weight" not in changed
    assert "in_conv.conv.conv.0.weight" in changed and "up_blocks.0.up.0.weight" in changed
    assert "temporal_encoder.mlp.2.weight" in changed and "temporal_encoder.inconv.bias" in changed


def test_no_cpu_fallback():
    import crop2seg_amd as C2S
    net = C2S.UTAE(input_dim=10, out_conv=[32, 15])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        net(torch.zeros(1, 2, 10, 32, 32), batch_positions=torch.zeros(1, 2, dtype=torch.long))


def test_offdefault_flags_fail_loudly():
    import crop2seg_amd as C2S
    with pytest.raises(ValueError, match="divisible"):          # the reference's constructor raises the same way: the GroupNorm(4)
        C2S.UTAE(input_dim=10, out_conv=[32, 15], use_mbconv=True)     # head of MBConvBlock needs a multiple of 4 classes
    assert len(C2S.UTAE(input_dim=10, out_conv=[32, 20], use_mbconv=True).state_dict()) == 348
    assert len(C2S.WTAE(input_dim=10, out_conv=[32, 20], use_mbconv=True).state_dict()) == 435
    with pytest.raises(NotImplementedError):
        C2S.WTAE(input_dim=10, agg_mode="max")
    # compiled-in limits surface in the constructor, naming the limit (not as a C-ABI code from the first forward)
    with pytest.raises(NotImplementedError, match="n_head=16"):
        C2S.UTAE(input_dim=10, n_head=8)
    with pytest.raises(NotImplementedError, match="d_k=4"):
        C2S.TimeUNet_v1(input_dim=10, d_k=8)
    with pytest.raises(NotImplementedError, match="str_conv_k=4"):
        C2S.WTAE(input_dim=10, str_conv_k=3, str_conv_s=1, str_conv_p=1)
    with pytest.raises(NotImplementedError, match="multiple of 64"):
        C2S.UTAE(input_dim=10, encoder_widths=[64, 64, 64, 96], decoder_widths=[32, 32, 64, 96])
    # use_abs_rel_enc together with use_doy / add_linear builds both encoders, as the reference does (tae.py:407-423)
    sd = C2S.UTAE(input_dim=10, use_abs_rel_enc=True, use_doy=True).state_dict()
    assert tuple(sd["temporal_encoder.positional_encoder.fc.weight"].shape) == (16, 365)
    assert tuple(sd["temporal_encoder.positional_encoder_abs.fc.weight"].shape) == (16, 365)
    sd = C2S.TimeUNet_v1(input_dim=10, use_abs_rel_enc=True, add_linear=True).state_dict()
    assert tuple(sd["temporal_encoder.positional_encoder.fc.weight"].shape) == (256, 256)
    assert tuple(sd["temporal_encoder.positional_encoder_abs.fc.weight"].shape) == (16, 365)


def test_positional_encoder_flags_extend_the_state_dict_like_the_reference():
    """use_doy / use_abs_rel_enc / add_linear (tae.py:404-430): the encoder's parameters sit between inconv and the attention
    head, as in the reference's constructor; num_queries > 1 is accepted (Q [16,n,4]) and the forward raises as the
    reference's own forward does."""
    import crop2seg_amd as C2S
    base = list(C2S.UTAE(input_dim=10).state_dict())
    at = base.index("temporal_encoder.inconv.bias") + 1
    for kw, extra in ((dict(use_doy=True), [("temporal_encoder.positional_encoder.fc.weight", (16, 365)),
                                            ("temporal_encoder.positional_encoder.fc.bias", (16,))]),
                      (dict(use_abs_rel_enc=True), [("temporal_encoder.positional_encoder_abs.fc.weight", (16, 365)),
                                                    ("temporal_encoder.positional_encoder_abs.fc.bias", (16,))]),
                      (dict(add_linear=True), [("temporal_encoder.positional_encoder.fc.weight", (256, 256)),
                                               ("temporal_encoder.positional_encoder.fc.bias", (256,))]),
                      (dict(add_linear=True, use_doy=True), [("temporal_encoder.positional_encoder.fc.weight", (256, 256)),
                                                             ("temporal_encoder.positional_encoder.fc.bias", (256,))])):
        for cls in (C2S.UTAE, C2S.WTAE, C2S.TimeUNet_v1):
            sd = cls(input_dim=10, **kw).state_dict()
            b0 = list(cls(input_dim=10).state_dict())
            i = b0.index("temporal_encoder.inconv.bias") + 1
            assert list(sd) == b0[:i] + [k for k, _ in extra] + b0[i:], (cls.__name__, kw)
            assert all(tuple(sd[k].shape) == s for k, s in extra)
    net = C2S.UTAE(input_dim=10, num_queries=2)
    assert tuple(net.state_dict()["temporal_encoder.attention_head.Q"].shape) == (16, 2, 4)


def test_optional_heads_extend_the_state_dict_like_the_reference():
    """add_boundary_loss adds boundary_conv = ConvBlock([dec0, 32, 2]) after out_conv (utae.py:195-198, wtae.py:215-218);
    TimeUNet_v1 swallows the flag through **kwargs (timeunet.py:19-45)."""
    import crop2seg_amd as C2S
    base = list(C2S.UTAE(input_dim=10, out_conv=[32, 15]).state_dict())
    withb = list(C2S.UTAE(input_dim=10, out_conv=[32, 15], add_boundary_loss=True, agg_mode="att_mean").state_dict())
    extra = withb[len(base):]
    assert withb[:len(base)] == base and len(extra) == 14 and all(k.startswith("boundary_conv.conv.conv.") for k in extra)
    assert len(C2S.WTAE(input_dim=10, add_boundary_loss=True, agg_mode="mean").state_dict()) == 185 + 14
    assert len(C2S.TimeUNet_v1(input_dim=10, add_boundary_loss=True, use_mbconv=True).state_dict()) == 158
    assert C2S.UTAE(input_dim=10, encoder=True).spec.return_maps


def test_get_model_mapping():
    from crop2seg_amd.learning.utils import default_config, get_model
    import crop2seg_amd as C2S
    assert isinstance(get_model(default_config("utae")), C2S.UTAE)
    assert isinstance(get_model(default_config("wtae")), C2S.WTAE)
    assert isinstance(get_model(default_config("timeunet")), C2S.TimeUNet_v1)
    m = get_model(default_config("utae"))
    assert m.spec.out_conv == [32, 15] and m.spec.padding_mode == "reflect"


@pytest.mark.gpu
def test_wgrad_winograd_launches_without_c2s_init():
    """c2s_conv_wgrad's Winograd branch needs the raised dynamic-LDS limit; a caller that never ran c2s_init
    (fresh process, raw ctypes) must still get a working launch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "abi_noinit_worker.py")], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0 and "ABI_NOINIT_OK" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
