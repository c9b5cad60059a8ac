"""Forward passes of U-TAE / TimeUNet_v1 / W-TAE expressed over the HIP engine (crop2seg_amd.engine).

Each function mirrors one reference forward (cited) but runs every op as a hand-written kernel through
the C ABI; the engine's tape gives the backward.  Padded frames are handled with a device-side per-frame
flag array (no host sync, no boolean compaction): kernels skip padded frames and block outputs are filled
with pad_value for them, which reproduces TemporallySharedBlock.smart_forward
(reference src/backbones/temp_shared_block.py:18-47).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import torch

from .. import _lib
from .. import engine as E

Tensor = torch.Tensor


@dataclass
class BackboneSpec:
    model: str = "utae"
    input_dim: int = 10
    encoder_widths: List[int] = field(default_factory=lambda: [64, 64, 64, 128])
    decoder_widths: List[int] = field(default_factory=lambda: [32, 32, 64, 128])
    out_conv: List[int] = field(default_factory=lambda: [32, 20])
    str_conv_k: int = 4
    str_conv_s: int = 2
    str_conv_p: int = 1
    agg_mode: str = "att_group"
    encoder_norm: str = "group"
    n_head: int = 16
    d_model: int = 256
    d_k: int = 4
    pad_value: float = 0.0
    padding_mode: str = "reflect"
    conv_type: str = "2d"             # "depthwise_separable": in_conv and the down blocks use DepthwiseSeparableConv2D (utae.py:144,158)
    add_boundary_loss: bool = False   # second (2-class) head on the last decoder map (reference utae.py:195-198,236-244)
    encoder: bool = False             # return (last decoder map, maps) instead of logits (utae.py:233-234)
    return_maps: bool = False         # also return the decoder feature maps (utae.py:224-231)
    pe_period: float = 1000.0
    pe_mode: str = "rel"              # "doy" / "abs_rel" / "linear": the learnable positional encoders (tae.py:404-430)
    num_queries: int = 1              # > 1: accepted by the constructors, the forward raises as the reference's does
    add_squeeze_excit: bool = False   # SqueezeAndExcitation after in_conv and after every encoder down block (utae.py:145,159)
    use_mbconv: bool = False          # MBConv blocks instead of the classical conv blocks (utae.py:118-122; mbconv.py)
    attn_dropout: float = 0.1       # reference tae.py:816
    mlp_dropout: float = 0.2        # reference tae.py:361


@dataclass
class DropoutState:
    """Dropout randomness of one forward: RNG seeds (product) or explicit keep masks (parity tests)."""
    attn_seed: int = 0
    mlp_seed: int = 0
    attn_keep: Optional[Tensor] = None    # [n_head, P, T]
    mlp_keep: Optional[Tensor] = None     # [P, C']
    seed_dev: Optional[Tensor] = None     # device uint64 step counter mixed into both seeds (hipGraph replay)


def _mode(spec: BackboneSpec) -> int:
    return _lib.PAD_REFLECT if spec.padding_mode == "reflect" else _lib.PAD_ZEROS


def _norm_kind(norm: str) -> int:
    if norm not in ("group", "batch", "instance"):
        raise NotImplementedError(f"norm {norm!r}")
    return _lib.NORM_BATCH if norm == "batch" else _lib.NORM_GROUP


def conv_layer(ctx: E.Ctx, srcs: Sequence[Tensor], prefix: str, n_convs: int, norm: str, k: int, s: int, p: int,
               spec: BackboneSpec, valid: Optional[Tensor], residual: Optional[Tensor] = None,
               depthwise_separable: bool = False, need_input_grad: bool = True, add_squeeze: bool = False) -> Tensor:
    """ConvLayer: [conv -> norm -> ReLU] * n_convs (reference conv.py:29-96); the optional residual is added
    after the last ReLU (conv.py:292,410); add_squeeze appends SqueezeAndExcitation at Sequential index 3*n_convs
    (conv.py:90-91)."""
    mode = _mode(spec)
    x = list(srcs)
    y = None
    for i in range(n_convs):
        cp = f"{prefix}.conv.{3 * i}"
        if depthwise_separable:
            t = E.depthwise_conv2d(ctx, x[0], cp + ".depthwise.weight", k, s, p, mode, valid)
            y = E.conv2d(ctx, [t], cp + ".pointwise.weight", None, 1, 1, 0, _lib.PAD_ZEROS, valid)
            bias = None
        else:
            y = E.conv2d(ctx, x, cp + ".weight", cp + ".bias", k, s, p, mode, valid,
                         need_input_grad=need_input_grad or i > 0)
            bias = cp + ".bias"
        last = i == n_convs - 1
        # "instance" = nn.InstanceNorm2d (conv.py:54-55): one group per channel, no affine parameters, no running statistics
        y = E.norm_act(ctx, y, f"{prefix}.conv.{3 * i + 1}", _norm_kind(norm), y.shape[1] if norm == "instance" else 4, True,
                       residual if last else None, valid, spec.pad_value if valid is not None else 0.0, conv_bias=bias,
                       affine=norm != "instance")
        x = [y]
    if add_squeeze:
        y = E.squeeze_excite(ctx, y, f"{prefix}.conv.{3 * n_convs}", valid, spec.pad_value if valid is not None else 0.0)
    return y


def _norm_groups(norm: str, channels: int) -> int:
    return channels if norm == "instance" else 4


def mbconv(ctx: E.Ctx, srcs: Sequence[Tensor], prefix: str, norm: str, spec: BackboneSpec, valid: Optional[Tensor],
           need_input_grad: bool = True) -> Tensor:
    """MBConv (reference mbconv.py:25-97): 1x1 expansion (x4) -> norm -> ReLU -> depthwise 3x3 (reflect, with bias) -> norm ->
    ReLU -> SqueezeAndExcitation -> 1x1 projection -> norm (no ReLU), plus the input when in == out channels (ResidualAdd).
    `prefix` names the MBConv; its inner Sequential is prefix.0.0.block with the residual, prefix.0.0.0 without."""
    cin = sum(t.shape[1] for t in srcs)
    res = (prefix + ".0.0.block.0.weight") in ctx.p
    base = prefix + (".0.0.block" if res else ".0.0.0")
    cout = ctx.p[base + ".7.weight"].shape[0]
    assert res == (cin == cout) and (not res or len(srcs) == 1)
    kind = _norm_kind(norm)
    pv = spec.pad_value if valid is not None else 0.0
    aff = norm != "instance"
    o = E.conv2d(ctx, srcs, base + ".0.weight", base + ".0.bias", 1, 1, 0, _lib.PAD_ZEROS, valid, need_input_grad=need_input_grad)
    o = E.norm_act(ctx, o, base + ".1", kind, _norm_groups(norm, o.shape[1]), True, None, valid, pv, conv_bias=base + ".0.bias",
                   affine=aff)
    o = E.depthwise_conv2d(ctx, o, base + ".3.weight", 3, 1, 1, _lib.PAD_REFLECT, valid, bname=base + ".3.bias")
    o = E.norm_act(ctx, o, base + ".4", kind, _norm_groups(norm, o.shape[1]), True, None, valid, pv, conv_bias=base + ".3.bias",
                   affine=aff)
    o = E.squeeze_excite(ctx, o, base + ".6", valid, pv)
    o = E.conv2d(ctx, [o], base + ".7.weight", base + ".7.bias", 1, 1, 0, _lib.PAD_ZEROS, valid)
    return E.norm_act(ctx, o, base + ".8", kind, _norm_groups(norm, o.shape[1]), False, srcs[0] if res else None, valid, pv,
                      conv_bias=base + ".7.bias", affine=aff)


def mbconv_layer(ctx, srcs, prefix, n, norm, spec, valid, need_input_grad=True):
    """MBConvLayer (reference mbconv.py:100-128): n MBConv blocks, prefix.conv.{i}."""
    x = list(srcs)
    for i in range(n):
        x = [mbconv(ctx, x, f"{prefix}.conv.{i}", norm, spec, valid, need_input_grad=need_input_grad or i > 0)]
    return x[0]


def conv_block(ctx, x, prefix, n_convs, norm, spec, valid, need_input_grad=True, depthwise_separable=False, add_squeeze=False):
    """ConvBlock (reference conv.py:168-200); with use_mbconv: MBConvBlock (mbconv.py:131-152)."""
    if spec.use_mbconv:
        return mbconv_layer(ctx, [x], prefix + ".conv", n_convs, norm, spec, valid, need_input_grad=need_input_grad)
    return conv_layer(ctx, [x], prefix + ".conv", n_convs, norm, 3, 1, 1, spec, valid, need_input_grad=need_input_grad,
                      depthwise_separable=depthwise_separable, add_squeeze=add_squeeze)


def down_conv_block(ctx, x, prefix, norm, spec, valid, depthwise_separable=False, add_squeeze=False):
    """DownConvBlock (reference conv.py:238-296): down -> conv1 -> out + conv2(out) (-> sae, conv.py:294)."""
    o = conv_layer(ctx, [x], prefix + ".down", 1, norm, spec.str_conv_k, spec.str_conv_s, spec.str_conv_p, spec, valid,
                   depthwise_separable=depthwise_separable)
    if spec.use_mbconv:             # MBDownConvBlock (mbconv.py:155-198): down -> conv1 -> conv2, no outer residual, no sae
        o1 = mbconv_layer(ctx, [o], prefix + ".conv1", 1, norm, spec, valid)
        return mbconv_layer(ctx, [o1], prefix + ".conv2", 1, norm, spec, valid)
    o1 = conv_layer(ctx, [o], prefix + ".conv1", 1, norm, 3, 1, 1, spec, valid, depthwise_separable=depthwise_separable)
    o2 = conv_layer(ctx, [o1], prefix + ".conv2", 1, norm, 3, 1, 1, spec, valid, residual=o1,
                    depthwise_separable=depthwise_separable)
    if add_squeeze:
        o2 = E.squeeze_excite(ctx, o2, prefix + ".sae", valid, spec.pad_value if valid is not None else 0.0)
    return o2


def up_conv_block(ctx, x, skip, prefix, spec):
    """UpConvBlock (reference conv.py:362-413).  torch.cat([up, skip]) is never materialised: conv1 reads two
    source tensors."""
    sk = E.conv2d(ctx, [skip], prefix + ".skip_conv.0.weight", prefix + ".skip_conv.0.bias", 1, 1, 0, _lib.PAD_ZEROS, None)
    sk = E.norm_act(ctx, sk, prefix + ".skip_conv.1", _lib.NORM_BATCH, 1, True, None, None,
                    conv_bias=prefix + ".skip_conv.0.bias")
    assert spec.str_conv_k == 4 and spec.str_conv_s == 2 and spec.str_conv_p == 1, "only k=4,s=2,p=1 up-convs are built"
    up = E.conv_transpose2d(ctx, x, prefix + ".up.0.weight", prefix + ".up.0.bias")
    up = E.norm_act(ctx, up, prefix + ".up.1", _lib.NORM_BATCH, 1, True, None, None, conv_bias=prefix + ".up.0.bias")
    if spec.use_mbconv:             # MBUpConvBlock (mbconv.py:201-250): conv1 -> conv2 (norm 'batch'), no outer residual
        o1 = mbconv_layer(ctx, [up, sk], prefix + ".conv1", 1, "batch", spec, None)
        return mbconv_layer(ctx, [o1], prefix + ".conv2", 1, "batch", spec, None)
    o1 = conv_layer(ctx, [up, sk], prefix + ".conv1", 1, "batch", 3, 1, 1, spec, None)
    return conv_layer(ctx, [o1], prefix + ".conv2", 1, "batch", 3, 1, 1, spec, None, residual=o1)


def ltae(ctx, x5, dates, valid, prefix, spec: BackboneSpec, drop: DropoutState, with_tail: bool, need_attn: bool = True):
    """LTAE.forward / LTAE4WTAE.forward (reference tae.py:451-504, 589-635).  need_attn=False: the attention masks have no
    reader (TimeUNet_v1 hands them to the caller only with return_att, timeunet.py:204-205); they may come back as None."""
    emb, attn = E.ltae_attention(ctx, x5, dates, valid, prefix, spec.n_head, spec.d_k, spec.d_model, spec.pe_period,
                                 spec.attn_dropout, with_tail, drop.attn_seed, drop.attn_keep, drop.seed_dev,
                                 pe_mode=spec.pe_mode, need_attn=need_attn)
    if not with_tail:
        return None, attn
    o = E.conv2d(ctx, [emb], prefix + ".mlp.0.weight", prefix + ".mlp.0.bias", 1, 1, 0, _lib.PAD_ZEROS, None)
    o = E.norm_act(ctx, o, prefix + ".mlp.2", _lib.NORM_BATCH, 1, True, None, None, conv_bias=prefix + ".mlp.0.bias")
    o = E.dropout_nchw(ctx, o, spec.mlp_dropout, drop.mlp_seed, drop.mlp_keep, drop.seed_dev)
    o = E.pixel_group_norm(ctx, o, prefix + ".out_norm", spec.n_head)
    return o, attn


@dataclass
class BackboneOutput:
    """What the reference's forward() can return (utae.py:233-252), before it is packed into a tuple."""
    logits: Optional[Tensor]              # None with encoder=True
    att: Tensor
    boundary: Optional[Tensor] = None     # [B,2,H,W] logits of the boundary head
    maps: Optional[List[Tensor]] = None   # decoder feature maps, coarsest first
    last: Optional[Tensor] = None         # last decoder map (the `out` of encoder=True)


def _decoder_and_head(ctx, out, skips, spec, att):
    n_stages = len(spec.encoder_widths)
    maps = [out]
    for i in range(n_stages - 1):
        out = up_conv_block(ctx, out, skips[i], f"up_blocks.{i}", spec)
        maps.append(out)
    keep_maps = maps if (spec.return_maps or spec.encoder) else None
    if spec.encoder:
        return BackboneOutput(None, att, None, keep_maps, out)
    if spec.use_mbconv:             # MBConvBlock heads: their norm defaults to 'group' (mbconv.py:136-141)
        logits = mbconv_layer(ctx, [out], "out_conv.conv", len(spec.out_conv), "group", spec, None)
        boundary = mbconv_layer(ctx, [out], "boundary_conv.conv", 2, "group", spec, None) if spec.add_boundary_loss else None
        return BackboneOutput(logits, att, boundary, keep_maps, out)
    logits = conv_layer(ctx, [out], "out_conv.conv", len(spec.out_conv), "batch", 3, 1, 1, spec, None)
    boundary = None
    if spec.add_boundary_loss:
        boundary = conv_layer(ctx, [out], "boundary_conv.conv", 2, "batch", 3, 1, 1, spec, None)
    return BackboneOutput(logits, att, boundary, keep_maps, out)


def _fold(x5: Tensor) -> Tensor:
    B, T = x5.shape[:2]
    return x5.view(B * T, *x5.shape[2:])


def _unfold(x4: Tensor, B: int, T: int) -> Tensor:
    return x4.view(B, T, *x4.shape[1:])


def utae_forward(ctx, spec, x5, dates, drop):
    """UTAE.forward (reference utae.py:200-252), default flags."""
    B, T = x5.shape[:2]
    valid = E.frame_flags(x5, spec.pad_value)
    dws = spec.conv_type == "depthwise_separable"
    se = spec.add_squeeze_excit
    f = conv_block(ctx, _fold(x5), "in_conv", 2, spec.encoder_norm, spec, valid, need_input_grad=False, depthwise_separable=dws,
                   add_squeeze=se)
    fmaps = [f]
    n_stages = len(spec.encoder_widths)
    for i in range(n_stages - 1):
        f = down_conv_block(ctx, f, f"down_blocks.{i}", spec.encoder_norm, spec, valid, depthwise_separable=dws, add_squeeze=se)
        fmaps.append(f)
    if ctx.tape is not None and ctx.early_hook is not None:
        ctx.tape.record(ctx.early_hook)         # in the backward pass: decoder and temporal encoder are done, the encoder follows
    out, att = ltae(ctx, _unfold(fmaps[-1], B, T), dates, valid, "temporal_encoder", spec, drop, True)
    skips = [E.temporal_aggregate(ctx, _unfold(fmaps[-(i + 2)], B, T), att, valid, spec.n_head, spec.agg_mode)
             for i in range(n_stages - 1)]
    return _decoder_and_head(ctx, out, skips, spec, att)


def timeunet_forward(ctx, spec, x5, dates, drop):
    """TimeUNet_v1.forward (reference timeunet.py:169-210)."""
    B, T = x5.shape[:2]
    valid = E.frame_flags(x5, spec.pad_value)
    dws = spec.conv_type == "depthwise_separable"
    se = spec.add_squeeze_excit
    f0 = conv_block(ctx, _fold(x5), "in_conv", 2, spec.encoder_norm, spec, valid, need_input_grad=False, depthwise_separable=dws,
                    add_squeeze=se)
    if ctx.tape is not None and ctx.early_hook is not None:
        ctx.tape.record(ctx.early_hook)         # in the backward pass: decoder and temporal encoder are done, the encoder follows
    out, att = ltae(ctx, _unfold(f0, B, T), dates, valid, "temporal_encoder", spec, drop, True, need_attn=ctx.want_att)
    fmaps = [out]
    n_stages = len(spec.encoder_widths)
    for i in range(n_stages - 1):
        fmaps.append(down_conv_block(ctx, fmaps[-1], f"down_blocks.{i}", spec.encoder_norm, spec, None, depthwise_separable=dws,
                                     add_squeeze=se))
    skips = [fmaps[-(i + 2)] for i in range(n_stages - 1)]
    return _decoder_and_head(ctx, fmaps[-1], skips, spec, att)


def wtae_forward(ctx, spec, x5, dates, drop):
    """WTAE.forward (reference wtae.py:220-279)."""
    B, T = x5.shape[:2]
    valid = E.frame_flags(x5, spec.pad_value)
    dws = spec.conv_type == "depthwise_separable"
    se = spec.add_squeeze_excit
    f0 = conv_block(ctx, _fold(x5), "in_conv", 2, spec.encoder_norm, spec, valid, need_input_grad=False, depthwise_separable=dws,
                    add_squeeze=se)
    red = f0
    n_stages = len(spec.encoder_widths)
    for i in range(n_stages - 1):
        red = down_conv_block(ctx, red, f"spatial_reduction.{i}", spec.encoder_norm, spec, valid, depthwise_separable=True,
                              add_squeeze=se)
    if ctx.tape is not None and ctx.early_hook is not None:
        ctx.tape.record(ctx.early_hook)         # in the backward pass: decoder and temporal encoder are done, the encoder follows
    _, att = ltae(ctx, _unfold(red, B, T), dates, valid, "temporal_encoder", spec, drop, False)
    fmaps = [E.temporal_aggregate(ctx, _unfold(f0, B, T), att, valid, spec.n_head, spec.agg_mode)]
    for i in range(n_stages - 1):
        fmaps.append(down_conv_block(ctx, fmaps[-1], f"down_blocks.{i}", spec.encoder_norm, spec, None, depthwise_separable=dws,
                                     add_squeeze=se))
    skips = [fmaps[-(i + 2)] for i in range(n_stages - 1)]
    return _decoder_and_head(ctx, fmaps[-1], skips, spec, att)


FORWARDS = {"utae": utae_forward, "timeunet": timeunet_forward, "wtae": wtae_forward}
