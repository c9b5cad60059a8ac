"""CPU oracle for the Crop2Seg backbone hot path (U-TAE / TimeUNet_v1 / W-TAE).

TEST INFRASTRUCTURE ONLY.  This module is a plain-PyTorch (CPU, fp32 or fp64)
*functional restatement* of the reference's algorithm.  It is imported only by
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` -- never by the product package ``crop2seg_amd`` (which fails
loudly when its HIP library is missing).

Parity status: PINNED.  ``oracle/make_golden.py`` imports the reference from
``/root/reference`` (possible only in the build container) and writes seeded
input/output vectors to ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``
checks this restatement against them.  The reference itself ships no tests or
golden vectors for this path.

It is written as pure functions over a ``state_dict``-style mapping
(``name -> tensor``, same keys/shapes as the reference's ``state_dict()``,
SURVEY.md Appendix K), not as a copy of the reference's module tree.  Every
function cites the reference lines whose behaviour it restates (paths relative
to the reference root).

Deliberate differences from the reference (results identical):
  * no dummy all-zero forward to discover the output shape
    (src/backbones/temp_shared_block.py:24-26);
  * the per-pixel query stack / value split / positional table are expressed
    with broadcasting instead of P-fold Python stacks (src/backbones/tae.py:764,776);
  * dropout takes explicit keep-masks so train-mode parity is defined.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
State = Dict[str, Tensor]


# --------------------------------------------------------------------------
# configuration (defaults = train.py:32-47,153-166 of the reference)
# --------------------------------------------------------------------------
@dataclass
class BackboneConfig:
    model: str = "utae"                      # utae | timeunet | wtae
    input_dim: int = 10
    encoder_widths: List[int] = field(default_factory=lambda: [64, 64, 64, 128])
    decoder_widths: List[int] = field(default_factory=lambda: [32, 32, 64, 128])
    out_conv: List[int] = field(default_factory=lambda: [32, 15])
    str_conv_k: int = 4
    str_conv_s: int = 2
    str_conv_p: int = 1
    agg_mode: str = "att_group"
    encoder_norm: str = "group"
    n_head: int = 16
    d_model: int = 256
    d_k: int = 4
    pad_value: Optional[float] = 0.0
    padding_mode: str = "reflect"
    conv_type: str = "2d"                    # "depthwise_separable": in_conv and the down blocks (utae.py:144,158; conv.py:66-71)
    add_boundary_loss: bool = False          # second head boundary_conv = ConvBlock([dec0, 32, 2]) (utae.py:195-198)
    boundary_gamma: float = 2.0              # FocalCELoss(gamma=2.0) (src/learning/utils.py:259)
    pe_period: float = 1000.0                # PositionalEncoder T (positional_encoding.py:11)
    # positional encoder of the L-TAE (tae.py:404-430): "rel" sinusoid of the relative dates (default); "doy" =
    # AbsolutePositionalEncoder on the day of year (use_doy); "abs_rel" = sinusoid(dates[...,0]) + absolute(dates[...,1])
    # (use_abs_rel_enc, dates [B,T,2]); "linear" = Linear(256,256) on the tiled sinusoid (add_linear)
    pe_mode: str = "rel"
    # the reference's constructor flags for the same thing (fixtures store constructor kwargs): resolved into pe_mode
    use_doy: bool = False
    use_abs_rel_enc: bool = False
    add_linear: bool = False
    add_squeeze_excit: bool = False          # SqueezeAndExcitation after in_conv and the encoder down blocks (utae.py:145,159)
    use_mbconv: bool = False                 # MBConv blocks instead of the classical conv blocks (utae.py:118-122; mbconv.py)
    attn_dropout: float = 0.1                # tae.py:816
    mlp_dropout: float = 0.2                 # tae.py:361
    bn_momentum: float = 0.1
    eps: float = 1e-5

    def __post_init__(self) -> None:
        if self.use_abs_rel_enc:         # + the second encoder on dates[...,1]; the first follows the other flags (tae.py:407-423)
            self.pe_mode = "abs_rel_linear" if self.add_linear else ("abs_rel_doy" if self.use_doy else "abs_rel")
        elif self.add_linear:
            self.pe_mode = "linear"      # with or without use_doy (tae.py:405-409,414-417)
        elif self.use_doy:
            self.pe_mode = "doy"


class BNState:
    """Collects BatchNorm running-stat updates made during a train-mode pass."""

    def __init__(self) -> None:
        self.updates: Dict[str, Tensor] = {}


# --------------------------------------------------------------------------
# elementary layers
# --------------------------------------------------------------------------
# ReLU probe.  Gradients of a ReLU network are discontinuous where a pre-activation is exactly 0: two valid fp32
# evaluations (different summation order) can put an element with |pre-activation| ~ 1e-7 on different sides of
# the kink, which changes every upstream gradient by O(1/sqrt(#elements)) (measured: 3e-4..1.6e-3 on a 128x128
# block from ONE flipped element).  Tests therefore pick inputs whose fp64 evaluation keeps every pre-activation
# away from 0: set RELU_PROBE = [] and read the smallest |pre-activation| of each ReLU after a forward.
RELU_PROBE: Optional[list] = None


def relu(x: Tensor) -> Tensor:
    if RELU_PROBE is not None and x.numel():
        RELU_PROBE.append(float(x.detach().abs().min()))
    return F.relu(x)


def relu_margin(fn) -> float:
    """Smallest |pre-activation| over all ReLUs evaluated by fn()."""
    global RELU_PROBE
    RELU_PROBE = []
    try:
        with torch.no_grad():
            fn()
        return min(RELU_PROBE) if RELU_PROBE else float("inf")
    finally:
        RELU_PROBE = None

def _pad2d(x: Tensor, p: int, mode: str) -> Tensor:
    if p == 0:
        return x
    if mode == "zeros":
        return F.pad(x, (p, p, p, p))
    return F.pad(x, (p, p, p, p), mode=mode)


def conv2d(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int, pad: int,
           padding_mode: str, groups: int = 1) -> Tensor:
    """nn.Conv2d with padding_mode (src/backbones/conv.py:70-80)."""
    return F.conv2d(_pad2d(x, pad, padding_mode), w, b, stride=stride, groups=groups)


def batch_norm(x: Tensor, sd: State, prefix: str, training: bool, cfg: BackboneConfig,
               bn: Optional[BNState]) -> Tensor:
    """nn.BatchNorm{1,2}d: train = biased batch variance for normalisation,
    unbiased for the running update (momentum 0.1); eval = running stats
    (conv.py:52-53,380,388; tae.py:445)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    if not training:
        return F.batch_norm(x, rm, rv, w, b, False, cfg.bn_momentum, cfg.eps)
    # functional: the running buffers are updated on copies and handed back through `bn`
    rm2, rv2 = rm.detach().clone().to(x.dtype), rv.detach().clone().to(x.dtype)
    out = F.batch_norm(x, rm2, rv2, w, b, True, cfg.bn_momentum, cfg.eps)
    if bn is not None:
        bn.updates[prefix + ".running_mean"] = rm2
        bn.updates[prefix + ".running_var"] = rv2
        bn.updates[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
    return out


def squeeze_excite(x: Tensor, sd: State, prefix: str) -> Tensor:
    """SqueezeAndExcitation.forward (squeeze_and_excitation.py:16-30): x * sigmoid(W2 relu(W1 mean_hw(x)))."""
    y = x.mean(dim=(2, 3))
    y = relu(F.linear(y, sd[prefix + ".sae.1.weight"]))
    y = torch.sigmoid(F.linear(y, sd[prefix + ".sae.3.weight"]))
    return x * y[:, :, None, None]


def conv_layer(x: Tensor, sd: State, prefix: str, n_convs: int, norm: str, k: int, s: int, p: int,
               cfg: BackboneConfig, training: bool, bn: Optional[BNState],
               depthwise_separable: bool = False, add_squeeze: bool = False) -> Tensor:
    """ConvLayer: [conv -> norm -> ReLU] * n_convs; Sequential indices 3*i, 3*i+1
    (src/backbones/conv.py:29-96).  norm 'group' = GroupNorm(4 groups) (conv.py:42,56-60)."""
    for i in range(n_convs):
        cp = f"{prefix}.conv.{3 * i}"
        if depthwise_separable:          # conv.py:11-26, both convs bias-free
            dw = sd[cp + ".depthwise.weight"]
            x = conv2d(x, dw, None, s, p, cfg.padding_mode, groups=dw.shape[0])
            x = F.conv2d(x, sd[cp + ".pointwise.weight"])
        else:
            x = conv2d(x, sd[cp + ".weight"], sd[cp + ".bias"], s, p, cfg.padding_mode)
        np_ = f"{prefix}.conv.{3 * i + 1}"
        if norm == "group":
            x = F.group_norm(x, 4, sd[np_ + ".weight"], sd[np_ + ".bias"], cfg.eps)
        elif norm == "batch":
            x = batch_norm(x, sd, np_, training, cfg, bn)
        elif norm == "instance":
            x = F.instance_norm(x, eps=cfg.eps)
        x = relu(x)
    if add_squeeze:                      # conv.py:90-91: appended once, after the last conv-norm-ReLU
        x = squeeze_excite(x, sd, f"{prefix}.conv.{3 * n_convs}")
    return x


def _norm(x: Tensor, sd: State, prefix: str, norm: str, cfg: BackboneConfig, training: bool, bn: Optional[BNState]) -> Tensor:
    if norm == "group":
        return F.group_norm(x, 4, sd[prefix + ".weight"], sd[prefix + ".bias"], cfg.eps)
    if norm == "batch":
        return batch_norm(x, sd, prefix, training, cfg, bn)
    if norm == "instance":
        return F.instance_norm(x, eps=cfg.eps)
    raise ValueError(norm)


def mbconv(x: Tensor, sd: State, prefix: str, norm: str, cfg: BackboneConfig, training: bool, bn: Optional[BNState]) -> Tensor:
    """MBConv (mbconv.py:25-97): 1x1 expansion (x4) -> norm -> ReLU -> depthwise 3x3 (reflect, WITH bias) -> norm -> ReLU ->
    SqueezeAndExcitation -> 1x1 projection -> norm, wrapped in ResidualAdd when in == out channels.  `prefix` names the MBConv
    (a Sequential in a Sequential): the inner block is prefix.0.0.block.* with the residual, prefix.0.0.0.* without."""
    res = (prefix + ".0.0.block.0.weight") in sd
    base = prefix + (".0.0.block" if res else ".0.0.0")
    o = F.conv2d(x, sd[base + ".0.weight"], sd[base + ".0.bias"])
    o = relu(_norm(o, sd, base + ".1", norm, cfg, training, bn))
    dw = sd[base + ".3.weight"]
    o = conv2d(o, dw, sd[base + ".3.bias"], 1, 1, "reflect", groups=dw.shape[0])
    o = relu(_norm(o, sd, base + ".4", norm, cfg, training, bn))
    o = squeeze_excite(o, sd, base + ".6")
    o = F.conv2d(o, sd[base + ".7.weight"], sd[base + ".7.bias"])
    o = _norm(o, sd, base + ".8", norm, cfg, training, bn)
    return o + x if res else o


def mbconv_layer(x: Tensor, sd: State, prefix: str, n: int, norm: str, cfg: BackboneConfig, training: bool,
                 bn: Optional[BNState]) -> Tensor:
    """MBConvLayer (mbconv.py:100-128): n MBConv blocks, prefix.conv.{i}."""
    for i in range(n):
        x = mbconv(x, sd, f"{prefix}.conv.{i}", norm, cfg, training, bn)
    return x


def frame_pad_mask(x5: Tensor, pad_value: float) -> Tensor:
    """[B,T] mask of frames equal to pad_value everywhere (utae.py:201-203)."""
    return (x5 == pad_value).flatten(2).all(dim=-1)


def shared_over_time(fn, x: Tensor, pad_value: Optional[float]) -> Tensor:
    """TemporallySharedBlock.smart_forward (temp_shared_block.py:18-47):
    4-D input -> fn(x).  5-D: fold (B,T); frames that are entirely == pad_value are
    not processed and come out as pad_value; the remaining frames are processed as
    one compacted batch."""
    if x.dim() == 4:
        return fn(x)
    b, t = x.shape[:2]
    flat = x.reshape(b * t, *x.shape[2:])
    if pad_value is None:
        out = fn(flat)
        return out.view(b, t, *out.shape[1:])
    padded = (flat == pad_value).flatten(1).all(dim=-1)
    if not bool(padded.any()):
        out = fn(flat)
        return out.view(b, t, *out.shape[1:])
    valid_idx = (~padded).nonzero().squeeze(1)
    res = fn(flat.index_select(0, valid_idx))
    out = torch.full((b * t, *res.shape[1:]), float(pad_value), dtype=res.dtype)
    out = out.index_copy(0, valid_idx, res)
    return out.view(b, t, *out.shape[1:])


def conv_block(x: Tensor, sd: State, prefix: str, n_convs: int, norm: str, cfg: BackboneConfig,
               training: bool, bn: Optional[BNState], pad_value: Optional[float], depthwise_separable: bool = False,
               add_squeeze: bool = False) -> Tensor:
    """ConvBlock (conv.py:168-200) applied through smart_forward; with use_mbconv: MBConvBlock (mbconv.py:131-152)."""
    if cfg.use_mbconv:
        return shared_over_time(lambda z: mbconv_layer(z, sd, prefix + ".conv", n_convs, norm, cfg, training, bn), x, pad_value)
    return shared_over_time(
        lambda z: conv_layer(z, sd, prefix + ".conv", n_convs, norm, 3, 1, 1, cfg, training, bn, depthwise_separable,
                             add_squeeze),
        x, pad_value)


def down_conv_block(x: Tensor, sd: State, prefix: str, norm: str, cfg: BackboneConfig, training: bool,
                    bn: Optional[BNState], pad_value: Optional[float], depthwise_separable: bool = False,
                    add_squeeze: bool = False) -> Tensor:
    """DownConvBlock (conv.py:238-296): down(k,s,p) -> conv1 -> out + conv2(out) (-> sae, conv.py:294)."""
    def fn(z: Tensor) -> Tensor:
        o = conv_layer(z, sd, prefix + ".down", 1, norm, cfg.str_conv_k, cfg.str_conv_s, cfg.str_conv_p,
                       cfg, training, bn, depthwise_separable)
        if cfg.use_mbconv:               # MBDownConvBlock (mbconv.py:155-198): down -> conv1 -> conv2, no outer residual
            o = mbconv_layer(o, sd, prefix + ".conv1", 1, norm, cfg, training, bn)
            return mbconv_layer(o, sd, prefix + ".conv2", 1, norm, cfg, training, bn)
        o = conv_layer(o, sd, prefix + ".conv1", 1, norm, 3, 1, 1, cfg, training, bn, depthwise_separable)
        o = o + conv_layer(o, sd, prefix + ".conv2", 1, norm, 3, 1, 1, cfg, training, bn, depthwise_separable)
        return squeeze_excite(o, sd, prefix + ".sae") if add_squeeze else o
    return shared_over_time(fn, x, pad_value)


def up_conv_block(x: Tensor, skip: Tensor, sd: State, prefix: str, cfg: BackboneConfig, training: bool,
                  bn: Optional[BNState]) -> Tensor:
    """UpConvBlock (conv.py:362-413): skip 1x1+BN+ReLU; ConvTranspose2d(k,s,p)+BN+ReLU;
    concat [up, skip]; conv1; out + conv2(out).  Norm is always 'batch' (utae.py:171)."""
    sk = F.conv2d(skip, sd[prefix + ".skip_conv.0.weight"], sd[prefix + ".skip_conv.0.bias"])
    sk = relu(batch_norm(sk, sd, prefix + ".skip_conv.1", training, cfg, bn))
    up = F.conv_transpose2d(x, sd[prefix + ".up.0.weight"], sd[prefix + ".up.0.bias"],
                            stride=cfg.str_conv_s, padding=cfg.str_conv_p)
    up = relu(batch_norm(up, sd, prefix + ".up.1", training, cfg, bn))
    o = torch.cat([up, sk], dim=1)
    if cfg.use_mbconv:                   # MBUpConvBlock (mbconv.py:201-250): conv1 -> conv2 (MBConvLayers, norm 'batch'), no outer residual
        o = mbconv_layer(o, sd, prefix + ".conv1", 1, "batch", cfg, training, bn)
        return mbconv_layer(o, sd, prefix + ".conv2", 1, "batch", cfg, training, bn)
    o = conv_layer(o, sd, prefix + ".conv1", 1, "batch", 3, 1, 1, cfg, training, bn)
    return o + conv_layer(o, sd, prefix + ".conv2", 1, "batch", 3, 1, 1, cfg, training, bn)


# --------------------------------------------------------------------------
# L-TAE
# --------------------------------------------------------------------------
def positional_table(dates: Tensor, d: int, period: float, repeat: int, dtype=torch.float32) -> Tensor:
    """PositionalEncoder (positional_encoding.py:7-43): [B,T] int days -> [B,T,d*repeat].
    denom_i = period^(2*(i//2)/d); even i -> sin, odd i -> cos; tiled `repeat` times."""
    i = torch.arange(d, dtype=torch.float32)
    denom = torch.pow(torch.tensor(period, dtype=torch.float32), 2 * torch.div(i, 2, rounding_mode="floor") / d)
    tab = dates.to(torch.float32)[:, :, None] / denom[None, None, :]
    out = torch.empty_like(tab)
    out[..., 0::2] = torch.sin(tab[..., 0::2])
    out[..., 1::2] = torch.cos(tab[..., 1::2])
    return out.repeat(1, 1, repeat).to(dtype)


def absolute_table(doy: Tensor, w: Tensor, b: Tensor, repeat: int) -> Tensor:
    """AbsolutePositionalEncoder (positional_encoding.py:46-73): one_hot(day of year, 365) -> Linear(365, d) -> tiled."""
    oh = F.one_hot(doy.to(torch.int64), num_classes=365).to(w.dtype)       # [B,T,365]
    return F.linear(oh, w, b).repeat(1, 1, repeat)


def positional_encoding(dates: Tensor, sd: State, prefix: str, cfg: BackboneConfig, dtype=torch.float32) -> Tensor:
    """The positional term of LTAE.forward (tae.py:404-430,467-479) for the four encoder configurations, [B,T,d_model]."""
    H, dm = cfg.n_head, cfg.d_model
    mode = getattr(cfg, "pe_mode", "rel")
    if mode == "rel":
        return positional_table(dates, dm // H, cfg.pe_period, H, dtype)
    if mode == "doy":
        return absolute_table(dates, sd[prefix + ".positional_encoder.fc.weight"], sd[prefix + ".positional_encoder.fc.bias"], H)
    if mode == "abs_rel":
        return (positional_table(dates[..., 0], dm // H, cfg.pe_period, H, dtype)
                + absolute_table(dates[..., 1], sd[prefix + ".positional_encoder_abs.fc.weight"],
                                 sd[prefix + ".positional_encoder_abs.fc.bias"], H))
    if mode == "linear":
        tab = positional_table(dates, dm // H, cfg.pe_period, H, dtype)
        return F.linear(tab, sd[prefix + ".positional_encoder.fc.weight"], sd[prefix + ".positional_encoder.fc.bias"])
    if mode in ("abs_rel_doy", "abs_rel_linear"):      # tae.py:407-423: both encoders are learnable; :473: their sum
        w1, b1 = sd[prefix + ".positional_encoder.fc.weight"], sd[prefix + ".positional_encoder.fc.bias"]
        if mode == "abs_rel_doy":
            first = absolute_table(dates[..., 0], w1, b1, H)
        else:
            first = F.linear(positional_table(dates[..., 0], dm // H, cfg.pe_period, H, dtype), w1, b1)
        return first + absolute_table(dates[..., 1], sd[prefix + ".positional_encoder_abs.fc.weight"],
                                      sd[prefix + ".positional_encoder_abs.fc.bias"], H)
    raise ValueError(mode)


def ltae_attention(x: Tensor, dates: Tensor, pad_mask: Optional[Tensor], sd: State, prefix: str,
                   cfg: BackboneConfig, attn_keep: Optional[Tensor]) -> Tuple[Tensor, Tensor]:
    """Steps 1-6 of LTAE.forward (tae.py:451-481) + LightweightMultiHeadAttention (tae.py:738-807)
    + ScaledDotProductAttention (tae.py:810-847).

    x [B,T,C,h,w]; returns (embedding [P,d_model], attn [n_head,P,T]) with P=B*h*w in (b,h,w) order.
    ``attn`` is post-dropout (tae.py:836-839).  attn_keep: None (no dropout) or a [n_head,P,T]
    0/1 keep-mask applied as attn*keep/(1-p)."""
    B, T, C, h, w = x.shape
    H, dk, dm = cfg.n_head, cfg.d_k, cfg.d_model
    P = B * h * w
    seq = x.permute(0, 3, 4, 2, 1).reshape(P, C, T)                       # [P,C,T]  (tae.py:460-461)
    seq = F.group_norm(seq, H, sd[prefix + ".in_norm.weight"], sd[prefix + ".in_norm.bias"], cfg.eps)
    e = F.conv1d(seq, sd[prefix + ".inconv.weight"], sd[prefix + ".inconv.bias"])   # [P,dm,T] (tae.py:464)
    e = e.permute(0, 2, 1)                                                 # [P,T,dm]
    pe = positional_encoding(dates, sd, prefix, cfg, e.dtype)              # [B,T,dm] (tae.py:467-479)
    e = (e.view(B, h * w, T, dm) + pe[:, None]).view(P, T, dm)
    k = F.linear(e, sd[prefix + ".attention_head.fc1_k.weight"], sd[prefix + ".attention_head.fc1_k.bias"])
    k = k.view(P, T, H, dk)                                                # tae.py:768
    q = sd[prefix + ".attention_head.Q"][:, 0, :]                          # [H,dk]   (tae.py:752; n=1)
    scores = torch.einsum("hd,pthd->hpt", q, k) / math.sqrt(dk)            # tae.py:827-828
    if pad_mask is not None:
        pm = pad_mask[:, None, :].expand(B, h * w, T).reshape(P, T)
        scores = scores.masked_fill(pm[None], -1e6)                        # tae.py:831
    attn = torch.softmax(scores, dim=-1)
    if attn_keep is not None:
        attn = attn * attn_keep / (1.0 - cfg.attn_dropout)
    v = e.view(P, T, H, dm // H)                                           # values = unprojected e (tae.py:776)
    out = torch.einsum("hpt,pthc->phc", attn, v).reshape(P, dm)           # tae.py:839,796-798
    return out, attn


def ltae(x: Tensor, dates: Tensor, pad_mask: Optional[Tensor], sd: State, prefix: str, cfg: BackboneConfig,
         training: bool, bn: Optional[BNState], attn_keep: Optional[Tensor] = None,
         mlp_keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """LTAE.forward (tae.py:451-504), num_queries == 1.
    Returns (out [B,C',h,w], attn [n_head,B,T,h,w])."""
    B, T, C, h, w = x.shape
    emb, attn = ltae_attention(x, dates, pad_mask, sd, prefix, cfg, attn_keep)
    o = F.linear(emb, sd[prefix + ".mlp.0.weight"], sd[prefix + ".mlp.0.bias"])     # tae.py:443
    o = batch_norm(o, sd, prefix + ".mlp.2", training, cfg, bn)                     # BN1d over P (tae.py:445)
    o = relu(o)
    if mlp_keep is not None:
        o = o * mlp_keep / (1.0 - cfg.mlp_dropout)                                  # tae.py:448
    o = F.group_norm(o, cfg.n_head, sd[prefix + ".out_norm.weight"], sd[prefix + ".out_norm.bias"], cfg.eps)
    o = o.view(B, h, w, -1).permute(0, 3, 1, 2)                                     # tae.py:494
    attn = attn.view(cfg.n_head, B, h, w, T).permute(0, 1, 4, 2, 3)                 # tae.py:491-493
    return o, attn


def ltae_for_wtae(x: Tensor, dates: Tensor, pad_mask: Optional[Tensor], sd: State, prefix: str,
                  cfg: BackboneConfig, attn_keep: Optional[Tensor] = None) -> Tensor:
    """LTAE4WTAE.forward (tae.py:589-635): attention masks only."""
    B, T, C, h, w = x.shape
    _, attn = ltae_attention(x, dates, pad_mask, sd, prefix, cfg, attn_keep)
    return attn.view(cfg.n_head, B, h, w, T).permute(0, 1, 4, 2, 3)


def temporal_aggregate(x: Tensor, pad_mask: Optional[Tensor], attn: Tensor, mode: str = "att_group") -> Tensor:
    """TemporalAggregator.forward (temporal_aggregator.py:14-77).
    x [B,T,C,H,W], attn [n_head,B,T,h,w] -> [B,C,H,W]."""
    B, T, C, H, W = x.shape
    any_pad = pad_mask is not None and bool(pad_mask.any())
    if mode == "att_group":
        nh, _, _, h, w = attn.shape
        a = attn.reshape(nh * B, T, h, w)
        if H > w:
            a = F.interpolate(a, size=(H, W), mode="bilinear", align_corners=False)
        else:
            a = F.avg_pool2d(a, kernel_size=w // H)
        a = a.view(nh, B, T, H, W)
        if any_pad:
            a = a * (~pad_mask).to(a.dtype)[None, :, :, None, None]
        xs = x.view(B, T, nh, C // nh, H, W)
        return torch.einsum("nbthw,btnchw->bnchw", a, xs).reshape(B, C, H, W)
    if mode == "att_mean":
        a = attn.mean(dim=0)
        a = F.interpolate(a, size=(H, W), mode="bilinear", align_corners=False)
        if any_pad:
            a = a * (~pad_mask).to(a.dtype)[:, :, None, None]
        return (x * a[:, :, None]).sum(dim=1)
    if mode == "mean":
        if any_pad:
            keep = (~pad_mask).to(x.dtype)
            return (x * keep[:, :, None, None, None]).sum(dim=1) / keep.sum(dim=1)[:, None, None, None]
        return x.mean(dim=1)
    raise ValueError(mode)


# --------------------------------------------------------------------------
# whole models
# --------------------------------------------------------------------------
def _decoder_and_head(out: Tensor, skips: List[Tensor], sd: State, cfg: BackboneConfig, training: bool,
                      bn: Optional[BNState]) -> Tensor:
    n_stages = len(cfg.encoder_widths)
    for i in range(n_stages - 1):
        out = up_conv_block(out, skips[i], sd, f"up_blocks.{i}", cfg, training, bn)
    # out_conv = ConvBlock([dec0]+out_conv), BatchNorm + ReLU after BOTH convs (utae.py:191; conv.py:184)
    if cfg.use_mbconv:                   # out_conv = MBConvBlock(nkernels): its norm defaults to 'group' (mbconv.py:136-141)
        logits = mbconv_layer(out, sd, "out_conv.conv", len(cfg.out_conv), "group", cfg, training, bn)
        if cfg.add_boundary_loss:
            LAST_BOUNDARY.clear()
            LAST_BOUNDARY.append(mbconv_layer(out, sd, "boundary_conv.conv", 2, "group", cfg, training, bn))
        return logits
    logits = conv_layer(out, sd, "out_conv.conv", len(cfg.out_conv), "batch", 3, 1, 1, cfg, training, bn)
    if cfg.add_boundary_loss:                # utae.py:236-238: out_ = out_conv(out); out_b = boundary_conv(out)
        LAST_BOUNDARY.clear()
        LAST_BOUNDARY.append(conv_layer(out, sd, "boundary_conv.conv", 2, "batch", 3, 1, 1, cfg, training, bn))
    return logits


LAST_BOUNDARY: List[Tensor] = []             # boundary-head logits of the most recent forward (add_boundary_loss)


def utae_forward(sd: State, x: Tensor, dates: Tensor, cfg: BackboneConfig, training: bool = False,
                 bn: Optional[BNState] = None, attn_keep: Optional[Tensor] = None,
                 mlp_keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """UTAE.forward (utae.py:200-252), default flags.  Returns (logits, attn)."""
    pad_mask = frame_pad_mask(x, cfg.pad_value)
    dws = cfg.conv_type == "depthwise_separable"
    fmaps = [conv_block(x, sd, "in_conv", 2, cfg.encoder_norm, cfg, training, bn, cfg.pad_value, dws,
                    cfg.add_squeeze_excit)]
    n_stages = len(cfg.encoder_widths)
    for i in range(n_stages - 1):
        fmaps.append(down_conv_block(fmaps[-1], sd, f"down_blocks.{i}", cfg.encoder_norm, cfg, training, bn,
                                     cfg.pad_value, dws, cfg.add_squeeze_excit))
    out, att = ltae(fmaps[-1], dates, pad_mask, sd, "temporal_encoder", cfg, training, bn, attn_keep, mlp_keep)
    skips = [temporal_aggregate(fmaps[-(i + 2)], pad_mask, att, cfg.agg_mode) for i in range(n_stages - 1)]
    return _decoder_and_head(out, skips, sd, cfg, training, bn), att


def timeunet_forward(sd: State, x: Tensor, dates: Tensor, cfg: BackboneConfig, training: bool = False,
                     bn: Optional[BNState] = None, attn_keep: Optional[Tensor] = None,
                     mlp_keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """TimeUNet_v1.forward (timeunet.py:169-210): in_conv per frame -> L-TAE at full resolution
    -> plain U-Net on the single aggregated image (down blocks see 4-D input)."""
    pad_mask = frame_pad_mask(x, cfg.pad_value)
    dws = cfg.conv_type == "depthwise_separable"
    f0 = conv_block(x, sd, "in_conv", 2, cfg.encoder_norm, cfg, training, bn, cfg.pad_value, dws,
                    cfg.add_squeeze_excit)
    out, att = ltae(f0, dates, pad_mask, sd, "temporal_encoder", cfg, training, bn, attn_keep, mlp_keep)
    fmaps = [out]
    n_stages = len(cfg.encoder_widths)
    for i in range(n_stages - 1):
        fmaps.append(down_conv_block(fmaps[-1], sd, f"down_blocks.{i}", cfg.encoder_norm, cfg, training, bn,
                                     cfg.pad_value, dws, cfg.add_squeeze_excit))
    skips = [fmaps[-(i + 2)] for i in range(n_stages - 1)]
    return _decoder_and_head(fmaps[-1], skips, sd, cfg, training, bn), att


def wtae_forward(sd: State, x: Tensor, dates: Tensor, cfg: BackboneConfig, training: bool = False,
                 bn: Optional[BNState] = None, attn_keep: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """WTAE.forward (wtae.py:220-279): in_conv -> depthwise-separable spatial reduction ->
    attention masks (LTAE4WTAE) -> aggregate the full-resolution features -> plain U-Net."""
    pad_mask = frame_pad_mask(x, cfg.pad_value)
    dws = cfg.conv_type == "depthwise_separable"
    f0 = conv_block(x, sd, "in_conv", 2, cfg.encoder_norm, cfg, training, bn, cfg.pad_value, dws,
                    cfg.add_squeeze_excit)
    red = f0
    n_stages = len(cfg.encoder_widths)
    for i in range(n_stages - 1):
        red = down_conv_block(red, sd, f"spatial_reduction.{i}", cfg.encoder_norm, cfg, training, bn,
                              cfg.pad_value, depthwise_separable=True, add_squeeze=cfg.add_squeeze_excit)
    att = ltae_for_wtae(red, dates, pad_mask, sd, "temporal_encoder", cfg, attn_keep)
    fmaps = [temporal_aggregate(f0, pad_mask, att, cfg.agg_mode)]
    for i in range(n_stages - 1):
        fmaps.append(down_conv_block(fmaps[-1], sd, f"down_blocks.{i}", cfg.encoder_norm, cfg, training, bn,
                                     cfg.pad_value, dws, cfg.add_squeeze_excit))
    skips = [fmaps[-(i + 2)] for i in range(n_stages - 1)]
    return _decoder_and_head(fmaps[-1], skips, sd, cfg, training, bn), att


FORWARDS = {"utae": utae_forward, "timeunet": timeunet_forward, "wtae": wtae_forward}


def forward(sd: State, x: Tensor, dates: Tensor, cfg: BackboneConfig, **kw) -> Tuple[Tensor, Tensor]:
    return FORWARDS[cfg.model](sd, x, dates, cfg, **kw)


# --------------------------------------------------------------------------
# train step harness (src/learning/utils.py:312-328; train.py:454,463-468)
# --------------------------------------------------------------------------
def cross_entropy(logits: Tensor, target: Tensor, num_classes: int, ignore_index: int = -1,
                  label_smoothing: float = 0.0) -> Tensor:
    """nn.CrossEntropyLoss(weight=ones with weight[ignore_index]=0) (train.py:463-468):
    weighted mean, sum_i w[y_i]*nll_i / sum_i w[y_i]."""
    wgt = torch.ones(num_classes, dtype=logits.dtype)
    wgt[ignore_index] = 0
    return F.cross_entropy(logits, target, weight=wgt, label_smoothing=label_smoothing)


def adam_step(params: Dict[str, Tensor], grads: Dict[str, Tensor], m: Dict[str, Tensor], v: Dict[str, Tensor],
              step: int, lr: float = 1e-3, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> None:
    """torch.optim.Adam defaults (train.py:454), in place; `step` is 1-based."""
    for n, p in params.items():
        g = grads[n]
        m[n].mul_(b1).add_(g, alpha=1 - b1)
        v[n].mul_(b2).addcmul_(g, g, value=1 - b2)
        denom = (v[n].sqrt() / math.sqrt(1 - b2 ** step)).add_(eps)
        p.addcdiv_(m[n], denom, value=-lr / (1 - b1 ** step))


def parameter_names(sd: State) -> List[str]:
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var")
                                  or k.endswith("num_batches_tracked"))]


def loss_and_grads(sd: State, x: Tensor, dates: Tensor, y: Tensor, cfg: BackboneConfig, training: bool,
                   **kw) -> Tuple[Tensor, Tensor, Dict[str, Tensor], BNState]:
    """zero_grad -> forward -> CE -> backward (src/learning/utils.py:314-327).
    Returns (logits, loss, grads by name, BN updates)."""
    names = parameter_names(sd)
    work = {k: (v.detach().clone().requires_grad_(True) if k in names else v) for k, v in sd.items()}
    bn = BNState()
    label_smoothing = kw.pop("label_smoothing", 0.0)
    logits, _ = forward(work, x, dates, cfg, training=training, bn=bn, **kw)
    loss = cross_entropy(logits, y, cfg.out_conv[-1], label_smoothing=label_smoothing)
    if cfg.add_boundary_loss:                # src/learning/utils.py:283-285,318-324
        from . import tail_oracle as TO
        y_b = TO.boundary_target(y, cfg.out_conv[-1])
        loss = loss + TO.focal_ce(LAST_BOUNDARY[0], y_b, cfg.boundary_gamma)
    gs = torch.autograd.grad(loss, [work[n] for n in names], allow_unused=True)
    grads = {n: (g if g is not None else torch.zeros_like(work[n])) for n, g in zip(names, gs)}
    return logits.detach(), loss.detach(), grads, bn
