"""Drop-in model classes: UTAE, TimeUNet_v1, WTAE with the reference's constructor signature, forward()
contract and state_dict layout, executing on the HIP engine.

The submodules are *parameter holders* built from the same torch layer classes at the same module paths as
the reference (so `model.apply(weight_init)`, `load_state_dict` of a reference checkpoint, `.to()`,
`.train()/.eval()` and optimisers behave identically; reference src/learning/weight_init.py:13-46 dispatches on
isinstance).  Their own forward() is never used: the top-level model runs
crop2seg_amd.backbones.functional.* through libc2s_hip.so.  There is no PyTorch/CPU fallback: calling a model on
a non-HIP tensor or without the library raises.

Reference classes mirrored: src/backbones/utae.py:14, timeunet.py:10, wtae.py:15, conv.py:11,29,168,238,362,
tae.py:349,507,738, temporal_aggregator.py:6, positional_encoding.py:7.
"""
from __future__ import annotations

import copy
from typing import Dict, List, Optional

import numpy as np
import torch
import torch.nn as nn

from .. import engine as E
from . import functional as Fn

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# parameter-holder blocks (module paths == reference)
# ------------------------------------------------------------------------------------------------
class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError(f"{type(self).__name__} is a parameter holder; run the enclosing backbone instead")


class DepthwiseSeparableConv2D(_Holder):
    """reference conv.py:11-26 (both convolutions bias-free)."""

    def __init__(self, in_channels, out_channels, kernel_size=3, padding=1, padding_mode="zeros", stride=1, bias=False):
        super().__init__()
        self.depthwise = nn.Conv2d(in_channels, in_channels, kernel_size, padding=padding, padding_mode=padding_mode,
                                   stride=stride, groups=in_channels, bias=bias)
        self.pointwise = nn.Conv2d(in_channels, out_channels, kernel_size=1, bias=bias)


class SqueezeAndExcitation(_Holder):
    """reference squeeze_and_excitation.py:7-30: Sequential(Reduce, Linear(C, C/16, bias=False), ReLU, Linear(C/16, C,
    bias=False), Sigmoid, Rearrange) -- the Linear layers sit at indices 1 and 3."""

    def __init__(self, channel, reduction_ratio=16):
        super().__init__()
        if reduction_ratio != 16 or channel < 16:
            raise NotImplementedError("SqueezeAndExcitation: reduction_ratio 16 and >= 16 channels are built")
        self.sae = nn.Sequential(nn.Identity(), nn.Linear(channel, channel // reduction_ratio, bias=False), nn.ReLU(inplace=True),
                                 nn.Linear(channel // reduction_ratio, channel, bias=False), nn.Sigmoid(), nn.Identity())


class ConvLayer(_Holder):
    """reference conv.py:29-96: Sequential [conv, norm, ReLU] * (len(nkernels)-1) (+ SqueezeAndExcitation)."""

    def __init__(self, nkernels, norm="batch", k=3, s=1, p=1, n_groups=4, last_relu=True, padding_mode="reflect",
                 conv_type="2d", add_squeeze=False):
        super().__init__()
        if not last_relu or n_groups != 4:
            raise NotImplementedError("crop2seg_amd builds ConvLayer with last_relu and 4 normalisation groups only")
        layers: List[nn.Module] = []
        for i in range(len(nkernels) - 1):
            if conv_type == "depthwise_separable":
                layers.append(DepthwiseSeparableConv2D(nkernels[i], nkernels[i + 1], kernel_size=k, padding=p, stride=s,
                                                       padding_mode=padding_mode))
            else:
                layers.append(nn.Conv2d(nkernels[i], nkernels[i + 1], kernel_size=k, padding=p, stride=s,
                                        padding_mode=padding_mode))
            if norm == "batch":
                layers.append(nn.BatchNorm2d(nkernels[i + 1]))
            elif norm == "group":
                layers.append(nn.GroupNorm(num_channels=nkernels[i + 1], num_groups=n_groups))
            elif norm == "instance":
                layers.append(nn.InstanceNorm2d(nkernels[i + 1]))       # no parameters, no buffers (conv.py:54-55)
            else:
                raise NotImplementedError(f"norm={norm!r}: crop2seg_amd builds 'group', 'batch' and 'instance'")
            layers.append(nn.ReLU())
        if add_squeeze:
            layers.append(SqueezeAndExcitation(nkernels[-1]))            # conv.py:90-91
        self.conv = nn.Sequential(*layers)


class ConvBlock(_Holder):
    """reference conv.py:168-200."""

    def __init__(self, nkernels, pad_value=None, norm="batch", last_relu=True, padding_mode="reflect", conv_type="2d",
                 add_squeeze=False):
        super().__init__()
        self.pad_value = pad_value
        self.conv = ConvLayer(nkernels, norm=norm, last_relu=last_relu, padding_mode=padding_mode, conv_type=conv_type,
                              add_squeeze=add_squeeze)


class DownConvBlock(_Holder):
    """reference conv.py:238-296."""

    def __init__(self, d_in, d_out, k, s, p, pad_value=None, norm="batch", padding_mode="reflect", conv_type="2d",
                 add_squeeze=False):
        super().__init__()
        self.pad_value = pad_value
        self.down = ConvLayer([d_in, d_in], norm=norm, k=k, s=s, p=p, padding_mode=padding_mode, conv_type=conv_type)
        self.conv1 = ConvLayer([d_in, d_out], norm=norm, padding_mode=padding_mode, conv_type=conv_type)
        self.conv2 = ConvLayer([d_out, d_out], norm=norm, padding_mode=padding_mode, conv_type=conv_type)
        self.add_squeeze = add_squeeze
        if add_squeeze:
            self.sae = SqueezeAndExcitation(d_out)                       # conv.py:286-287


class UpConvBlock(_Holder):
    """reference conv.py:362-413."""

    def __init__(self, d_in, d_out, k, s, p, norm="batch", d_skip=None, padding_mode="reflect", conv_type="2d",
                 add_squeeze=False):
        super().__init__()
        if add_squeeze:
            raise NotImplementedError("squeeze-and-excitation is not built")
        d = d_out if d_skip is None else d_skip
        self.skip_conv = nn.Sequential(nn.Conv2d(d, d, kernel_size=1), nn.BatchNorm2d(d), nn.ReLU())
        self.up = nn.Sequential(nn.ConvTranspose2d(d_in, d_out, kernel_size=k, stride=s, padding=p),
                                nn.BatchNorm2d(d_out), nn.ReLU())
        self.conv1 = ConvLayer([d_out + d, d_out], norm=norm, padding_mode=padding_mode, conv_type=conv_type)
        self.conv2 = ConvLayer([d_out, d_out], norm=norm, padding_mode=padding_mode, conv_type=conv_type)


class ResidualAdd(_Holder):
    """reference mbconv.py:10-22."""

    def __init__(self, block: nn.Module):
        super().__init__()
        self.block = block


def _norm_layer(norm, n_groups=4):
    if norm == "batch":
        return nn.BatchNorm2d
    if norm == "instance":
        return nn.InstanceNorm2d
    if norm == "group":
        return lambda c: nn.GroupNorm(num_channels=c, num_groups=n_groups)     # raises for c % n_groups != 0, as the reference does
    raise NotImplementedError(f"norm={norm!r}: crop2seg_amd builds 'group', 'batch' and 'instance'")


class MBConv(nn.Sequential):
    """reference mbconv.py:25-97: Sequential(Sequential(residual(Sequential(1x1 expansion, norm, ReLU, depthwise 3x3 reflect,
    norm, ReLU, SqueezeAndExcitation, 1x1 projection, norm)))) with residual = ResidualAdd when in == out channels."""

    def __init__(self, in_channels: int, out_channels: int, expansion: int = 4, n_groups: int = 4, add_squeeze: bool = True,
                 norm: str = "group"):
        if expansion != 4 or n_groups != 4 or not add_squeeze:
            raise NotImplementedError("MBConv: expansion 4, 4 normalisation groups and squeeze-and-excitation are built")
        nl = _norm_layer(norm, n_groups)
        e = in_channels * expansion
        inner = nn.Sequential(nn.Conv2d(in_channels, e, kernel_size=1), nl(e), nn.ReLU(),
                              nn.Conv2d(e, e, groups=e, kernel_size=3, padding=1, padding_mode="reflect"), nl(e), nn.ReLU(),
                              SqueezeAndExcitation(e, reduction_ratio=16),
                              nn.Conv2d(e, out_channels, kernel_size=1), nl(out_channels))
        residual = ResidualAdd if in_channels == out_channels else nn.Sequential
        super().__init__(nn.Sequential(residual(inner)))

    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("MBConv is a parameter holder; run the enclosing backbone instead")


class MBConvLayer(_Holder):
    """reference mbconv.py:100-128."""

    def __init__(self, nkernels, norm):
        super().__init__()
        self.conv = nn.Sequential(*[MBConv(nkernels[i], nkernels[i + 1], expansion=4, norm=norm) for i in range(len(nkernels) - 1)])


class MBConvBlock(_Holder):
    """reference mbconv.py:131-152 (padding_mode / conv_type / add_squeeze are swallowed by **kwargs there, too)."""

    def __init__(self, nkernels, pad_value=None, norm="group", *args, **kwargs):
        super().__init__()
        self.pad_value = pad_value
        self.conv = MBConvLayer(nkernels=nkernels, norm=norm)


class MBDownConvBlock(_Holder):
    """reference mbconv.py:155-198: a classical ConvLayer for the strided down-sampling, then two MBConvLayers."""

    def __init__(self, d_in, d_out, k, s, p, pad_value=None, norm="batch", padding_mode="reflect", conv_type="2d", *args,
                 **kwargs):
        super().__init__()
        self.pad_value = pad_value
        self.down = ConvLayer([d_in, d_in], norm=norm, k=k, s=s, p=p, padding_mode=padding_mode, conv_type=conv_type)
        self.conv1 = MBConvLayer([d_in, d_out], norm=norm)
        self.conv2 = MBConvLayer([d_out, d_out], norm=norm)


class MBUpConvBlock(_Holder):
    """reference mbconv.py:201-250."""

    def __init__(self, d_in, d_out, k, s, p, d_skip=None, norm="batch", *args, **kwargs):
        super().__init__()
        d = d_out if d_skip is None else d_skip
        self.skip_conv = nn.Sequential(nn.Conv2d(d, d, kernel_size=1), nn.BatchNorm2d(d), nn.ReLU())
        self.up = nn.Sequential(nn.ConvTranspose2d(d_in, d_out, kernel_size=k, stride=s, padding=p),
                                nn.BatchNorm2d(d_out), nn.ReLU())
        self.conv1 = MBConvLayer([d_out + d, d_out], norm=norm)
        self.conv2 = MBConvLayer([d_out, d_out], norm=norm)


class LightweightMultiHeadAttention(_Holder):
    """reference tae.py:738-758: learnable master query Q [n_head, n, d_k] and the key projection fc1_k."""

    def __init__(self, n_head, d_k, d_in, n=1):
        super().__init__()
        self.n_head, self.d_k, self.d_in, self.n = n_head, d_k, d_in, n
        self.Q = nn.Parameter(torch.zeros((n_head, n, d_k))).requires_grad_(True)
        nn.init.normal_(self.Q, mean=0, std=np.sqrt(2.0 / d_k))
        self.fc1_k = nn.Linear(d_in, n_head * d_k)
        nn.init.normal_(self.fc1_k.weight, mean=0, std=np.sqrt(2.0 / d_k))


class PositionalEncoder(_Holder):
    """reference positional_encoding.py:7-43: no parameters unless add_linear, which puts a Linear(d*repeat, d*repeat) on
    the tiled sinusoid (the tables are computed by crop2seg_amd.engine.positional_table / c2s_ltae_pe_table)."""

    def __init__(self, d_model, T=1000, repeat=None, offset=0, add_linear=False):
        super().__init__()
        if offset != 0:
            raise NotImplementedError("PositionalEncoder(offset) is not built")
        self.d, self.T, self.repeat, self.add_linear = d_model, T, repeat, add_linear
        if add_linear:
            n = d_model * repeat if repeat is not None else d_model
            self.fc = nn.Linear(n, n)


class AbsolutePositionalEncoder(_Holder):
    """reference positional_encoding.py:46-73: one_hot(day of year, 365) -> Linear(365, d_model), tiled `repeat` times."""

    def __init__(self, d_model: int, repeat=None):
        super().__init__()
        self.d, self.repeat = d_model, repeat
        self.fc = nn.Linear(365, d_model)


def _pe_mode(use_abs_rel_enc, use_doy, add_linear) -> str:
    """Which positional term LTAE.forward adds for a flag combination (tae.py:404-430,467-479)."""
    if use_abs_rel_enc:                 # + AbsolutePositionalEncoder on batch_positions[...,1] (tae.py:419-422,473)
        if add_linear:
            return "abs_rel_linear"     # first encoder: PositionalEncoder(add_linear=True), with or without use_doy
        return "abs_rel_doy" if use_doy else "abs_rel"
    if add_linear:
        return "linear"                 # with or without use_doy: PositionalEncoder(add_linear=True) (tae.py:405-409,414-417)
    return "doy" if use_doy else "rel"


def _positional_encoders(mod, d_model, n_head, T, mode):
    if mode in ("doy", "abs_rel_doy"):
        mod.positional_encoder = AbsolutePositionalEncoder(d_model // n_head, repeat=n_head)
    else:
        mod.positional_encoder = PositionalEncoder(d_model // n_head, T=T, repeat=n_head,
                                                   add_linear=mode in ("linear", "abs_rel_linear"))
    if mode.startswith("abs_rel"):
        mod.positional_encoder_abs = AbsolutePositionalEncoder(d_model // n_head, repeat=n_head)


class LTAE(_Holder):
    """reference tae.py:349-449."""

    def __init__(self, in_channels=128, n_head=16, d_k=4, mlp=[256, 128], dropout=0.2, d_model=256, T=1000,
                 positional_encoding=True, use_abs_rel_enc=False, use_doy=False, num_queries=1, add_linear=False,
                 *args, **kwargs):
        super().__init__()
        if not positional_encoding or d_model is None:
            raise NotImplementedError("crop2seg_amd builds the L-TAE with an input projection and a positional encoder")
        self.in_channels, self.n_head, self.d_k, self.d_model, self.T = in_channels, n_head, d_k, d_model, T
        self.num_queries, self.use_abs_rel_enc, self.add_linear = num_queries, use_abs_rel_enc, add_linear
        self.pe_mode = _pe_mode(use_abs_rel_enc, use_doy, add_linear)
        self.dropout_p = dropout
        mlp = copy.deepcopy(mlp)
        assert mlp[0] == d_model and len(mlp) == 2
        self.inconv = nn.Conv1d(in_channels, d_model, 1)
        _positional_encoders(self, d_model, n_head, T, self.pe_mode)
        self.attention_head = LightweightMultiHeadAttention(n_head=n_head, d_k=d_k, d_in=d_model, n=num_queries)
        self.in_norm = nn.GroupNorm(num_groups=n_head, num_channels=in_channels)
        self.out_norm = nn.GroupNorm(num_groups=n_head, num_channels=mlp[-1])
        # indices 0 and 2 carry parameters (the reference has einops Rearrange at 1 and 3: tae.py:442-449)
        self.mlp = nn.Sequential(nn.Linear(mlp[0], mlp[1]), nn.Identity(), nn.BatchNorm1d(mlp[1]), nn.Identity(),
                                 nn.ReLU(), nn.Dropout(dropout))


class LTAE4WTAE(_Holder):
    """reference tae.py:507-587: attention masks only (no mlp / out_norm parameters)."""

    def __init__(self, in_channels=128, n_head=16, d_k=4, d_model=256, positional_encoding=True, use_abs_rel_enc=False,
                 num_queries=1, use_doy=False, add_linear=False, *args, **kwargs):
        super().__init__()
        if not positional_encoding or d_model is None:
            raise NotImplementedError("crop2seg_amd builds the L-TAE with an input projection and a positional encoder")
        self.in_channels, self.n_head, self.d_k, self.d_model = in_channels, n_head, d_k, d_model
        self.num_queries, self.use_abs_rel_enc, self.add_linear = num_queries, use_abs_rel_enc, add_linear
        self.pe_mode = _pe_mode(use_abs_rel_enc, use_doy, add_linear)
        self.inconv = nn.Conv1d(in_channels, d_model, 1)
        _positional_encoders(self, d_model, n_head, 1000, self.pe_mode)
        self.attention_head = LightweightMultiHeadAttention(n_head=n_head, d_k=d_k, d_in=d_model, n=num_queries)
        self.in_norm = nn.GroupNorm(num_groups=n_head, num_channels=in_channels)


class TemporalAggregator(_Holder):
    """reference temporal_aggregator.py:6-12 (modes att_group, att_mean, mean)."""

    def __init__(self, mode="mean"):
        super().__init__()
        if mode not in ("att_group", "att_mean", "mean"):
            raise NotImplementedError(f"agg_mode={mode!r}: crop2seg_amd builds 'att_group', 'att_mean' and 'mean'")
        self.mode = mode


# ------------------------------------------------------------------------------------------------
# autograd bridge: one Function for the whole backbone (explicit tape inside)
# ------------------------------------------------------------------------------------------------
def _pack_outputs(out: "Fn.BackboneOutput"):
    """Differentiable outputs of one forward in a fixed order: [head or last map, boundary head?, *maps?]."""
    tensors = [out.logits if out.logits is not None else out.last]
    if out.boundary is not None:
        tensors.append(out.boundary)
    if out.maps is not None:
        tensors.extend(out.maps)
    return tensors


class _BackboneFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, module, x, dates, drop, names, *params):
        p = dict(zip(names, params))
        grads = {n: torch.empty_like(t) for n, t in p.items()}
        tape = E.Tape()
        ectx = E.Ctx(p, dict(module.named_buffers()), grads, module._workspace(x.device), module.training, tape)
        with torch.no_grad():
            out = Fn.FORWARDS[module.spec.model](ectx, module.spec, x, dates, drop)
        diff = _pack_outputs(out)
        # a map that is also returned as `last` (encoder=True) must be a distinct autograd output
        ret = [diff[0]] + [t.view_as(t) if any(t is u for u in diff[:k + 1]) else t for k, t in enumerate(diff[1:])]
        ctx.tape, ctx.ectx, ctx.names, ctx.diff = tape, ectx, names, diff
        ctx.mark_non_differentiable(out.att)
        return (out.att, *ret)

    @staticmethod
    def backward(ctx, _g_att, *g_outs):
        tape, ectx = ctx.tape, ctx.ectx
        with torch.no_grad():
            for t, g in zip(ctx.diff, g_outs):
                if g is not None:
                    tape.add_grad(t, g.contiguous().clone())
            tape.backward()
            out = []
            for n in ctx.names:
                g = ectx.g[n]
                if n not in ectx._gwritten:
                    g.zero_()
                out.append(g)
        return (None, None, None, None, None, *out)


class _Backbone(nn.Module):
    spec: Fn.BackboneSpec

    def _workspace(self, device) -> E.Workspace:
        ws = getattr(self, "_ws", None)
        if ws is None or ws.device != device:
            ws = E.Workspace(device)
            object.__setattr__(self, "_ws", ws)
        return ws

    def check_health(self) -> None:
        """Host-synchronising check of the module's own workspace (the autograd / no-grad forward paths): raises if a
        one-pass normalisation wait gave up (engine.Workspace.check_sync)."""
        ws = getattr(self, "_ws", None)
        if ws is not None:
            ws.check_sync()

    def _check_inputs(self, input: Tensor, batch_positions: Optional[Tensor]):
        if batch_positions is None:
            raise ValueError("batch_positions (acquisition dates [B,T]) is required by the L-TAE positional encoding")
        if not input.is_cuda:
            raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback): move the model and inputs to 'cuda'")
        if input.dim() != 5 or input.dtype != torch.float32:
            raise ValueError("input must be float32 [B,T,C,H,W]")
        H, W = input.shape[-2:]
        if H % 8 or W % 8 or H < 16 or W < 16:
            raise ValueError("H and W must be multiples of 8 and >= 16")
        if input.shape[1] > 64:
            raise NotImplementedError("crop2seg_amd: the L-TAE kernels hold a whole series per workgroup: at most 64 time steps "
                                      f"(the dataset's longest series has 61, README.md:92); got T={input.shape[1]}")
        want = 3 if self.spec.pe_mode.startswith("abs_rel") else 2
        if batch_positions.dim() != want or (want == 3 and batch_positions.shape[-1] != 2):
            raise ValueError("batch_positions must be [B,T,2] (relative date, day of year) with use_abs_rel_enc, else [B,T]")
        if self.spec.num_queries != 1:
            # The reference's own forward fails for num_queries > 1 (probed with the imported reference): U-TAE / W-TAE unpack a
            # 6-D attention tensor into five names in the temporal aggregator (temporal_aggregator.py:25), TimeUNet feeds a
            # 5-D embedding to ConvTranspose2d.  Same exception types here.
            if self.spec.model == "timeunet":
                raise RuntimeError("Expected 3D (unbatched) or 4D (batched) input to conv_transpose2d: num_queries > 1 yields "
                                   "a 5-D embedding (the reference's forward raises the same way)")
            raise ValueError("too many values to unpack (expected 5): num_queries > 1 yields a 6-D attention tensor "
                             "(the reference's forward raises the same way, temporal_aggregator.py:25)")

    def forward(self, input: Tensor, batch_positions: Optional[Tensor] = None, return_att: bool = False,
                dropout_state: Optional[Fn.DropoutState] = None, *args, **kwargs):
        self._check_inputs(input, batch_positions)
        x = input.contiguous()
        dates = batch_positions.contiguous()
        drop = dropout_state
        if drop is None:
            drop = Fn.DropoutState()
            if self.training:
                seeds = torch.randint(0, 2 ** 62, (2,), device="cpu")   # host RNG (torch.manual_seed reproducible)
                drop.attn_seed, drop.mlp_seed = int(seeds[0]), int(seeds[1])
        named = [(n, p) for n, p in self.named_parameters()]
        names = [n for n, _ in named]
        params = [p for _, p in named]
        spec = self.spec
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            att, *diff = _BackboneFunction.apply(self, x, dates, drop, names, *params)
        else:
            with torch.no_grad():
                ectx = E.Ctx(dict(named), dict(self.named_buffers()), None, self._workspace(x.device), self.training, None)
                ectx.want_att = bool(return_att)
                out = Fn.FORWARDS[spec.model](ectx, spec, x, dates, drop)
                att, diff = out.att, _pack_outputs(out)
        # tuple conventions of the reference (utae.py:233-252, wtae.py:259-279, timeunet.py:201-210)
        head = diff[0]
        boundary = diff[1] if spec.add_boundary_loss and not spec.encoder else None
        n_lead = 1 + (1 if boundary is not None else 0)
        maps = list(diff[n_lead:]) if (spec.return_maps or spec.encoder) else None
        if spec.encoder:
            return head, maps
        lead = (head, boundary) if boundary is not None else (head,)
        if return_att:
            return (*lead, att)
        if spec.return_maps:
            return (*lead, maps)
        return lead if boundary is not None else head


def _common_init(self, model, input_dim, encoder_widths, decoder_widths, out_conv, str_conv_k, str_conv_s, str_conv_p,
                 agg_mode, encoder_norm, n_head, d_model, d_k, encoder, return_maps, pad_value, padding_mode, conv_type,
                 use_mbconv, add_squeeze_excit, use_abs_rel_enc, num_queries, use_doy, add_linear, add_boundary_loss):
    bad = []
    if conv_type not in ("2d", "depthwise_separable") or agg_mode not in ("att_group", "att_mean", "mean"):
        raise NotImplementedError(
            "crop2seg_amd builds the reference's default blocks (train.py:32-47,153-166) plus agg_mode in {att_group, att_mean, "
            "mean}, conv_type in {2d, depthwise_separable}, encoder_norm in {group, batch, instance}, add_boundary_loss, encoder, "
            "return_maps, add_squeeze_excit, use_mbconv and the positional encoders of use_doy / use_abs_rel_enc / add_linear; not built: "
            f"{bad or dict(conv_type=conv_type, agg_mode=agg_mode)}")
    # Limits compiled into the kernels (include/c2s_hip.h, INTEGRATION.md "Shape limits"): raise HERE, naming the limit, not as a
    # C-ABI error code from the first forward.  The reference builds any combination (train.py:35-43, tae.py:355-449).
    if (n_head, d_model, d_k) != (16, 256, 4):
        raise NotImplementedError(
            f"crop2seg_amd: the L-TAE kernels are built for n_head=16, d_model=256, d_k=4 (the reference's defaults, train.py:40-43); "
            f"got n_head={n_head}, d_model={d_model}, d_k={d_k}")
    if (str_conv_k, str_conv_s, str_conv_p) != (4, 2, 1):
        raise NotImplementedError(
            f"crop2seg_amd: the strided / transposed convolutions are built for str_conv_k=4, str_conv_s=2, str_conv_p=1 (the "
            f"reference's defaults, train.py:35-37); got k={str_conv_k}, s={str_conv_s}, p={str_conv_p}")
    c_ltae = encoder_widths[0] if model == "timeunet" else encoder_widths[-1]
    if c_ltae % 64 != 0 or c_ltae > 256:
        raise NotImplementedError(
            "crop2seg_amd: the L-TAE kernels take a multiple of 64 input channels, at most 256 (encoder_widths[-1]; "
            f"encoder_widths[0] for TimeUNet_v1); got encoder_widths={list(encoder_widths)}")
    if encoder:
        return_maps = True                      # utae.py:129-130
    if decoder_widths is None:
        decoder_widths = encoder_widths
    assert len(encoder_widths) == len(decoder_widths)
    assert encoder_widths[-1] == decoder_widths[-1]
    self.n_stages = len(encoder_widths)
    self.return_maps = return_maps
    self.encoder = encoder
    self.encoder_widths = encoder_widths
    self.decoder_widths = decoder_widths
    self.enc_dim = decoder_widths[0]
    self.stack_dim = sum(decoder_widths)
    self.pad_value = pad_value
    self.conv_type = conv_type
    self.spec = Fn.BackboneSpec(model=model, input_dim=input_dim, encoder_widths=list(encoder_widths),
                                decoder_widths=list(decoder_widths), out_conv=list(out_conv), str_conv_k=str_conv_k,
                                str_conv_s=str_conv_s, str_conv_p=str_conv_p, agg_mode=agg_mode,
                                encoder_norm=encoder_norm, n_head=n_head, d_model=d_model, d_k=d_k,
                                pad_value=float(pad_value), padding_mode=padding_mode, conv_type=conv_type,
                                add_boundary_loss=bool(add_boundary_loss), encoder=bool(encoder),
                                return_maps=bool(return_maps), pe_mode=_pe_mode(use_abs_rel_enc, use_doy, add_linear),
                                num_queries=int(num_queries), add_squeeze_excit=bool(add_squeeze_excit) and not use_mbconv,
                                use_mbconv=bool(use_mbconv))
    self.add_squeeze_excit, self.use_mbconv = bool(add_squeeze_excit), bool(use_mbconv)
    self.use_abs_rel_enc, self.use_doy, self.add_linear, self.num_queries = use_abs_rel_enc, use_doy, add_linear, num_queries
    self.add_boundary_loss = bool(add_boundary_loss)
    return decoder_widths


def _enc_blocks(encoder_widths, k, s, p, pad_value, norm, padding_mode, conv_type="2d", add_squeeze=False, use_mbconv=False):
    block = MBDownConvBlock if use_mbconv else DownConvBlock            # utae.py:118-122
    return nn.ModuleList(
        block(d_in=encoder_widths[i], d_out=encoder_widths[i + 1], k=k, s=s, p=p, pad_value=pad_value,
                      norm=norm, padding_mode=padding_mode, conv_type=conv_type, add_squeeze=add_squeeze)
        for i in range(len(encoder_widths) - 1))


def _dec_blocks(encoder_widths, decoder_widths, k, s, p, padding_mode, use_mbconv=False):
    n = len(encoder_widths)
    block = MBUpConvBlock if use_mbconv else UpConvBlock
    return nn.ModuleList(
        block(d_in=decoder_widths[i], d_out=decoder_widths[i - 1], d_skip=encoder_widths[i - 1], k=k, s=s, p=p,
                    norm="batch", padding_mode=padding_mode)
        for i in range(n - 1, 0, -1))


class UTAE(_Backbone):
    """reference src/backbones/utae.py:14-252."""

    def __init__(self, input_dim, encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 20],
                 str_conv_k=4, str_conv_s=2, str_conv_p=1, agg_mode="att_group", encoder_norm="group", n_head=16,
                 d_model=256, d_k=4, encoder=False, return_maps=False, pad_value=0, padding_mode="reflect",
                 conv_type="2d", use_mbconv=False, add_squeeze_excit=False, use_abs_rel_enc=False, num_queries=1,
                 use_doy=False, add_linear=False, add_boundary_loss=False, *args, **kwargs):
        super().__init__()
        decoder_widths = _common_init(self, "utae", input_dim, encoder_widths, decoder_widths, out_conv, str_conv_k,
                                      str_conv_s, str_conv_p, agg_mode, encoder_norm, n_head, d_model, d_k, encoder,
                                      return_maps, pad_value, padding_mode, conv_type, use_mbconv, add_squeeze_excit,
                                      use_abs_rel_enc, num_queries, use_doy, add_linear, add_boundary_loss)
        conv_block, head_block = (MBConvBlock, MBConvBlock) if use_mbconv else (ConvBlock, ConvBlock)      # utae.py:118-126
        self.in_conv = conv_block([input_dim, encoder_widths[0], encoder_widths[0]], pad_value=pad_value,
                                 norm=encoder_norm, padding_mode=padding_mode, conv_type=conv_type,
                                 add_squeeze=bool(add_squeeze_excit))
        self.down_blocks = _enc_blocks(encoder_widths, str_conv_k, str_conv_s, str_conv_p, pad_value, encoder_norm,
                                       padding_mode, conv_type=conv_type, add_squeeze=bool(add_squeeze_excit),
                                       use_mbconv=bool(use_mbconv))
        self.up_blocks = _dec_blocks(encoder_widths, decoder_widths, str_conv_k, str_conv_s, str_conv_p, padding_mode,
                                     use_mbconv=bool(use_mbconv))
        self.temporal_encoder = LTAE(in_channels=encoder_widths[-1], d_model=d_model, n_head=n_head, d_k=d_k, use_abs_rel_enc=use_abs_rel_enc, num_queries=num_queries, use_doy=use_doy, add_linear=add_linear)
        self.temporal_aggregator = TemporalAggregator(mode=agg_mode)
        self.out_conv = head_block([decoder_widths[0]] + list(out_conv), padding_mode=padding_mode)
        if add_boundary_loss:                                           # utae.py:195-198
            self.boundary_conv = head_block([decoder_widths[0]] + [32, 2], padding_mode=padding_mode)


class TimeUNet_v1(_Backbone):
    """reference src/backbones/timeunet.py:10-210."""

    def __init__(self, input_dim, encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 20],
                 str_conv_k=4, str_conv_s=2, str_conv_p=1, encoder_norm="group", n_head=16, d_model=256, d_k=4,
                 encoder=False, return_maps=False, pad_value=0, padding_mode="reflect", conv_type="2d",
                 add_squeeze_excit=False, use_abs_rel_enc=False, num_queries=1, use_doy=False, add_linear=False,
                 *args, **kwargs):
        super().__init__()
        kwargs.pop("agg_mode", None)
        decoder_widths = _common_init(self, "timeunet", input_dim, encoder_widths, decoder_widths, out_conv, str_conv_k,
                                      str_conv_s, str_conv_p, "att_group", encoder_norm, n_head, d_model, d_k, encoder,
                                      return_maps, pad_value, padding_mode, conv_type, False,     # use_mbconv / add_boundary_loss:
                                      add_squeeze_excit, use_abs_rel_enc, num_queries, use_doy, add_linear,
                                      False)                                                     # swallowed by **kwargs in the reference
        self.in_conv = ConvBlock([input_dim, encoder_widths[0], encoder_widths[0]], pad_value=pad_value,
                                 norm=encoder_norm, padding_mode=padding_mode, conv_type=conv_type,
                                 add_squeeze=bool(add_squeeze_excit))
        self.down_blocks = _enc_blocks(encoder_widths, str_conv_k, str_conv_s, str_conv_p, pad_value, encoder_norm,
                                       padding_mode, conv_type=conv_type, add_squeeze=bool(add_squeeze_excit))
        self.up_blocks = _dec_blocks(encoder_widths, decoder_widths, str_conv_k, str_conv_s, str_conv_p, padding_mode)
        self.temporal_encoder = LTAE(in_channels=encoder_widths[0], d_model=d_model, n_head=n_head, d_k=d_k,
                                     mlp=[d_model, encoder_widths[0]], use_abs_rel_enc=use_abs_rel_enc, num_queries=num_queries, use_doy=use_doy, add_linear=add_linear)
        self.out_conv = ConvBlock([decoder_widths[0]] + list(out_conv), padding_mode=padding_mode)


class WTAE(_Backbone):
    """reference src/backbones/wtae.py:15-279."""

    def __init__(self, input_dim, encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 20],
                 str_conv_k=4, str_conv_s=2, str_conv_p=1, agg_mode="att_group", encoder_norm="group", n_head=16,
                 d_model=256, d_k=4, encoder=False, return_maps=False, pad_value=0, padding_mode="reflect",
                 conv_type="2d", use_mbconv=False, add_squeeze_excit=False, use_abs_rel_enc=False, num_queries=1,
                 use_doy=False, add_linear=False, add_boundary_loss=False, *args, **kwargs):
        super().__init__()
        decoder_widths = _common_init(self, "wtae", input_dim, encoder_widths, decoder_widths, out_conv, str_conv_k,
                                      str_conv_s, str_conv_p, agg_mode, encoder_norm, n_head, d_model, d_k, encoder,
                                      return_maps, pad_value, padding_mode, conv_type, use_mbconv, add_squeeze_excit,
                                      use_abs_rel_enc, num_queries, use_doy, add_linear, add_boundary_loss)
        conv_block, head_block = (MBConvBlock, MBConvBlock) if use_mbconv else (ConvBlock, ConvBlock)      # utae.py:118-126
        self.in_conv = conv_block([input_dim, encoder_widths[0], encoder_widths[0]], pad_value=pad_value,
                                 norm=encoder_norm, padding_mode=padding_mode, conv_type=conv_type,
                                 add_squeeze=bool(add_squeeze_excit))
        self.spatial_reduction = _enc_blocks(encoder_widths, str_conv_k, str_conv_s, str_conv_p, pad_value, encoder_norm,
                                             padding_mode, conv_type="depthwise_separable", add_squeeze=bool(add_squeeze_excit),
                                             use_mbconv=bool(use_mbconv))
        self.down_blocks = _enc_blocks(encoder_widths, str_conv_k, str_conv_s, str_conv_p, pad_value, encoder_norm,
                                       padding_mode, conv_type=conv_type, add_squeeze=bool(add_squeeze_excit),
                                       use_mbconv=bool(use_mbconv))
        self.up_blocks = _dec_blocks(encoder_widths, decoder_widths, str_conv_k, str_conv_s, str_conv_p, padding_mode,
                                     use_mbconv=bool(use_mbconv))
        self.temporal_encoder = LTAE4WTAE(in_channels=encoder_widths[-1], d_model=d_model, n_head=n_head, d_k=d_k, use_abs_rel_enc=use_abs_rel_enc, num_queries=num_queries, use_doy=use_doy, add_linear=add_linear)
        self.temporal_aggregator = TemporalAggregator(mode=agg_mode)
        self.out_conv = head_block([decoder_widths[0]] + list(out_conv), padding_mode=padding_mode)
        if add_boundary_loss:                                           # wtae.py:215-218
            self.boundary_conv = head_block([decoder_widths[0]] + [32, 2], padding_mode=padding_mode)
