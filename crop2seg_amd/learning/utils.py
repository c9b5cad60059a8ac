"""Harness pieces that sit directly on the hot path (reference src/learning/utils.py:50-136, 312-328;
src/learning/weight_init.py:4-46; train.py:454,463-468): model selection, weight initialisation and the
train step  zero_grad -> forward -> CrossEntropy -> backward -> (gradient all-reduce) -> Adam.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.init as init

from .. import engine as E
from ..backbones import functional as Fn
from ..backbones.modules import UTAE, WTAE, TimeUNet_v1

Tensor = torch.Tensor

# Two-bucket gradient exchange overlapped with the encoder's backward pass (TrainStep._forward_backward); C2S_DDP_OVERLAP=0
# keeps the single all-reduce after the backward pass.
OVERLAP_EXCHANGE = __import__("os").environ.get("C2S_DDP_OVERLAP", "1") != "0"


def get_model(config):
    """reference src/learning/utils.py:50-136: dispatch on config.model with the same kwarg mapping
    (encoder=False, return_maps=False fixed)."""
    common = dict(
        input_dim=config.input_dim, encoder_widths=config.encoder_widths, decoder_widths=config.decoder_widths,
        out_conv=config.out_conv, str_conv_k=config.str_conv_k, str_conv_s=config.str_conv_s,
        str_conv_p=config.str_conv_p, agg_mode=config.agg_mode, encoder_norm=config.encoder_norm, n_head=config.n_head,
        d_model=config.d_model, d_k=config.d_k, encoder=False, return_maps=False, pad_value=config.pad_value,
        padding_mode=config.padding_mode, conv_type=config.conv_type, use_mbconv=config.use_mbconv,
        add_squeeze_excit=config.add_squeeze, use_abs_rel_enc=config.use_abs_rel_enc, num_queries=config.num_queries,
        use_doy=config.use_doy, add_linear=config.add_linear)
    if config.model == "utae":
        return UTAE(add_boundary_loss=config.add_boundary_loss, **common)
    if config.model == "wtae":
        return WTAE(add_boundary_loss=config.add_boundary_loss, **common)
    if config.model == "timeunet":
        return TimeUNet_v1(**common)
    raise NotImplementedError(f"model {config.model!r}: crop2seg_amd builds utae / wtae / timeunet")


def default_config(model: str = "utae", **overrides):
    """argparse defaults of the reference's train.py:25-186 that reach get_model()."""
    from types import SimpleNamespace
    cfg = dict(model=model, encoder_widths=[64, 64, 64, 128], decoder_widths=[32, 32, 64, 128], out_conv=[32, 15],
               str_conv_k=4, str_conv_s=2, str_conv_p=1, agg_mode="att_group", encoder_norm="group", n_head=16,
               d_model=256, d_k=4, input_dim=10, num_queries=1, pad_value=0, padding_mode="reflect", conv_type="2d",
               use_mbconv=False, add_squeeze=False, use_doy=False, use_abs_rel_enc=False, add_linear=False,
               add_boundary_loss=False, num_classes=15, ignore_index=-1, lr=1e-3, label_smoothing=0.0)
    cfg.update(overrides)
    return SimpleNamespace(**cfg)


def weight_init(m):
    """reference src/learning/weight_init.py:4-46 (the module types that occur in the three backbones)."""
    if isinstance(m, nn.Conv1d):
        init.normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)
    elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        init.xavier_normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)
    elif isinstance(m, (nn.BatchNorm1d, nn.BatchNorm2d)):
        init.normal_(m.weight.data, mean=0, std=1)
        init.constant_(m.bias.data, 0)
    elif isinstance(m, nn.Linear):
        init.xavier_normal_(m.weight.data)
        if m.bias is not None:
            init.normal_(m.bias.data)


class TrainStep:
    """One optimiser step of the reference's training loop (src/learning/utils.py:314-328) with everything on
    the HIP engine and no autograd graph:

        zero_grad -> model(x, batch_positions=dates) -> CrossEntropyLoss(weight) -> backward -> Adam.step()

    Parameters, gradients and Adam moments live in three flat fp32 buffers (the module's parameters are
    re-pointed at views of the flat parameter buffer), so data-parallel training needs exactly one all-reduce of
    `flat_grad` per step (RCCL over xGMI; torch.distributed backend "nccl") and the optimiser is one kernel.
    """

    def __init__(self, model, num_classes: int = 15, ignore_index: int = -1, lr: float = 1e-3,
                 betas: Tuple[float, float] = (0.9, 0.999), eps: float = 1e-8, process_group=None,
                 distributed: bool = False, label_smoothing: float = 0.0, boundary_gamma: float = 2.0):
        self.model = model
        self.label_smoothing = float(label_smoothing)       # train.py:172,466-468
        self.boundary_gamma = float(boundary_gamma)         # FocalCELoss(gamma=2.0), src/learning/utils.py:259
        self.num_classes = num_classes
        if model.spec.encoder:
            raise ValueError("TrainStep needs a model with a classification head (encoder=False)")
        self.lr, self.betas, self.eps = lr, betas, eps
        named = list(model.named_parameters())
        self.names = [n for n, _ in named]
        dev = named[0][1].device
        sizes = [p.numel() for _, p in named]
        # 16-byte aligned slots
        offs, o = [], 0
        for s in sizes:
            offs.append(o)
            o += (s + 3) // 4 * 4
        self.total = o
        self.offsets = offs
        self.flat_param = torch.zeros(o, device=dev, dtype=torch.float32)
        self.flat_grad = torch.zeros(o, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(o, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(o, device=dev, dtype=torch.float32)
        self.params: Dict[str, Tensor] = {}
        self.grads: Dict[str, Tensor] = {}
        for (n, p), off, s in zip(named, offs, sizes):
            view = self.flat_param[off:off + s].view_as(p)
            view.copy_(p.data)
            p.data = view                               # module parameters now alias the flat buffer
            self.params[n] = view
            self.grads[n] = self.flat_grad[off:off + s].view_as(p)
        cw = torch.ones(num_classes, device=dev, dtype=torch.float32)
        cw[ignore_index] = 0                            # train.py:463-464
        self.class_w = cw
        self.step_count = 0
        self.ws = E.Workspace(dev)
        self.dp = None
        if distributed:
            from .ddp import FlatDataParallel
            self.dp = FlatDataParallel(process_group)
            # identical initial weights AND BatchNorm buffers on every rank
            self.dp.sync_parameters(self.flat_param, [b for _, b in model.named_buffers()])

    # ------------------------------------------------------------------------------------------------
    def _forward_backward(self, x: Tensor, dates: Tensor, y: Tensor, drop: Fn.DropoutState, overlap_exchange: bool = False) -> Tuple[Tensor, Tensor]:
        """zero_grad -> forward -> CE -> backward into the flat gradient buffer (stream-ordered, no host sync)."""
        model = self.model
        tape = E.Tape()
        ctx = E.Ctx(self.params, dict(model.named_buffers()), self.grads, self.ws, model.training, tape)
        ctx.want_att = False                             # the step returns (loss, logits): nobody reads the attention masks
        self._early, self._early_off = None, 0
        if (overlap_exchange and self.dp is not None and self.dp.active and OVERLAP_EXCHANGE and not E.REDUCE_BATCH
                and not torch.cuda.is_current_stream_capturing()):
            # Gradient exchange in two buckets: the tape runs this hook once the backward pass has left the decoder
            # and the temporal encoder -- everything behind the per-frame encoder in the flat buffer is final then -- and that
            # suffix is summed over the ranks on a communication stream while the encoder's backward pass (the bulk of the
            # step) still runs; the encoder's own gradients follow after the join.
            def early_exchange():
                tape.flush_side()                        # the weight-gradient launches queued so far
                i = len(self.names)
                while i > 0 and self.names[i - 1] in ctx._gwritten:
                    i -= 1
                if i == 0 or i == len(self.names):
                    return
                self._early_off = self.offsets[i]
                self._early = self.dp.reduce_async(self.flat_grad[self._early_off:],
                                                   after=(E._side_stream(),) if tape.side_used else ())
            ctx.early_hook = early_exchange
        out = Fn.FORWARDS[model.spec.model](ctx, model.spec, x, dates, drop)
        logits = out.logits
        loss, glogits = E.cross_entropy(logits, y, self.class_w, self.ws, want_grad=True, label_smoothing=self.label_smoothing)
        tape.grads[logits.data_ptr()] = glogits
        if out.boundary is not None:
            # src/learning/utils.py:283-285,318-324: y_b from the dilated one-hot labels, loss = CE + FocalCE(out_b, y_b)
            from .losses import boundary_target, focal_ce
            y_b = boundary_target(y)
            _, g_b = focal_ce(out.boundary, y_b, self.boundary_gamma, want_grad=True, ws=self.ws, loss_out=loss)
            tape.grads[out.boundary.data_ptr()] = g_b
            self.last_boundary = out.boundary
        tape.backward()
        if not torch.cuda.is_current_stream_capturing():
            self.ws.finalize_pack_plan()                 # from the second step on, all weight packs are one launch
        for n in self.names:                             # parameters no kernel wrote to (none in the default models)
            if n not in ctx._gwritten:
                self.grads[n].zero_()
        return loss, logits

    def bad_targets(self) -> int:
        """Labels outside [0, num_classes) (other than ignore_index) in the last step's batch: torch's CrossEntropyLoss raises on
        them, the loss kernel skips and counts them -- call this every display_step to surface a mislabelled dataset
        (host synchronisation).  Also the place where a failed one-pass normalisation wait surfaces (`check_health`)."""
        n = E.bad_target_count(self.ws)
        self.check_health()
        return n

    def check_health(self) -> None:
        """Host-synchronising health check (every display_step; `StepMeters.watch(step)` calls it from `get_miou_acc()` /
        `loss_mean()`): raises if a one-pass normalisation wait gave up (engine.Workspace.check_sync: the area is reset and
        the process continues on the two-pass kernels).  A captured graph has the one-pass launches baked in, so it is
        dropped and has to be captured again."""
        try:
            self.ws.check_sync()
        except RuntimeError:
            self.graph_fb = self.graph_opt = None
            raise

    def _fresh_dropout(self) -> Fn.DropoutState:
        drop = Fn.DropoutState()
        if self.model.training:
            self._seed_calls = getattr(self, "_seed_calls", 0) + 1
            rank = self.dp.rank if self.dp is not None else 0
            base = (torch.initial_seed() * 0x9E3779B1 + self._seed_calls * 2 + rank * 0x51ED27) & ((1 << 62) - 1)
            drop.attn_seed, drop.mlp_seed = base, base + 1
        return drop

    @torch.no_grad()
    def __call__(self, x: Tensor, dates: Tensor, y: Tensor, dropout_state: Optional[Fn.DropoutState] = None,
                 apply_update: bool = True) -> Tuple[Tensor, Tensor]:
        """Eager step.  Returns (loss[1] device tensor, logits).  No host synchronisation inside."""
        self.model._check_inputs(x, dates)
        drop = dropout_state if dropout_state is not None else self._fresh_dropout()
        loss, logits = self._forward_backward(x.contiguous(), dates.contiguous(), y, drop, overlap_exchange=True)
        scale = 1.0
        if self.dp is not None:
            if self._early is not None:                                   # the decoder's bucket has been under way since the
                scale = self.dp.reduce_gradients(self.flat_grad[:self._early_off])     # backward pass entered the encoder
                self._early.wait()
                self._early = None
            else:
                scale = self.dp.reduce_gradients(self.flat_grad)          # one 4.3 MB bucket per step
        if apply_update:
            self.step_count += 1
            E.adam_flat(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, self.step_count, self.lr,
                        self.betas[0], self.betas[1], self.eps, grad_scale=scale)
        return loss, logits

    # ------------------------------------------------------------------------------------------------
    # hipGraph path: the whole step (about 335 kernel launches, ~3 ms of host time when launched eagerly) is
    # captured once and replayed.  Everything the step needs to vary between replays lives on the device: the
    # dropout seed offset and the Adam step count are device counters advanced inside the graph.
    @torch.no_grad()
    def capture(self, x: Tensor, dates: Tensor, y: Tensor) -> None:
        """Capture the step for inputs of this shape.  `x`, `dates`, `y` are copied into static buffers; call
        `replay(x, dates, y)` (or `replay()` to reuse the buffers' content) afterwards."""
        self.model._check_inputs(x, dates)
        dev = x.device
        self.static_x, self.static_dates, self.static_y = x.clone().contiguous(), dates.clone().contiguous(), y.clone()
        self.step_dev = torch.full((1,), self.step_count, device=dev, dtype=torch.int32)
        self.seed_dev = torch.zeros(1, device=dev, dtype=torch.int64)
        drop = self._fresh_dropout()
        drop.seed_dev = self.seed_dev
        saved = {k: v.clone() for k, v in self.model.named_buffers()}     # the warm-up pass must not count as a step
        # the captured step runs on one stream: all slice sums of the weight gradients go into ONE launch at its end
        # (engine.REDUCE_BATCH; same order of additions); the warm-up pass below builds the job table the capture reuses
        batch0, E.REDUCE_BATCH = E.REDUCE_BATCH, True
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up on the side stream: lazy one-time initialisation
                self._forward_backward(self.static_x, self.static_dates, self.static_y, drop)   # (function attributes, workspaces)
            torch.cuda.current_stream().wait_stream(side)
            for k, v in self.model.named_buffers():
                v.copy_(saved[k])
            torch.cuda.synchronize()
            self.check_health()                             # never bake a poisoned sync area into a graph
            scale = 1.0 / self.dp.world if self.dp is not None else 1.0
            self.graph_fb = torch.cuda.CUDAGraph()
            # capture_error_mode thread_local: the process group's watchdog thread keeps polling the events of earlier collectives
            # (hipEventQuery), which the default global mode forbids while ANY thread captures
            with torch.cuda.graph(self.graph_fb, capture_error_mode="thread_local"):
                self.seed_dev.add_(1)
                self.static_loss, self.static_logits = self._forward_backward(self.static_x, self.static_dates, self.static_y, drop)
        finally:
            E.REDUCE_BATCH = batch0
        self.graph_opt = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph_opt, capture_error_mode="thread_local"):
            self.step_dev.add_(1)
            E.adam_flat(self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq, 0, self.lr, self.betas[0],
                        self.betas[1], self.eps, grad_scale=scale, step_dev=self.step_dev)

    @torch.no_grad()
    def replay(self, x: Optional[Tensor] = None, dates: Optional[Tensor] = None, y: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
        if getattr(self, "graph_fb", None) is None:
            raise RuntimeError("TrainStep.replay(): no captured step (capture() was not called, or check_health() dropped it)")
        if x is not None:
            self.static_x.copy_(x)
            self.static_dates.copy_(dates)
            self.static_y.copy_(y)
        self.graph_fb.replay()
        if self.dp is not None:
            self.dp.reduce_gradients(self.flat_grad)          # between the two graphs, on the same stream
        self.graph_opt.replay()
        self.step_count += 1
        return self.static_loss, self.static_logits
