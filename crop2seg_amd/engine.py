"""Host-side execution engine: thin op wrappers over the C ABI (libc2s_hip.so) and an explicit backward tape.

PyTorch is used for device memory (torch.empty on the current HIP device), the current stream and
nn.Parameter storage only -- every activation-sized computation below is a hand-written HIP kernel.
The tape is a plain list of closures recorded during the forward pass and replayed in reverse: no
tracing, no autograd graph, stream-ordered launches only (hipGraph-capturable).
"""
from __future__ import annotations

import ctypes as C
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import AggDesc, ConvDesc, LtaeDesc, NormDesc, WgradDesc, check, lib

Tensor = torch.Tensor


def _ptr(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


import os as _os

_SIDE = None
SIDE_WGRAD = _os.environ.get("C2S_WGRAD_STREAM", "1") != "0"
SIDE_BATCH = int(_os.environ.get("C2S_WGRAD_BATCH", "8"))
SIDE_FLUSH_POSITIONS = int(_os.environ.get("C2S_WGRAD_FLUSH_POSITIONS", str(1 << 22)))
# While a hipGraph is being captured the fork / join events become cross-stream edges of the graph.  Measured (round 3,
# U-TAE B=4 T=32): the captured two-stream step replays at 17.1 ms against 12.65 ms for the single-stream capture and
# 12.25 ms for eager two-stream launches -- the graph executor serialises around the cross-stream edges -- so a capture stays
# on one stream unless C2S_GRAPH_SIDE=1.
GRAPH_SIDE = _os.environ.get("C2S_GRAPH_SIDE", "0") != "0"


def _side_ok() -> bool:
    """Parameter-gradient launches go to the side stream (eager always; under capture when GRAPH_SIDE)."""
    return SIDE_WGRAD and (GRAPH_SIDE or not torch.cuda.is_current_stream_capturing())


def _side_stream():
    global _SIDE
    if _SIDE is None:
        _SIDE = torch.cuda.Stream()
    return _SIDE


def _tap_array(offs: Sequence[int]):
    return (C.c_int * len(offs))(*offs)


class Workspace:
    """Named scratch buffers, grown on demand and reused across calls (never freed inside a step)."""

    def __init__(self, device):
        self.device = device
        if getattr(device, "type", None) == "cuda":
            idx = device.index if device.index is not None else torch.cuda.current_device()
            with torch.cuda.device(idx):
                _lib.init_device(idx)              # one-time per device, before any launch or hipGraph capture
        self.bufs: Dict[str, Tensor] = {}
        # weight-pack plan: the jobs recorded during one step become a device table that later steps run in one launch
        self.pack_record: Dict[Tuple, Tuple] = {}
        self.pack_plan: Optional[dict] = None
        self.reduce_jobs: List[Tuple] = []          # slice sums of the weight gradients recorded during this backward pass
        self.reduce_plan: Optional[dict] = None     # their one-launch table (built once, reused while the jobs stay the same)

    def finalize_pack_plan(self) -> None:
        """Turn the packs recorded during the step that just ran into a one-launch plan (host -> device table copy:
        call outside hipGraph capture; TrainStep does it at the end of every eager forward/backward)."""
        if self.pack_plan is not None or not self.pack_record:
            return
        L_ = lib()
        rec_bytes = L_.c2s_pack_job_bytes()
        jobs = list(self.pack_record.items())
        table = torch.zeros(len(jobs) * rec_bytes, dtype=torch.uint8).pin_memory() if torch.cuda.is_available() else \
            torch.zeros(len(jobs) * rec_bytes, dtype=torch.uint8)
        outs, block = {}, 0
        for i, (key, (src_ptr, cin, cout, coutP, ntaps, so, sc, wino, taps, nfloats)) in enumerate(jobs):
            out = torch.empty(nfloats, device=self.device, dtype=torch.float32)
            check(L_.c2s_pack_job_fill(table.data_ptr() + i * rec_bytes, src_ptr, out.data_ptr(), cin, cout, coutP, ntaps, so, sc,
                                       wino, _tap_array(taps), block), "pack_job_fill")
            block += L_.c2s_pack_job_blocks(cin, coutP, ntaps, wino)
            outs[key] = (out, src_ptr)
        self.pack_plan = {"table": table.to(self.device), "njobs": len(jobs), "blocks": block, "outs": outs}
        self.pack_record = {}

    def run_reduce_batch(self) -> None:
        """Sum the split-K slabs of every weight gradient recorded during this backward pass in one launch
        (c2s_wgrad_reduce_batch).  The job table is built on the host the first time (and again if a job changed) -- outside
        hipGraph capture; under capture a matching table must exist, otherwise the sums are launched one by one."""
        jobs, self.reduce_jobs = self.reduce_jobs, []
        if not jobs:
            return
        L_ = lib()
        key = tuple(j[:-1] for j in jobs)
        plan = self.reduce_plan
        if plan is None or plan["key"] != key:
            if torch.cuda.is_current_stream_capturing():
                import warnings
                warnings.warn("crop2seg_amd: the batched weight-gradient slice sum has no job table for this capture (the warm-up "
                              "pass ran other layers or buffers): falling back to one launch per layer inside the graph")
                for (dbytes, slabs_ptr, dst_ptr, so, sc, taps, acc, d) in jobs:
                    check(L_.c2s_wgrad_reduce(C.byref(d), slabs_ptr, dst_ptr, so, sc, _tap_array(taps), acc, _stream()), "wgrad_reduce")
                return
            rec = L_.c2s_wgrad_reduce_job_bytes()
            table = torch.zeros(len(jobs) * rec, dtype=torch.uint8).pin_memory()
            block = 0
            for i, (dbytes, slabs_ptr, dst_ptr, so, sc, taps, acc, d) in enumerate(jobs):
                check(L_.c2s_wgrad_reduce_job_fill(table.data_ptr() + i * rec, C.byref(d), slabs_ptr, dst_ptr, so, sc,
                                                   _tap_array(taps), acc, block), "wgrad_reduce_job_fill")
                block += L_.c2s_wgrad_reduce_job_blocks(C.byref(d))
            plan = self.reduce_plan = {"key": key, "table": table.to(self.device), "njobs": len(jobs), "blocks": block}
            # one slab buffer per weight lives as long as the plan (nslices * taps * CinP * CoutB floats each, ~70 MB for a U-TAE
            # step): buffers of an older plan (re-allocated gradient tensors) are dropped here
            live = {j[1] for j in jobs}
            for name in [k for k, b in self.bufs.items() if k.startswith("wgrad_slabs:") and b.data_ptr() not in live]:
                del self.bufs[name]
        check(L_.c2s_wgrad_reduce_batch(plan["table"].data_ptr(), plan["njobs"], plan["blocks"], _stream()), "wgrad_reduce_batch")

    def sync_area(self, nbytes: int) -> Tensor:
        """Zero-initialised area for kernels whose workgroups meet through memory (one-pass normalisation): all zero at rest
        (the kernels restore that state themselves), one area per workspace = per stream of launches.  When the area grows,
        the error word of the old one is carried over (device copy, stream-ordered)."""
        b = self.bufs.get("sync")
        if b is None or b.numel() < nbytes:
            nb = torch.zeros(max(int(nbytes), 4096), device=self.device, dtype=torch.uint8)
            if b is not None:
                nb[12:16].copy_(b[12:16])
            b = self.bufs["sync"] = nb
        return b

    def sync_error(self) -> int:
        """Error word of the sync area (host synchronisation): non-zero when a wait gave up."""
        b = self.bufs.get("sync")
        return 0 if b is None else int(b[:16].view(torch.int32)[3])

    def check_sync(self) -> None:
        """Raise if a one-pass normalisation wait gave up since the last check (host synchronisation -- call where the
        caller synchronises anyway).  The groups that gave up wrote NaN, so the step that hit it is lost; recovery: the area
        is re-zeroed and the process falls back to the two-pass normalisation kernels (`engine.ONEPASS_NORM = False`),
        which need no residency assumption, so the caller may catch the error and carry on with the next batch."""
        if self.sync_error() == 0:
            return
        global ONEPASS_NORM
        ONEPASS_NORM = False
        torch.cuda.synchronize(self.device)
        self.bufs["sync"].zero_()
        raise RuntimeError(
            "crop2seg_amd: a one-pass normalisation wait gave up (the workgroups of a group were not co-resident: "
            "shared or masked GPU, profiler serialisation?).  The affected outputs are NaN.  The sync area has been "
            "reset and this process now uses the two-pass normalisation kernels; repeat the step.")

    def get(self, name: str, nfloats: int) -> Tensor:
        b = self.bufs.get(name)
        if b is None or b.numel() < nfloats:
            b = torch.empty(max(int(nfloats), 1), device=self.device, dtype=torch.float32)
            self.bufs[name] = b
        return b


class Tape:
    """Reverse-mode tape.  Gradients are keyed by the data pointer of the forward activation, so a tensor and
    its reshaped views (4-D frames <-> 5-D [B,T,...]) share one gradient buffer."""

    def __init__(self):
        self.ops: List[Callable[[], None]] = []
        self.grads: Dict[int, Tensor] = {}
        self.keep: List[Tensor] = []      # keeps forward tensors alive (ids stay unique)
        self.side_keep: List[Tensor] = []  # operands of kernels running on the side stream (alive until the join)
        self.side_used = False
        self.side_pending: List[Callable[[], None]] = []   # weight-gradient launches waiting for the next fork
        self.finalizers: Dict[int, Callable[[], None]] = {}   # run once after the last side-stream launch (batched slice sums)

    def fork(self) -> "torch.cuda.Stream":
        """Side HIP stream ordered after everything issued so far on the current stream.  Weight gradients feed nothing
        until the optimizer, so they run there, next to the data-gradient chain, and fill the CUs that the small
        kernels of the chain (16x16 maps, decoder at N = B, L-TAE) leave idle; `backward()` joins the stream."""
        side = _side_stream()
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        self.side_used = True
        return side

    def record(self, fn: Callable[[], None]) -> None:
        self.ops.append(fn)

    def track(self, t: Tensor) -> Tensor:
        self.keep.append(t)
        return t

    def grad_of(self, t: Tensor) -> Optional[Tensor]:
        return self.grads.get(t.data_ptr())

    def pop_grad(self, t: Tensor) -> Optional[Tensor]:
        return self.grads.pop(t.data_ptr(), None)

    def add_grad(self, t: Tensor, g: Tensor, own: bool = True) -> None:
        """Accumulate g into the gradient of t.  If t has no gradient yet, g becomes it (own=True) or is copied."""
        cur = self.grads.get(t.data_ptr())
        if cur is None:
            self.grads[t.data_ptr()] = g if own else g.clone()
        else:
            check(lib().c2s_add_inplace(cur.data_ptr(), g.data_ptr(), g.numel(), _stream()), "add_inplace")

    def defer(self, fn: Callable[[], None], keep: Sequence[Tensor]) -> None:
        """Queue a launch for the side stream; a fork is issued every SIDE_BATCH launches (each fork/join is an edge of
        the captured hipGraph, and edges are not free)."""
        self.side_pending.append(fn)
        self.side_keep.extend(keep)
        if len(self.side_pending) >= SIDE_BATCH:
            self.flush_side()

    def flush_side(self) -> None:
        if not self.side_pending:
            return
        with torch.cuda.stream(self.fork()):
            for fn in self.side_pending:
                fn()
        self.side_pending.clear()

    def backward(self) -> None:
        for fn in reversed(self.ops):
            fn()
        self.flush_side()
        if self.finalizers:
            fins, self.finalizers = list(self.finalizers.values()), {}
            if self.side_used:                      # after the weight-gradient kernels, on their stream
                with torch.cuda.stream(_side_stream()):
                    for fn in fins:
                        fn()
            else:
                for fn in fins:
                    fn()
        if self.side_used:
            ev = torch.cuda.Event()
            ev.record(_side_stream())
            torch.cuda.current_stream().wait_event(ev)
            self.side_used = False
        self.side_keep.clear()
        self.ops.clear()
        self.keep.clear()


class Ctx:
    """Per-forward context: parameters by name, parameter-gradient views, scratch, tape, mode flags."""

    def __init__(self, params: Dict[str, Tensor], buffers: Dict[str, Tensor], grads: Optional[Dict[str, Tensor]],
                 ws: Workspace, training: bool, tape: Optional[Tape], eps: float = 1e-5, momentum: float = 0.1):
        self.p = params
        self.b = buffers
        self.g = grads            # name -> gradient tensor (same shape as the parameter), written by backward
        self.ws = ws
        self.training = training
        self.tape = tape
        self.eps = eps
        self.momentum = momentum
        self.device = ws.device
        self._packed: Dict[Tuple, Tensor] = {}
        self._plan_ran = False
        self._gwritten: set = set()
        self.early_hook = None    # TrainStep (data parallel): run by the tape when the backward pass enters the per-frame encoder
        self.want_att = True      # False: the caller never reads the attention masks a forward returns (TrainStep; inference
                                  # without return_att): TimeUNet's full-resolution L-TAE then does not store them
        cus = lib().c2s_device_cus()
        self.cus = cus if cus > 0 else 256

    # -- parameter gradient sinks ---------------------------------------------------------------
    def grad_sink(self, name: str) -> Tuple[Tensor, int]:
        """Returns (gradient tensor, accumulate flag) for parameter `name`."""
        acc = 1 if name in self._gwritten else 0
        self._gwritten.add(name)
        return self.g[name], acc

    def add_param_grad(self, name: str, g: Tensor) -> None:
        dst, acc = self.grad_sink(name)
        if acc:
            check(lib().c2s_add_inplace(dst.data_ptr(), g.data_ptr(), g.numel(), _stream()), "add_inplace")
        else:
            dst.copy_(g.view_as(dst))

    # -- weight packing --------------------------------------------------------------------------
    def _planned(self, key: Tuple, src_ptr: int) -> Optional[Tensor]:
        """Packed weights from the one-launch plan (run at the first pack request of the step), if the plan covers `key`
        for this source pointer; a stale plan (parameters re-allocated) is dropped."""
        plan = self.ws.pack_plan
        if plan is None:
            return None
        hit = plan["outs"].get(key)
        if hit is None or hit[1] != src_ptr:
            self.ws.pack_plan = None
            self.ws.pack_record = {}
            return None
        if not self._plan_ran:
            check(lib().c2s_pack_batch(plan["table"].data_ptr(), plan["njobs"], plan["blocks"], _stream()), "pack_batch")
            self._plan_ran = True
        return hit[0]

    def pack(self, key: Tuple, src: Tensor, src_off: int, cin: int, cout: int, ntaps: int, so: int, sc: int,
             taps: Sequence[int]) -> Tuple[Tensor, int]:
        """Pack (cached per forward) weights into [ntaps][cin][coutP]."""
        hit = self._packed.get(key)
        coutP = (cout + 31) // 32 * 32
        if hit is not None:
            return hit, coutP
        src_ptr = src.data_ptr() + 4 * src_off
        wpk = self._planned(key, src_ptr)
        if wpk is None:
            wpk = torch.empty(ntaps * cin * coutP, device=self.device, dtype=torch.float32)
            check(lib().c2s_pack_weights(src_ptr, wpk.data_ptr(), cin, cout, coutP, ntaps, so, sc, _tap_array(taps), _stream()),
                  "pack_weights")
            self.ws.pack_record[key] = (src_ptr, cin, cout, coutP, ntaps, so, sc, 0, tuple(taps), ntaps * cin * coutP)
        self._packed[key] = wpk
        return wpk, coutP


def _pack_bf16x3(ctx: "Ctx", key: Tuple, src: Tensor, src_off: int, cin: int, cout: int, so: int, sc: int,
                 taps: Sequence[int]) -> Tuple[Tensor, Tensor, int]:
    hit = ctx._packed.get(key)
    coutP = (cout + 31) // 32 * 32
    if hit is not None:
        return hit[0], hit[1], coutP
    n = lib().c2s_bf16x3_packed_elems(cin, coutP)
    whi = torch.empty(n, device=ctx.device, dtype=torch.bfloat16)
    wlo = torch.empty(n, device=ctx.device, dtype=torch.bfloat16)
    check(lib().c2s_pack_weights_bf16x3(src.data_ptr() + 4 * src_off, whi.data_ptr(), wlo.data_ptr(), cin, cout, coutP,
                                        len(taps), so, sc, _tap_array(taps), _stream()), "pack_weights_bf16x3")
    ctx._packed[key] = (whi, wlo)
    return whi, wlo, coutP


def _pack_winograd(ctx: "Ctx", key: Tuple, src: Tensor, src_off: int, cin: int, cout: int, so: int, sc: int,
                   taps: Sequence[int], wide: bool = False) -> Tuple[Tensor, int]:
    """U = G g Gt of every 3x3 filter, [16][cin][coutP] (cached per forward); `wide` = the layout of the 8-wave kernel
    (conv_winograd16.hip: [cout block][chunk][8 c][4 xi][64 o][4 nu])."""
    hit = ctx._packed.get(key)
    coutP = (cout + 63) // 64 * 64
    if hit is not None:
        return hit, coutP
    src_ptr = src.data_ptr() + 4 * src_off
    upk = ctx._planned(key, src_ptr)
    if upk is None:
        nfl = (lib().c2s_winograd16_packed_floats if wide else lib().c2s_winograd_packed_floats)(cin, coutP)
        upk = torch.empty(nfl, device=ctx.device, dtype=torch.float32)
        fn = lib().c2s_pack_weights_winograd16 if wide else lib().c2s_pack_weights_winograd
        check(fn(src_ptr, upk.data_ptr(), cin, cout, coutP, so, sc, _tap_array(taps), _stream()), "pack_weights_winograd")
        ctx.ws.pack_record[key] = (src_ptr, cin, cout, coutP, 9, so, sc, 2 if wide else 1, tuple(taps), nfl)
    ctx._packed[key] = upk
    return upk, coutP


def _use_winograd(K: int, S: int, pad: int, chans: Sequence[int], cout: int, H: int, W: int) -> bool:
    """Winograd F(2x2,3x3) pays off where the 16 transform-domain GEMMs are deep and wide enough."""
    return (WINOGRAD and CONV_MODE == "f32" and K == 3 and S == 1 and pad == 1 and sum(chans) >= 32 and cout >= 64
            and H % 2 == 0 and W % 4 == 0 and W >= 8 and (len(chans) == 1 or chans[0] % 8 == 0))


def _use_bf16x3(K: int, S: int, pad: int, chans: Sequence[int]) -> bool:
    return CONV_MODE == "bf16x3" and K == 3 and S == 1 and pad == 1 and all(c % 8 == 0 for c in chans)


# =================================================================================================
# frame flags
# =================================================================================================
def frame_flags(x5: Tensor, pad_value: float) -> Tensor:
    """valid[n] = any(x[n] != pad_value) (reference: utae.py:201-203, temp_shared_block.py:31)."""
    B, T = x5.shape[:2]
    valid = torch.empty(B * T, device=x5.device, dtype=torch.int32)
    check(lib().c2s_frame_flags(x5.data_ptr(), valid.data_ptr(), B * T, x5[0, 0].numel(), float(pad_value), _stream()),
          "frame_flags")
    return valid


# =================================================================================================
# convolutions
# =================================================================================================
# Convolution arithmetic: "f32" = exact fp32 MFMA everywhere (default); "bf16x3" = split-precision bf16 MFMA
# (three products per fp32 product, fp32 accumulate, ~1e-5 relative) for the 3x3 stride-1 forward / data-gradient
# launches whose channel counts are multiples of 8; everything else stays on the exact kernels.
import os as _os

CONV_MODE = _os.environ.get("C2S_CONV_MODE", "f32")
assert CONV_MODE in ("f32", "bf16x3"), CONV_MODE
# C2S_REDUCE_BATCH=1: the split-K slice sums of all weight gradients of a backward pass in one launch at its end (one slab
# buffer per layer) instead of one launch per layer right behind its weight-gradient kernel.  Measured neutral on the eager
# two-stream step (11.70 vs 11.71 ms), -0.2 ms under hipGraph capture: off by default, switched on by TrainStep.capture().
REDUCE_BATCH = _os.environ.get("C2S_REDUCE_BATCH", "0") != "0"
# fp32 Winograd F(2x2,3x3) for the wide 3x3 layers (forward + data gradient); C2S_WINOGRAD=0 keeps the direct kernel.
WINOGRAD = _os.environ.get("C2S_WINOGRAD", "1") != "0"
# the 8-wave Winograd kernel with the output transform in registers (conv_winograd16.hip) for planes >= 32 wide;
# C2S_WINO16=0 keeps the 4-wave kernel (conv_winograd.hip) everywhere
WINO16 = _os.environ.get("C2S_WINO16", "1") != "0"


# Winograd F(2x2,2x2) over the input parities for the 4x4 stride-2 forward convolutions (conv_s2wino.hip); C2S_S2WINO=0 keeps
# the direct kernel
S2WINO = _os.environ.get("C2S_S2WINO", "1") != "0"


def _wide_winograd(H: int, W: int, chans: Sequence[int]) -> bool:
    """c2s_conv3x3_winograd16_supported on top of _use_winograd: planes >= 32 wide, at least four chunks of 8 channels,
    whole chunks in every source (ragged channel counts stay on the 4-wave kernel, whose channel index is range-checked)."""
    return WINO16 and W >= 32 and H >= 8 and sum(chans) > 24 and all(c % 8 == 0 for c in chans)

# bench.py sets PROFILE = {"match": {field: value}, "events": []}: launches whose descriptor matches are bracketed
# with HIP events on the launch stream (the stream the kernel runs on) for the live roofline measurement.
PROFILE: Optional[dict] = None


def _igemm(desc: ConvDesc, src0: Tensor, src1: Optional[Tensor], wpk: Tensor, bias: Optional[Tensor], out: Tensor,
           valid: Optional[Tensor]) -> None:
    prof = PROFILE
    timed = prof is not None and all(getattr(desc, k) == v for k, v in prof["match"].items())
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().c2s_conv_igemm(C.byref(desc), src0.data_ptr(), _ptr(src1), wpk.data_ptr(), _ptr(bias), out.data_ptr(),
                               _ptr(valid), _stream()), "conv_igemm")
    if timed:
        e1.record()
        prof["events"].append((e0, e1))


def _winograd(desc: ConvDesc, src0: Tensor, src1: Optional[Tensor], upk: Tensor, bias: Optional[Tensor], out: Tensor,
              valid: Optional[Tensor], wide: bool = False) -> None:
    prof = PROFILE
    timed = prof is not None and all(getattr(desc, k) == v for k, v in prof["match"].items())
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    fn = lib().c2s_conv3x3_winograd16 if wide else lib().c2s_conv3x3_winograd
    check(fn(C.byref(desc), src0.data_ptr(), _ptr(src1), upk.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(valid), _stream()),
          "conv3x3_winograd")
    if timed:
        e1.record()
        prof["events"].append((e0, e1))


def _xpair_taps(py: int) -> List[int]:
    """4x4 weight taps of the transposed stride-2 convolution for output-row parity py, ordered [px][ty][tx]."""
    return [((3 - py) - 2 * ty) * 4 + ((3 - px) - 2 * tx) for px in range(2) for ty in range(2) for tx in range(2)]


def _xpair(d: ConvDesc, src: Tensor, wpk: Tensor, bias: Optional[Tensor], out: Tensor, valid: Optional[Tensor]) -> None:
    check(lib().c2s_conv_xpair(C.byref(d), src.data_ptr(), wpk.data_ptr(), _ptr(bias), out.data_ptr(), _ptr(valid),
                               _stream()), "conv_xpair")


def _wgrad_slices(ctx: Ctx, N: int, Hout: int, Wout: int, S: int, cin: int, cout: int) -> int:
    TP = 64 if S == 2 else 128
    l2 = 5
    while l2 > 2 and (1 << l2) > Wout:
        l2 -= 1
    PC = 1 << l2
    ntiles = N * ((Wout + PC - 1) // PC) * ((Hout + TP // PC - 1) // (TP // PC))
    blocks = ((cin + 31) // 32) * ((cout + 63) // 64)
    return max(1, min(ntiles, (2 * ctx.cus) // blocks))     # 2 resident workgroups per CU


def _wgrad(ctx: Ctx, srcs: Sequence[Tensor], gout: Tensor, Cout: int, Hout: int, Wout: int, K: int, S: int, pad: int,
           pad_mode: int, dst: Tensor, so: int, sc: int, taps: Sequence[int], accumulate: int,
           valid: Optional[Tensor]) -> None:
    """Weight gradient of a convolution into `dst` (split-K slabs + fixed-order slice sum), on the side stream."""
    if ctx.tape is not None and _side_ok():
        ctx.tape.defer(lambda: _wgrad_launch(ctx, srcs, gout, Cout, Hout, Wout, K, S, pad, pad_mode, dst, so, sc, taps,
                                             accumulate, valid), [gout, *srcs])
        if srcs[0].shape[0] * Hout * Wout >= SIDE_FLUSH_POSITIONS:
            ctx.tape.flush_side()       # very large layers (TimeUNet's 8M-position planes) come last in the backward pass: start
                                        # them now instead of after the rest of the batch.  (Round 3: with the one-pass norm
                                        # kernels the U-TAE layers of 2M positions run 0.15 ms/step better batched: 1<<22.)
    else:
        _wgrad_launch(ctx, srcs, gout, Cout, Hout, Wout, K, S, pad, pad_mode, dst, so, sc, taps, accumulate, valid)


def _wgrad_launch(ctx: Ctx, srcs: Sequence[Tensor], gout: Tensor, Cout: int, Hout: int, Wout: int, K: int, S: int, pad: int,
                  pad_mode: int, dst: Tensor, so: int, sc: int, taps: Sequence[int], accumulate: int,
                  valid: Optional[Tensor]) -> None:
    s0 = srcs[0]
    s1 = srcs[1] if len(srcs) > 1 else None
    N, C0, Hin, Win = s0.shape
    C1 = s1.shape[1] if s1 is not None else 0
    d = WgradDesc(N, C0, C1, Hin, Win, Cout, Hout, Wout, K, K, S, pad, pad, pad_mode,
                  _wgrad_slices(ctx, N, Hout, Wout, S, C0 + C1, Cout))
    nfl = lib().c2s_wgrad_workspace_floats(C.byref(d))
    batched = REDUCE_BATCH and ctx.tape is not None and not accumulate
    # batched slice sums: one slab buffer per weight (they all live until the end of the backward pass)
    slabs = ctx.ws.get(f"wgrad_slabs:{dst.data_ptr()}:{so}:{taps[0]}" if batched else "wgrad_slabs", nfl)
    check(lib().c2s_conv_wgrad(C.byref(d), s0.data_ptr(), _ptr(s1), gout.data_ptr(), slabs.data_ptr(), slabs.numel(),
                               _ptr(valid), _stream()), "conv_wgrad")
    if batched:
        ctx.ws.reduce_jobs.append((bytes(d), slabs.data_ptr(), dst.data_ptr(), so, sc, tuple(taps), accumulate, d))
        ctx.tape.finalizers[id(ctx.ws)] = ctx.ws.run_reduce_batch
    else:
        check(lib().c2s_wgrad_reduce(C.byref(d), slabs.data_ptr(), dst.data_ptr(), so, sc, _tap_array(taps), accumulate,
                                     _stream()), "wgrad_reduce")


def conv2d(ctx: Ctx, srcs: Sequence[Tensor], wname: str, bname: Optional[str], K: int, S: int, pad: int,
           pad_mode: int, valid: Optional[Tensor], need_input_grad: bool = True) -> Tensor:
    """nn.Conv2d (reference conv.py:70-80, 263-271, 378-382) over the channel concatenation of `srcs`."""
    W = ctx.p[wname]
    Cout, Cin = W.shape[0], W.shape[1]
    s0 = srcs[0]
    s1 = srcs[1] if len(srcs) > 1 else None
    N, C0, Hin, Win = s0.shape
    C1 = s1.shape[1] if s1 is not None else 0
    assert C0 + C1 == Cin, (wname, C0, C1, Cin)
    Ho = (Hin + 2 * pad - K) // S + 1
    Wo = (Win + 2 * pad - K) // S + 1
    KK = K * K
    out = torch.empty(N, Cout, Ho, Wo, device=s0.device, dtype=torch.float32)
    if _use_winograd(K, S, pad, [C0, C1] if C1 else [C0], Cout, Hin, Win):
        wide = _wide_winograd(Hin, Win, [C0, C1] if C1 else [C0]) and N <= 65536
        upk, CoutP = _pack_winograd(ctx, (wname, "fwd", "wino"), W, 0, Cin, Cout, Cin * KK, KK, list(range(KK)), wide)
        d = ConvDesc(N, C0, C1, Hin, Win, Cout, CoutP, Ho, Wo, Ho, Wo, K, K, S, pad, pad, pad_mode, 1, 1, 0, 0, 0)
        _winograd(d, s0, s1, upk, ctx.p[bname] if bname else None, out, valid, wide)
    elif _use_bf16x3(K, S, pad, [C0, C1]):
        whi, wlo, CoutP = _pack_bf16x3(ctx, (wname, "fwd", "bx"), W, 0, Cin, Cout, Cin * KK, KK, list(range(KK)))
        d = ConvDesc(N, C0, C1, Hin, Win, Cout, CoutP, Ho, Wo, Ho, Wo, K, K, S, pad, pad, pad_mode, 1, 1, 0, 0, 0)
        check(lib().c2s_conv3x3_bf16x3(C.byref(d), s0.data_ptr(), _ptr(s1), whi.data_ptr(), wlo.data_ptr(),
                                       _ptr(ctx.p[bname] if bname else None), out.data_ptr(), _ptr(valid), _stream()),
              "conv3x3_bf16x3")
    elif (S2WINO and CONV_MODE == "f32" and K == 4 and S == 2 and C1 == 0 and lib().c2s_conv4x4s2_winograd_supported(C.byref(
            ConvDesc(N, C0, 0, Hin, Win, Cout, (Cout + 63) // 64 * 64, Ho, Wo, Ho, Wo, K, K, S, pad, pad, pad_mode, 1, 1, 0, 0, 0)))):
        CoutP = (Cout + 63) // 64 * 64
        key = (wname, "fwd", "s2w")
        upk = ctx._packed.get(key)
        if upk is None:
            upk = ctx._planned(key, W.data_ptr())
            if upk is None:
                nfl = lib().c2s_s2wino_packed_floats(Cin, CoutP)
                upk = torch.empty(nfl, device=ctx.device, dtype=torch.float32)
                check(lib().c2s_pack_weights_s2wino(W.data_ptr(), upk.data_ptr(), Cin, Cout, CoutP, Cin * KK, KK,
                                                    _tap_array(list(range(KK))), _stream()), "pack_weights_s2wino")
                ctx.ws.pack_record[key] = (W.data_ptr(), Cin, Cout, CoutP, 16, Cin * KK, KK, 3, tuple(range(KK)), nfl)
            ctx._packed[key] = upk
        d = ConvDesc(N, C0, 0, Hin, Win, Cout, CoutP, Ho, Wo, Ho, Wo, K, K, S, pad, pad, pad_mode, 1, 1, 0, 0, 0)
        check(lib().c2s_conv4x4s2_winograd(C.byref(d), s0.data_ptr(), upk.data_ptr(), _ptr(ctx.p[bname] if bname else None),
                                           out.data_ptr(), _ptr(valid), _stream()), "conv4x4s2_winograd")
    else:
        wpk, CoutP = ctx.pack((wname, "fwd"), W, 0, Cin, Cout, KK, Cin * KK, KK, list(range(KK)))
        d = ConvDesc(N, C0, C1, Hin, Win, Cout, CoutP, Ho, Wo, Ho, Wo, K, K, S, pad, pad, pad_mode, 1, 1, 0, 0, 0)
        if lib().c2s_conv3x3_smallcin_supported(C.byref(d)):      # the first layer
            check(lib().c2s_conv3x3_smallcin(C.byref(d), s0.data_ptr(), wpk.data_ptr(), _ptr(ctx.p[bname] if bname else None),
                                             out.data_ptr(), _ptr(valid), _stream()), "conv3x3_smallcin")
        else:
            _igemm(d, s0, s1, wpk, ctx.p[bname] if bname else None, out, valid)
    if ctx.tape is None:
        return out
    tape = ctx.tape
    tape.track(out)

    def bwd():
        g = tape.pop_grad(out)
        if g is None:
            return
        gw, acc = ctx.grad_sink(wname)
        _wgrad(ctx, srcs, g, Cout, Ho, Wo, K, S, pad, pad_mode, gw, Cin * KK, KK, list(range(KK)), acc, valid)
        if not need_input_grad:
            return
        c_lo = 0
        for si, src in enumerate(srcs):
            Cs = src.shape[1]
            existing = tape.grad_of(src)
            gin = existing if existing is not None else torch.empty_like(src)
            accf = 1 if existing is not None else 0
            radj = 1 if (pad_mode == _lib.PAD_REFLECT and pad > 0) else 0   # reflection adjoint folded into the kernel
            if S == 1:
                taps = [(K - 1 - ky) * K + (K - 1 - kx) for ky in range(K) for kx in range(K)]
                if _use_winograd(K, S, pad, [Cout], Cs, Hin, Win):
                    wide = _wide_winograd(Hin, Win, [Cout]) and N <= 65536
                    upk, CP = _pack_winograd(ctx, (wname, "dgrad", "wino", si), W, c_lo * KK, Cout, Cs, KK, Cin * KK, taps, wide)
                    dd = ConvDesc(N, Cout, 0, Ho, Wo, Cs, CP, Hin, Win, Hin, Win, K, K, 1, 1, 1, _lib.PAD_ZEROS, 1, 1, 0, 0,
                                  accf, radj)
                    _winograd(dd, g, None, upk, None, gin, valid, wide)
                elif _use_bf16x3(K, S, pad, [Cout]):
                    whi, wlo, CP = _pack_bf16x3(ctx, (wname, "dgrad", "bx", si), W, c_lo * KK, Cout, Cs, KK, Cin * KK, taps)
                    dd = ConvDesc(N, Cout, 0, Ho, Wo, Cs, CP, Hin, Win, Hin, Win, K, K, 1, 1, 1, _lib.PAD_ZEROS, 1, 1, 0, 0,
                                  accf, radj)
                    check(lib().c2s_conv3x3_bf16x3(C.byref(dd), g.data_ptr(), None, whi.data_ptr(), wlo.data_ptr(), None,
                                                   gin.data_ptr(), _ptr(valid), _stream()), "conv3x3_bf16x3")
                else:
                    wd, CP = ctx.pack((wname, "dgrad", si), W, c_lo * KK, Cout, Cs, KK, KK, Cin * KK, taps)
                    dd = ConvDesc(N, Cout, 0, Ho, Wo, Cs, CP, Hin, Win, Hin, Win, K, K, 1, K - 1 - pad, K - 1 - pad,
                                  _lib.PAD_ZEROS, 1, 1, 0, 0, accf, radj)
                    _igemm(dd, g, None, wd, None, gin, valid)
            else:
                assert K == 4 and S == 2 and pad == 1
                assert Hin == 2 * Ho and Win == 2 * Wo
                CsP = (Cs + 63) // 64 * 64
                dw = ConvDesc(N, Cout, 0, Ho, Wo, Cs, CsP, Hin, Win, Hin, Win, 4, 4, 2, 1, 1, _lib.PAD_ZEROS, 1, 1, 0, 0, accf, radj)
                if S2WINO and CONV_MODE == "f32" and lib().c2s_conv4x4s2_dgrad_winograd_supported(C.byref(dw)):
                    key = (wname, "dgrad", "s2d", si)
                    upk = ctx._packed.get(key)
                    if upk is None:
                        src_ptr = W.data_ptr() + 4 * c_lo * KK
                        upk = ctx._planned(key, src_ptr)
                        if upk is None:
                            nfl = lib().c2s_s2dgrad_packed_floats(Cout, CsP)
                            upk = torch.empty(nfl, device=ctx.device, dtype=torch.float32)
                            check(lib().c2s_pack_weights_s2dgrad(src_ptr, upk.data_ptr(), Cout, Cs, CsP, KK, Cin * KK,
                                                                 _tap_array(list(range(KK))), _stream()), "pack_weights_s2dgrad")
                            # (pack-plan record: cin = gy channels, cout = input channels, strides as passed to the pack)
                            ctx.ws.pack_record[key] = (src_ptr, Cout, Cs, CsP, 16, KK, Cin * KK, 4, tuple(range(KK)), nfl)
                        ctx._packed[key] = upk
                    check(lib().c2s_conv4x4s2_dgrad_winograd(C.byref(dw), g.data_ptr(), upk.data_ptr(), gin.data_ptr(), _ptr(valid),
                                                             _stream()), "conv4x4s2_dgrad_winograd")
                    if existing is None:
                        tape.grads[src.data_ptr()] = gin
                    c_lo += Cs
                    continue
                for py in range(2):         # one launch per output-row parity, both column parities fused
                    wd, CP = ctx.pack((wname, "dgrad", si, py), W, c_lo * KK, Cout, Cs, 8, KK, Cin * KK, _xpair_taps(py))
                    dd = ConvDesc(N, Cout, 0, Ho, Wo, Cs, CP, Ho, Wo, Hin, Win, 2, 2, 1, 1 - py, 0, _lib.PAD_ZEROS,
                                  2, 2, py, 0, accf, radj)
                    _xpair(dd, g, wd, None, gin, valid)
            if existing is None:
                tape.grads[src.data_ptr()] = gin
            c_lo += Cs

    tape.record(bwd)
    return out


def conv_transpose2d(ctx: Ctx, x: Tensor, wname: str, bname: str) -> Tensor:
    """nn.ConvTranspose2d(k=4, s=2, p=1) (reference conv.py:384-390) as four 2x2 parity sub-convolutions."""
    Wt = ctx.p[wname]
    Cin, Cout = Wt.shape[0], Wt.shape[1]
    N, _, H, Wd = x.shape
    out = torch.empty(N, Cout, 2 * H, 2 * Wd, device=x.device, dtype=torch.float32)
    bias = ctx.p[bname]
    for py in range(2):
        wpk, CP = ctx.pack((wname, "fwd", py), Wt, 0, Cin, Cout, 8, 16, Cout * 16, _xpair_taps(py))
        d = ConvDesc(N, Cin, 0, H, Wd, Cout, CP, H, Wd, 2 * H, 2 * Wd, 2, 2, 1, 1 - py, 0, _lib.PAD_ZEROS, 2, 2, py, 0, 0)
        _xpair(d, x, wpk, bias, out, None)
    if ctx.tape is None:
        return out
    tape = ctx.tape
    tape.track(out)

    def bwd():
        g = tape.pop_grad(out)
        if g is None:
            return
        gw, acc = ctx.grad_sink(wname)
        # dW[ci,co,k] = sum x[ci,p] * g[co, 2p+k-1]  ==  conv4x4s2 weight gradient with (input=g, gout=x)
        _wgrad(ctx, [g], x, Cin, H, Wd, 4, 2, 1, _lib.PAD_ZEROS, gw, Cout * 16, 16, list(range(16)), acc, None)
        existing = tape.grad_of(x)
        gin = existing if existing is not None else torch.empty_like(x)
        wd, CP = ctx.pack((wname, "dgrad"), Wt, 0, Cout, Cin, 16, Cout * 16, 16, list(range(16)))
        dd = ConvDesc(N, Cout, 0, 2 * H, 2 * Wd, Cin, CP, H, Wd, H, Wd, 4, 4, 2, 1, 1, _lib.PAD_ZEROS, 1, 1, 0, 0,
                      1 if existing is not None else 0)
        _igemm(dd, g, None, wd, None, gin, None)
        if existing is None:
            tape.grads[x.data_ptr()] = gin

    tape.record(bwd)
    return out


def depthwise_conv2d(ctx: Ctx, x: Tensor, wname: str, K: int, S: int, pad: int, pad_mode: int,
                     valid: Optional[Tensor], bname: Optional[str] = None) -> Tensor:
    """Depthwise part of DepthwiseSeparableConv2D (reference conv.py:18-20; bias-free) and the depthwise convolution of
    MBConv (mbconv.py:71-79; `bname`: its bias, whose gradient the following norm_act(conv_bias=bname) delivers)."""
    W = ctx.p[wname]
    N, Cc, Hin, Win = x.shape
    Ho = (Hin + 2 * pad - K) // S + 1
    Wo = (Win + 2 * pad - K) // S + 1
    out = torch.empty(N, Cc, Ho, Wo, device=x.device, dtype=torch.float32)
    check(lib().c2s_dwconv_fwd(x.data_ptr(), W.data_ptr(), out.data_ptr(), _ptr(valid), N, Cc, Hin, Win, K, S, pad,
                               pad_mode, _stream()), "dwconv_fwd")
    if bname is not None:
        check(lib().c2s_channel_bias_add(out.data_ptr(), ctx.p[bname].data_ptr(), _ptr(valid), N, Cc, Ho * Wo, _stream()),
              "channel_bias_add")
    if ctx.tape is None:
        return out
    tape = ctx.tape
    tape.track(out)

    def bwd():
        g = tape.pop_grad(out)
        if g is None:
            return
        gw, acc = ctx.grad_sink(wname)
        part = ctx.ws.get("dw_partial", N * Cc * K * K)
        tgt = gw if not acc else torch.empty_like(gw)

        def wgrad():
            check(lib().c2s_dwconv_wgrad(x.data_ptr(), g.data_ptr(), part.data_ptr(), tgt.data_ptr(), _ptr(valid), N, Cc, Hin,
                                         Win, K, S, pad, pad_mode, _stream()), "dwconv_wgrad")
            if acc:
                check(lib().c2s_add_inplace(gw.data_ptr(), tgt.data_ptr(), tgt.numel(), _stream()), "add_inplace")

        if _side_ok():
            tape.defer(wgrad, [x, g, tgt])              # side stream, next to the data-gradient chain
        else:
            wgrad()
        existing = tape.grad_of(x)                      # e.g. the residual branch of the block: accumulate in the kernel
        gin = existing if existing is not None else torch.empty_like(x)
        check(lib().c2s_dwconv_dgrad(g.data_ptr(), W.data_ptr(), gin.data_ptr(), _ptr(valid), N, Cc, Hin, Win, K, S, pad,
                                     pad_mode, 1 if existing is not None else 0, _stream()), "dwconv_dgrad")
        if existing is None:
            tape.grads[x.data_ptr()] = gin

    tape.record(bwd)
    return out


# =================================================================================================
# normalisation (+ReLU, +residual)
# =================================================================================================
# One-pass normalisation (csrc/norm.hip): statistics / sums and the apply step in ONE read of the activation, the waves of a
# group meeting through memory.  C2S_NORM_ONEPASS=0 keeps the two-pass kernels (A/B runs, and the shapes the one-pass
# form does not take fall back to them anyway).
ONEPASS_NORM = _os.environ.get("C2S_NORM_ONEPASS", "1") != "0"
# Small planes: a wave holds too little data for the meeting to pay (tools/norm_bench.py, isolated: 32x32 planes break even,
# 16x16 planes lose 30-140 %).  Inside a step 32x32 planes still win (one launch instead of two per direction: U-TAE 12.26 ->
# 12.11 ms over three A/B runs), 16x16 planes do not (12.10 vs 12.04): planes below 1024 pixels keep the two-pass kernels.
ONEPASS_MIN_HW = int(_os.environ.get("C2S_NORM_ONEPASS_MIN_HW", "1024"))


def norm_act(ctx: Ctx, x: Tensor, prefix: str, kind: int, groups: int, relu: bool, residual: Optional[Tensor],
             valid: Optional[Tensor], pad_value: float = 0.0, conv_bias: Optional[str] = None, affine: bool = True) -> Tensor:
    """GroupNorm / BatchNorm (+ReLU) (+ residual add) on NCHW x.  `conv_bias` names the bias of the convolution
    that produced x: its gradient (= per-channel sum of dx) falls out of the same reduction.  affine=False: no learnable
    gain / shift (nn.InstanceNorm2d): the kernels run with constant ones / zeros and their gradients are discarded."""
    N, Cc = x.shape[0], x.shape[1]
    HW = x[0, 0].numel()
    batch = kind == _lib.NORM_BATCH
    training = 1 if (ctx.training or not batch) else 0
    d = NormDesc(N, Cc, HW, kind, groups if not batch else 1, training if batch else 1, ctx.eps, ctx.momentum)
    if affine:
        gamma, beta = ctx.p[prefix + ".weight"], ctx.p[prefix + ".bias"]
    else:
        gamma = ctx.ws.get(f"ones{Cc}", Cc)
        beta = ctx.ws.get(f"zeros{Cc}", Cc)
        check(lib().c2s_fill(gamma.data_ptr(), Cc, 1.0, _stream()), "fill")
        check(lib().c2s_fill(beta.data_ptr(), Cc, 0.0, _stream()), "fill")
    rm = ctx.b.get(prefix + ".running_mean") if batch else None
    rv = ctx.b.get(prefix + ".running_var") if batch else None
    ngroups = Cc if batch else N * groups
    gstats = torch.empty(ngroups * 2, device=x.device, dtype=torch.float32)
    row_ab = torch.empty(N * Cc * 3, device=x.device, dtype=torch.float32)   # per row: (scale, beta, mean)
    nws = lib().c2s_norm_workspace_floats(C.byref(d))
    ws = ctx.ws.get("norm", nws)
    y = torch.empty_like(x)
    nbt = ctx.b.get(prefix + ".num_batches_tracked") if (batch and ctx.training) else None   # int64, bumped by the kernel
    sync_bytes = (lib().c2s_norm_onepass_sync_bytes(C.byref(d), 1 if valid is not None else 0)
                  if ONEPASS_NORM and HW >= ONEPASS_MIN_HW else 0)
    if sync_bytes:
        sync = ctx.ws.sync_area(sync_bytes)
        check(lib().c2s_norm_fwd_onepass(C.byref(d), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(rm), _ptr(rv),
                                         _ptr(nbt), gstats.data_ptr(), row_ab.data_ptr(), _ptr(residual), y.data_ptr(),
                                         int(relu), _ptr(valid), float(pad_value), sync.data_ptr(), sync.numel(), _stream()),
              "norm_fwd_onepass")
    else:
        check(lib().c2s_norm_fwd(C.byref(d), x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _ptr(rm), _ptr(rv), _ptr(nbt),
                                 gstats.data_ptr(), row_ab.data_ptr(), _ptr(residual), y.data_ptr(), int(relu),
                                 ws.data_ptr(), ws.numel(), _ptr(valid), float(pad_value), _stream()), "norm_fwd")
    if ctx.tape is None:
        return y
    tape = ctx.tape
    tape.track(y)

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        if residual is not None:
            tape.add_grad(residual, g, own=True)      # the residual branch keeps g itself ...
            gx = torch.empty_like(g)                  # ... and the norm gradient goes to a fresh buffer (no copy of g)
        else:
            gx = g  # in place
        if affine:
            dgamma, _ = ctx.grad_sink(prefix + ".weight")
            dbeta, _ = ctx.grad_sink(prefix + ".bias")
        else:
            dgamma = torch.empty(Cc, device=x.device, dtype=torch.float32)
            dbeta = torch.empty(Cc, device=x.device, dtype=torch.float32)
        dbias = ctx.grad_sink(conv_bias)[0] if conv_bias else None

        def norm_bwd(ws_, dg_, db_, dbi_):
            if sync_bytes:
                sync = ctx.ws.sync_area(sync_bytes)
                check(lib().c2s_norm_bwd_onepass(C.byref(d), x.data_ptr(), g.data_ptr(), gamma.data_ptr(), gstats.data_ptr(),
                                                 row_ab.data_ptr(), int(relu), gx.data_ptr(), _ptr(dg_), _ptr(db_), _ptr(dbi_),
                                                 ws_.data_ptr(), ws_.numel(), _ptr(valid), sync.data_ptr(), sync.numel(),
                                                 _stream()), "norm_bwd_onepass")
            else:
                check(lib().c2s_norm_bwd(C.byref(d), x.data_ptr(), g.data_ptr(), gamma.data_ptr(), gstats.data_ptr(),
                                         row_ab.data_ptr(), int(relu), gx.data_ptr(), _ptr(dg_), _ptr(db_), _ptr(dbi_),
                                         ws_.data_ptr(), ws_.numel(), _ptr(valid), _stream()), "norm_bwd")

        if _side_ok():
            # the parameter gradients (one wave per channel: a launch that leaves the GPU idle) go to the side stream; their
            # partial sums live in a buffer of their own until the join
            ws2 = torch.empty(nws, device=x.device, dtype=torch.float32)
            norm_bwd(ws2, None, None, None)
            tape.defer(lambda: check(lib().c2s_norm_bwd_params(C.byref(d), ws2.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                                               _ptr(dbias), _ptr(valid), _stream()), "norm_bwd_params"),
                       [ws2] if affine else [ws2, dgamma, dbeta])     # affine=False: the discarded sums are temporaries too
        else:
            norm_bwd(ctx.ws.get("norm", nws), dgamma, dbeta, dbias)
        tape.add_grad(x, gx)

    tape.record(bwd)
    return y


def squeeze_excite(ctx: Ctx, x: Tensor, prefix: str, valid: Optional[Tensor], pad_value: float = 0.0) -> Tensor:
    """SqueezeAndExcitation (reference squeeze_and_excitation.py:7-30): x * sigmoid(W2 relu(W1 mean_hw(x))) per frame;
    `prefix` names the module (its Linear layers are prefix.sae.1 and prefix.sae.3, bias-free)."""
    W1, W2 = ctx.p[prefix + ".sae.1.weight"], ctx.p[prefix + ".sae.3.weight"]
    N, Cc = x.shape[:2]
    HW = x[0, 0].numel()
    R = Cc // 16
    assert tuple(W1.shape) == (R, Cc) and tuple(W2.shape) == (Cc, R), (prefix, W1.shape, W2.shape)
    dev = x.device
    pooled = torch.empty(N, Cc, device=dev, dtype=torch.float32)
    hidden = torch.empty(N, R, device=dev, dtype=torch.float32)
    scale = torch.empty(N, Cc, device=dev, dtype=torch.float32)
    y = torch.empty_like(x)
    nws = lib().c2s_se_workspace_floats(N, Cc, HW)
    ws = ctx.ws.get("se", nws)
    check(lib().c2s_se_fwd(x.data_ptr(), W1.data_ptr(), W2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(), scale.data_ptr(),
                           y.data_ptr(), _ptr(valid), N, Cc, HW, float(pad_value), ws.data_ptr(), ws.numel(), _stream()), "se_fwd")
    if ctx.tape is None:
        return y
    tape = ctx.tape
    tape.track(y)

    def bwd():
        g = tape.pop_grad(y)
        if g is None:
            return
        g1, a1 = ctx.grad_sink(prefix + ".sae.1.weight")
        g2, a2 = ctx.grad_sink(prefix + ".sae.3.weight")
        ws_ = ctx.ws.get("se", nws)
        check(lib().c2s_se_bwd(x.data_ptr(), g.data_ptr(), W1.data_ptr(), W2.data_ptr(), pooled.data_ptr(), hidden.data_ptr(),
                               scale.data_ptr(), g.data_ptr(), g1.data_ptr(), g2.data_ptr(), a1, a2, _ptr(valid), N, Cc, HW,
                               ws_.data_ptr(), ws_.numel(), _stream()), "se_bwd")
        tape.add_grad(x, g)

    tape.record(bwd)
    return y


# =================================================================================================
# temporal aggregation
# =================================================================================================
def temporal_aggregate(ctx: Ctx, x5: Tensor, attn: Tensor, valid: Optional[Tensor], n_head: int,
                       mode: str = "att_group") -> Tensor:
    """TemporalAggregator (reference temporal_aggregator.py:14-77).  att_group: head g weights channel group g;
    att_mean: the head-averaged attention weights every channel; mean: plain mean over the valid frames.  The two
    off-default modes run the same kernels on a derived weight tensor (c2s_attn_head_mean / c2s_frame_mean_weights)."""
    B, T, Cc, H, W = x5.shape
    src_attn = attn
    if mode == "att_mean":
        attn = torch.empty_like(src_attn)
        check(lib().c2s_attn_head_mean(src_attn.data_ptr(), attn.data_ptr(), n_head, src_attn[0].numel(), _stream()), "attn_head_mean")
    elif mode == "mean":
        attn = torch.empty(n_head, B, T, 1, 1, device=x5.device, dtype=torch.float32)
        check(lib().c2s_frame_mean_weights(_ptr(valid), attn.data_ptr(), n_head, B, T, _stream()), "frame_mean_weights")
    elif mode != "att_group":
        raise ValueError(f"agg_mode {mode!r}")
    h, w = attn.shape[-2:]
    d = AggDesc(B, T, Cc, H, W, n_head, h, w)
    out = torch.empty(B, Cc, H, W, device=x5.device, dtype=torch.float32)
    check(lib().c2s_temporal_aggregate_fwd(C.byref(d), x5.data_ptr(), attn.data_ptr(), _ptr(valid), out.data_ptr(),
                                           _stream()), "temporal_aggregate_fwd")
    if ctx.tape is None:
        return out
    tape = ctx.tape
    tape.track(out)

    def bwd():
        g = tape.pop_grad(out)
        if g is None:
            return
        existing = tape.grad_of(x5)
        gx = existing if existing is not None else torch.empty_like(x5)
        if mode == "att_group":
            gattn = tape.grad_of(attn)
            if gattn is None:
                gattn = torch.zeros_like(attn)
                tape.grads[attn.data_ptr()] = gattn
        else:
            gattn = torch.zeros_like(attn)          # gradient of the derived weights
        nws = lib().c2s_temporal_aggregate_bwd_workspace_floats(C.byref(d))
        ws = ctx.ws.get("agg", nws)
        check(lib().c2s_temporal_aggregate_bwd(C.byref(d), x5.data_ptr(), attn.data_ptr(), _ptr(valid), g.data_ptr(),
                                               gx.data_ptr(), 1 if existing is not None else 0, gattn.data_ptr(),
                                               ws.data_ptr(), ws.numel(), _stream()), "temporal_aggregate_bwd")
        if mode == "att_mean":
            cur = tape.grad_of(src_attn)
            tgt = cur if cur is not None else torch.empty_like(src_attn)
            check(lib().c2s_attn_head_mean_bwd(gattn.data_ptr(), tgt.data_ptr(), n_head, src_attn[0].numel(),
                                               1 if cur is not None else 0, _stream()), "attn_head_mean_bwd")
            if cur is None:
                tape.grads[src_attn.data_ptr()] = tgt
        if existing is None:
            tape.grads[x5.data_ptr()] = gx

    tape.record(bwd)
    return out


# =================================================================================================
# L-TAE
# =================================================================================================
def positional_table(dates: Tensor, d: int, period: float) -> Tensor:
    """[B,T] int days -> [B,T,16] sinusoid table (reference positional_encoding.py:16-33)."""
    assert d == 16, "the kernels are built for d_model / n_head = 16"
    dl = dates.to(torch.int64).contiguous()
    pe = torch.empty(*dl.shape, d, device=dl.device, dtype=torch.float32)
    check(lib().c2s_positional_table(dl.data_ptr(), pe.data_ptr(), dl.numel(), float(period), _stream()), "positional_table")
    return pe


# "abs_rel_doy" / "abs_rel_linear": use_abs_rel_enc together with use_doy / add_linear (tae.py:407-423): the first encoder
# (kernel mode 1 / 3 on dates[...,0]) plus the AbsolutePositionalEncoder `positional_encoder_abs` on dates[...,1]
PE_MODES = {"rel": 0, "doy": 1, "abs_rel": 2, "linear": 3, "abs_rel_doy": 4, "abs_rel_linear": 5}


def ltae_attention(ctx: Ctx, x5: Tensor, dates: Tensor, valid: Optional[Tensor], prefix: str, n_head: int, d_k: int,
                   d_model: int, period: float, dropout_p: float, with_embedding: bool, seed: int,
                   keep: Optional[Tensor], seed_dev: Optional[Tensor] = None, pe_mode: str = "rel",
                   need_attn: bool = True) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """L-TAE steps 1-6 (reference tae.py:451-481, 738-847).  Returns (emb [B,d_model,h,w] | None, attn [H,B,T,h,w]).
    pe_mode: "rel" = the default sinusoid of the relative dates; "doy" / "abs_rel" / "linear" = the learnable encoders of
    use_doy / use_abs_rel_enc (dates [B,T,2]) / add_linear (tae.py:404-430): the attention kernels then run with a zero
    table and the general table enters next to them (csrc/ltae_pe.hip).
    need_attn=False: the caller never reads the post-dropout weights (TimeUNet_v1 without return_att): where the kernels allow
    it (c2s_ltae_attn_optional) they are not stored -- the returned attn is then None -- and the backward re-derives the
    keep flags from the forward's counter hash."""
    B, T, Cc, h, w = x5.shape
    HW = h * w
    Q = ctx.p[prefix + ".attention_head.Q"]
    Wk = ctx.p[prefix + ".attention_head.fc1_k.weight"]
    bk = ctx.p[prefix + ".attention_head.fc1_k.bias"]
    Wc3 = ctx.p[prefix + ".inconv.weight"]
    bc = ctx.p[prefix + ".inconv.bias"]
    gamma, beta = ctx.p[prefix + ".in_norm.weight"], ctx.p[prefix + ".in_norm.bias"]
    assert n_head == 16 and d_k == 4 and d_model == 256, "the fold kernels are built for n_head=16, d_k=4, d_model=256"
    mode = PE_MODES[pe_mode]
    dev = x5.device
    pe256 = sin256 = d0 = d1 = None
    if mode == 0:
        pe = positional_table(dates, d_model // n_head, period)
    else:
        pe = ctx.ws.get("ltae_pe_zero", B * T * 16)
        check(lib().c2s_fill(pe.data_ptr(), B * T * 16, 0.0, _stream()), "fill")
        dl = dates.to(torch.int64)
        two = mode in (2, 4, 5)                      # dates [B,T,2]: (relative date, day of year)
        kmode = {4: 1, 5: 3}.get(mode, mode)         # the table kernel's mode for the first encoder
        d0 = (dl[..., 0] if two else dl).contiguous()
        d1 = dl[..., 1].contiguous() if two else None
        enc = prefix + (".positional_encoder_abs.fc" if mode == 2 else ".positional_encoder.fc")
        enc2 = prefix + ".positional_encoder_abs.fc" if mode in (4, 5) else None
        peW, peb = ctx.p[enc + ".weight"], ctx.p[enc + ".bias"]
        pe256 = torch.empty(B, T, d_model, device=dev, dtype=torch.float32)
        sin256 = torch.empty(B, T, d_model, device=dev, dtype=torch.float32) if kmode == 3 else None
        bad = ctx.ws.bufs.get("ltae_pe_bad")
        if bad is None:
            bad = ctx.ws.bufs["ltae_pe_bad"] = torch.zeros(1, device=dev, dtype=torch.int32)
        check(lib().c2s_ltae_pe_table(kmode, d0.data_ptr(), _ptr(d1), float(period), peW.data_ptr(), peb.data_ptr(),
                                      pe256.data_ptr(), _ptr(sin256), bad.data_ptr(), B * T, _stream()), "ltae_pe_table")
        if enc2 is not None:
            check(lib().c2s_ltae_pe_abs_add(d1.data_ptr(), ctx.p[enc2 + ".weight"].data_ptr(), ctx.p[enc2 + ".bias"].data_ptr(),
                                            pe256.data_ptr(), bad.data_ptr(), B * T, _stream()), "ltae_pe_abs_add")
    # parameter-only fold: U [16,C], s0 [B,T,16]; qwk is kept for the adjoint
    U = torch.empty(n_head, Cc, device=dev, dtype=torch.float32)
    s0 = torch.empty(B, T, n_head, device=dev, dtype=torch.float32)
    qwk = torch.empty(n_head, d_model, device=dev, dtype=torch.float32)
    check(lib().c2s_ltae_fold_fwd(Q.data_ptr(), Wk.data_ptr(), bk.data_ptr(), Wc3.data_ptr(), bc.data_ptr(), pe.data_ptr(),
                                  U.data_ptr(), s0.data_ptr(), qwk.data_ptr(), B * T, Cc, _stream()), "ltae_fold_fwd")
    if mode != 0:
        check(lib().c2s_ltae_pe_fwd(qwk.data_ptr(), pe256.data_ptr(), None, s0.data_ptr(), None, B, T, HW, 0, _stream()),
              "ltae_pe_fwd")
    Wc = Wc3.view(d_model, Cc)
    p_eff = dropout_p if ctx.training else 0.0
    d = LtaeDesc(B, T, Cc, HW, n_head, d_model, ctx.eps, p_eff, seed, _ptr(keep) if p_eff > 0 else None, _ptr(seed_dev))
    skip_attn = not need_attn and mode == 0 and with_embedding and (
        bool(lib().c2s_ltae_attn_optional(C.byref(d))) if ctx.tape is not None else lib().c2s_ltae_fwd_path(C.byref(d)) == 2)
    attn = None if skip_attn else torch.empty(n_head, B, T, h, w, device=x5.device, dtype=torch.float32)
    if skip_attn and ctx.tape is not None:      # the keep flags of the attention dropout as bits, for the backward pass
        keep_bits = torch.empty(B * HW * n_head, device=x5.device, dtype=torch.int64)
        d.keep_bits = keep_bits.data_ptr()
        ctx.tape.track(keep_bits)
    # softmax before dropout: saved for the backward (and the score scratch of the three-pass streaming kernels); a forward
    # without a tape (inference) does not store it: 16*B*T*h*w floats less to write
    need_pre = ctx.tape is not None or lib().c2s_ltae_fwd_path(C.byref(d)) == 1
    attn_pre = torch.empty(n_head, B, T, h, w, device=x5.device, dtype=torch.float32) if need_pre else None
    emb = torch.empty(B, d_model, h, w, device=x5.device, dtype=torch.float32) if with_embedding else None
    stats = torch.empty(B * HW * n_head * 2, device=x5.device, dtype=torch.float32)
    Ud, s0d = U, s0
    fws = ctx.ws.get("ltae_fwd", lib().c2s_ltae_fwd_workspace_floats(C.byref(d)))
    prof = PROFILE
    timed = prof is not None and "ltae_events" in prof
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    check(lib().c2s_ltae_attn_fwd_ws(C.byref(d), x5.data_ptr(), gamma.data_ptr(), beta.data_ptr(), Ud.data_ptr(),
                                     s0d.data_ptr(), Wc.data_ptr(), bc.data_ptr(), pe.data_ptr(), _ptr(valid),
                                     _ptr(attn), _ptr(attn_pre), _ptr(emb), stats.data_ptr(), fws.data_ptr(),
                                     fws.numel(), _stream()), "ltae_fwd")
    if timed:
        e1.record()
        prof["ltae_events"].append((e0, e1))
    if mode != 0 and emb is not None:
        check(lib().c2s_ltae_pe_fwd(None, pe256.data_ptr(), attn.data_ptr(), None, emb.data_ptr(), B, T, HW, 1, _stream()),
              "ltae_pe_fwd")
    if ctx.tape is None:
        return emb, attn
    tape = ctx.tape
    if attn is not None:
        tape.track(attn)
    if emb is not None:
        tape.track(emb)

    def bwd():
        g_attn = tape.pop_grad(attn) if attn is not None else None
        g_emb = tape.pop_grad(emb) if emb is not None else None
        if g_attn is None and g_emb is None:
            return
        gx = torch.empty_like(x5)
        dev = x5.device
        gU = torch.empty(n_head, Cc, device=dev)
        gs0 = torch.empty(B, T, n_head, device=dev)
        gWc = torch.empty(d_model, Cc, device=dev)
        gbc = torch.empty(d_model, device=dev)
        ggam, _ = ctx.grad_sink(prefix + ".in_norm.weight")
        gbet, _ = ctx.grad_sink(prefix + ".in_norm.bias")
        nws = lib().c2s_ltae_bwd_workspace_floats(C.byref(d))
        ws = ctx.ws.get("ltae", nws)
        if mode != 0 and g_emb is not None:          # the positional part of the values: g_a += <g_emb_h, pe_h>
            g_tot = torch.empty_like(attn)
            check(lib().c2s_ltae_pe_gattn(g_emb.data_ptr(), pe256.data_ptr(), _ptr(g_attn), g_tot.data_ptr(), B, T, HW,
                                          _stream()), "ltae_pe_gattn")
            g_attn = g_tot
        check(lib().c2s_ltae_attn_bwd(C.byref(d), x5.data_ptr(), gamma.data_ptr(), beta.data_ptr(), Ud.data_ptr(),
                                      s0d.data_ptr(), Wc.data_ptr(), bc.data_ptr(), pe.data_ptr(), _ptr(valid),
                                      _ptr(attn), attn_pre.data_ptr(), stats.data_ptr(), _ptr(g_emb), _ptr(g_attn),
                                      gx.data_ptr(), gU.data_ptr(), gs0.data_ptr(), gWc.data_ptr(), gbc.data_ptr(),
                                      ggam.data_ptr(), gbet.data_ptr(), ws.data_ptr(), ws.numel(), _stream()), "ltae_bwd")
        # adjoint of the parameter fold: final gradients of Q, fc1_k, inconv in one launch
        names = [prefix + ".attention_head.Q", prefix + ".attention_head.fc1_k.weight",
                 prefix + ".attention_head.fc1_k.bias", prefix + ".inconv.weight", prefix + ".inconv.bias"]
        sinks = [ctx.grad_sink(nme) for nme in names]
        acc_mask = sum((1 << i) for i, (_, acc) in enumerate(sinks) if acc)
        fbw = ctx.ws.get("ltae_fold_bwd", lib().c2s_ltae_fold_bwd_workspace_floats())
        check(lib().c2s_ltae_fold_bwd(Q.data_ptr(), Wk.data_ptr(), bk.data_ptr(), Wc3.data_ptr(), bc.data_ptr(), pe.data_ptr(),
                                      qwk.data_ptr(), gU.data_ptr(), gs0.data_ptr(),
                                      gWc.data_ptr() if emb is not None else None, gbc.data_ptr() if emb is not None else None,
                                      *[t.data_ptr() for t, _ in sinks], B * T, Cc, acc_mask, fbw.data_ptr(), fbw.numel(),
                                      _stream()), "ltae_fold_bwd")
        if mode != 0:
            g_pe = torch.empty(B, T, d_model, device=dev, dtype=torch.float32)
            gW, _ = ctx.grad_sink(enc + ".weight")
            gb, _ = ctx.grad_sink(enc + ".bias")
            check(lib().c2s_ltae_pe_bwd(kmode, d0.data_ptr(), _ptr(d1), Q.data_ptr(), Wk.data_ptr(), qwk.data_ptr(),
                                        pe256.data_ptr(), _ptr(sin256), attn.data_ptr(), _ptr(g_emb), gs0.data_ptr(),
                                        g_pe.data_ptr(), sinks[1][0].data_ptr(), sinks[0][0].data_ptr(), gW.data_ptr(),
                                        gb.data_ptr(), B, T, HW, _stream()), "ltae_pe_bwd")
            if enc2 is not None:
                gW2, _ = ctx.grad_sink(enc2 + ".weight")
                gb2, _ = ctx.grad_sink(enc2 + ".bias")
                check(lib().c2s_ltae_pe_abs_bwd(d1.data_ptr(), g_pe.data_ptr(), gW2.data_ptr(), gb2.data_ptr(), B * T, _stream()),
                      "ltae_pe_abs_bwd")
        tape.add_grad(x5, gx)

    tape.record(bwd)
    return emb, attn


def dropout_nchw(ctx: Ctx, x: Tensor, p: float, seed: int, keep: Optional[Tensor], seed_dev: Optional[Tensor] = None) -> Tensor:
    """nn.Dropout of the L-TAE MLP (reference tae.py:448); identity in eval mode."""
    if not ctx.training or p <= 0.0:
        return x
    B, Cc = x.shape[:2]
    HW = x[0, 0].numel()
    y = torch.empty_like(x)
    check(lib().c2s_dropout_nchw(x.data_ptr(), y.data_ptr(), B, Cc, HW, p, seed, _ptr(seed_dev), _ptr(keep), _stream()), "dropout")
    if ctx.tape is not None:
        tape = ctx.tape
        tape.track(y)

        def bwd():
            g = tape.pop_grad(y)
            if g is None:
                return
            gx = torch.empty_like(g)
            check(lib().c2s_dropout_nchw(g.data_ptr(), gx.data_ptr(), B, Cc, HW, p, seed, _ptr(seed_dev), _ptr(keep), _stream()),
                  "dropout_bwd")
            tape.add_grad(x, gx)

        tape.record(bwd)
    return y


def pixel_group_norm(ctx: Ctx, x: Tensor, prefix: str, groups: int) -> Tensor:
    """out_norm of L-TAE: GroupNorm over channel groups of each pixel (reference tae.py:437-440,488)."""
    B, Cc = x.shape[:2]
    HW = x[0, 0].numel()
    gamma, beta = ctx.p[prefix + ".weight"], ctx.p[prefix + ".bias"]
    y = torch.empty_like(x)
    stats = torch.empty(B * groups * HW * 2, device=x.device, dtype=torch.float32)
    check(lib().c2s_pixel_gn_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), y.data_ptr(), stats.data_ptr(), B, Cc,
                                 HW, groups, ctx.eps, _stream()), "pixel_gn_fwd")
    if ctx.tape is not None:
        tape = ctx.tape
        tape.track(y)

        def bwd():
            g = tape.pop_grad(y)
            if g is None:
                return
            gx = torch.empty_like(x)
            dg, _ = ctx.grad_sink(prefix + ".weight")
            db, _ = ctx.grad_sink(prefix + ".bias")
            nws = lib().c2s_pixel_gn_bwd_workspace_floats(B, Cc, HW)
            ws = ctx.ws.get("pixel_gn", nws)
            check(lib().c2s_pixel_gn_bwd(x.data_ptr(), g.data_ptr(), gamma.data_ptr(), stats.data_ptr(), gx.data_ptr(),
                                         dg.data_ptr(), db.data_ptr(), B, Cc, HW, groups, ws.data_ptr(), ws.numel(),
                                         _stream()), "pixel_gn_bwd")
            tape.add_grad(x, gx)

        tape.record(bwd)
    return y


# =================================================================================================
# loss / optimiser
# =================================================================================================
def cross_entropy(logits: Tensor, target: Tensor, class_w: Tensor, ws: Workspace, want_grad: bool,
                  label_smoothing: float = 0.0, ignore_index: int = -100) -> Tuple[Tensor, Optional[Tensor]]:
    """nn.CrossEntropyLoss(weight=class_w, label_smoothing=...) (reference train.py:463-468; ignore_index is torch's
    default -100: the reference ignores its last class through a zero class weight).  Returns (loss[1], dlogits | None)."""
    B, K = logits.shape[:2]
    HW = logits[0, 0].numel()
    loss = torch.empty(1, device=logits.device, dtype=torch.float32)
    gl = torch.empty_like(logits) if want_grad else None
    n = lib().c2s_cross_entropy_workspace_floats(B, HW)
    w = ws.get("ce", n)
    check(lib().c2s_cross_entropy(logits.data_ptr(), target.data_ptr(), class_w.data_ptr(), loss.data_ptr(), _ptr(gl), B,
                                  K, HW, float(label_smoothing), int(ignore_index), w.data_ptr(), w.numel(), _stream()),
          "cross_entropy")
    return loss, gl


def bad_target_count(ws: Workspace) -> int:
    """Targets outside [0, K) (other than ignore_index) the last cross_entropy call on this workspace met: torch's
    CrossEntropyLoss raises on them, the kernel skips and counts them.  Reads a device value (host synchronisation)."""
    w = ws.bufs.get("ce")
    return 0 if w is None else int(float(w[w.numel() - 1]))


def adam_flat(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float = 1e-3, b1: float = 0.9,
              b2: float = 0.999, eps: float = 1e-8, grad_scale: float = 1.0, step_dev: Optional[Tensor] = None) -> None:
    check(lib().c2s_adam_flat(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), lr, b1, b2, eps, step,
                              _ptr(step_dev), grad_scale, _stream()), "adam")
