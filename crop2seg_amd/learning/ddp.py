"""Data-parallel exchange for the flat parameter / gradient buffers (one process per GPU).

The reference has no multi-GPU code; the hot path shards over independent patches, so the only
exchange per step is a sum of the flat fp32 gradient buffer (4.2-4.5 MB): one collective, issued once, on the
stream that produced the gradients.  Backend "nccl" is RCCL over xGMI on MI355X; "gloo" is used by the CPU tests.
This module has no HIP dependency so that the N > 1 logic is testable without a GPU.
"""
from __future__ import annotations

import os
from typing import Iterable, Optional

import torch
import torch.distributed as dist


class FlatDataParallel:
    def __init__(self, process_group: Optional[dist.ProcessGroup] = None):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = process_group
        self.world = dist.get_world_size(process_group)
        self.rank = dist.get_rank(process_group)

    def sync_parameters(self, flat_param: torch.Tensor, buffers: Optional[Iterable[torch.Tensor]] = None, src: int = 0) -> None:
        """Identical initial state on every rank (DDP semantics: torch's DistributedDataParallel broadcasts rank 0's
        parameters AND buffers at construction): the flat parameter buffer in one collective, the BatchNorm buffers
        (running_mean / running_var / num_batches_tracked -- a few hundred floats) coalesced into one more."""
        dist.broadcast(flat_param, src=src, group=self.group)
        bufs = [b for b in (buffers or []) if b is not None and b.numel() > 0]
        if not bufs:
            return
        # one flat fp64 staging tensor: exact for fp32 statistics and for int64 counters below 2^53
        flat = torch.cat([b.detach().to(torch.float64).flatten() for b in bufs])
        dist.broadcast(flat, src=src, group=self.group)
        off = 0
        for b in bufs:
            n = b.numel()
            b.copy_(flat[off:off + n].view_as(b).to(b.dtype))
            off += n

    @property
    def active(self) -> bool:
        """True when a collective is really issued (world > 1, or the one-GPU rehearsal switch)."""
        return self.world > 1 or os.environ.get("C2S_BENCH_FORCE_DIST") == "1"

    def reduce_async(self, bucket: torch.Tensor, after=()):
        """Start the sum of `bucket` (a view of the flat gradient buffer) over ranks while the caller keeps launching the rest
        of the backward pass.  On HIP tensors the collective is issued from a communication stream that first waits for the
        caller's current stream and for every stream in `after` (the weight-gradient side stream), so it starts as soon as the
        bucket's producers have finished and runs next to the encoder's backward pass.  Returns a handle whose wait() orders
        the caller's current stream (HIP) or the host (gloo) behind the collective; None when no collective is needed."""
        if not self.active:
            return None
        if bucket.is_cuda:
            comm = getattr(self, "_comm", None)
            if comm is None:
                comm = self._comm = torch.cuda.Stream(device=bucket.device)
            comm.wait_stream(torch.cuda.current_stream(bucket.device))
            for st in after:
                comm.wait_stream(st)
            with torch.cuda.stream(comm):
                return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group, async_op=True)

    def reduce_gradients(self, flat_grad: torch.Tensor) -> float:
        """Sum the flat gradient buffer over ranks in place; returns the scale (1/world) the optimiser must apply
        (folded into the Adam kernel instead of a separate divide pass)."""
        if self.active and flat_grad.numel() > 0:     # (C2S_BENCH_FORCE_DIST rehearses the launch on one GPU)
            dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM, group=self.group)
        return 1.0 / self.world
