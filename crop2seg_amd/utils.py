"""Batch assembly in front of the hot path: the tail of the dataset's __getitem__ and `pad_collate`
as one HIP pass from the raw patch series to the model input in HBM.

Reference: src/datasets/s2_ts_cz_crop.py:366-374 (np.load(...).astype(np.float32), channel re-order), :393-398
((d - mean) / std), src/utils.py:14-66 (pad_tensor / pad_collate: zero frames up to the longest series, data and dates),
train.py:288-294 (norm_values re-ordered with channels_order = [2,1,0,4,5,6,3,7,8,9]).

The reference runs those steps on the host, one patch at a time, in fp32 torch ops, then copies the padded fp32 batch to
the GPU.  Here the raw series (in the storage type of the .npy files: float32, int16 or uint16) are placed back to back in
one pinned host buffer and a single kernel (c2s_collate_series) converts, re-orders, normalises, pads, writes the padded
dates and emits the per-frame flags -- reading the pinned buffer directly over PCIe ("zero_copy") or after one raw
host-to-device copy ("staged"; 16-bit sources move half the bytes of the reference's fp32 batch either way).
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import check, lib

Tensor = torch.Tensor

CHANNELS_LIKE_PASTIS = [2, 1, 0, 4, 5, 6, 3, 7, 8, 9]          # s2_ts_cz_crop.py:248
_SRC = {np.dtype(np.float32): _lib.SRC_F32, np.dtype(np.int16): _lib.SRC_I16, np.dtype(np.uint16): _lib.SRC_U16}


class SeriesCollator:
    """collate_fn for (series, dates) pairs -> (x [B,T,C,H,W] f32, dates [B,T] int64, valid [B*T] int32) on the device.

    mean / std are the reference's `norm_values` (already in output channel order); None = no normalisation.
    `max_size` pads to a fixed T (pad_collate's max_size), else to the longest series of the batch."""

    def __init__(self, channels_order: Optional[Sequence[int]] = None, mean=None, std=None, pad_value: float = 0.0,
                 device="cuda", max_size: Optional[int] = None, mode: str = "zero_copy"):
        assert mode in ("zero_copy", "staged")
        self.order = list(channels_order) if channels_order is not None else None
        self.mean = None if mean is None else np.ascontiguousarray(np.asarray(mean, dtype=np.float32))
        self.std = None if std is None else np.ascontiguousarray(np.asarray(std, dtype=np.float32))
        assert (self.mean is None) == (self.std is None), "mean and std come together"
        self.pad_value, self.device, self.max_size, self.mode = float(pad_value), torch.device(device), max_size, mode
        self._pinned: Optional[Tensor] = None        # raw series, back to back
        self._meta: Optional[Tensor] = None          # offsets [B+1] | dates [sum T_b]   (int64, pinned)
        self._dev_raw: Optional[Tensor] = None

    def _pin(self, nbytes: int) -> Tensor:
        if self._pinned is None or self._pinned.numel() < nbytes:
            self._pinned = torch.empty(max(nbytes, 1), dtype=torch.uint8).pin_memory()
        return self._pinned

    def __call__(self, series: Sequence[np.ndarray], dates: Sequence[np.ndarray]) -> Tuple[Tensor, Tensor, Tensor]:
        if self.device.type != "cuda":
            raise RuntimeError("crop2seg_amd runs on MI355X only (no CPU fallback)")
        B = len(series)
        assert B > 0 and len(dates) == B
        dt = np.dtype(series[0].dtype)
        if dt not in _SRC:
            raise TypeError(f"series dtype {dt}: float32, int16 and uint16 are built")
        _, Cs, H, W = series[0].shape
        lengths = [int(s.shape[0]) for s in series]
        for s, d in zip(series, dates):
            assert s.dtype == dt and s.shape[1:] == (Cs, H, W) and len(d) == s.shape[0]
        T = self.max_size if self.max_size is not None else max(lengths)
        assert max(lengths) <= T
        order = self.order if self.order is not None else list(range(Cs))
        Cc = len(order)
        frame = Cs * H * W
        total = sum(lengths)
        # stage: one pinned buffer, series back to back (a loader can np.load straight into `staging_view` instead)
        pin = self._pin(total * frame * dt.itemsize)
        host = pin.numpy()[: total * frame * dt.itemsize].view(dt).reshape(total, Cs, H, W)
        off = 0
        for s in series:
            host[off:off + s.shape[0]] = s
            off += s.shape[0]
        if self._meta is None or self._meta.numel() < B + 1 + total:
            self._meta = torch.empty(B + 1 + total, dtype=torch.int64).pin_memory()
        meta = self._meta.numpy()
        meta[0] = 0
        meta[1:B + 1] = np.cumsum(lengths)
        meta[B + 1:B + 1 + total] = np.concatenate([np.asarray(d, dtype=np.int64) for d in dates])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.mode == "staged":
            nby = total * frame * dt.itemsize
            if self._dev_raw is None or self._dev_raw.numel() < nby:
                self._dev_raw = torch.empty(nby, dtype=torch.uint8, device=self.device)
            self._dev_raw[:nby].copy_(pin[:nby], non_blocking=True)
            src_ptr = self._dev_raw.data_ptr()
        else:
            src_ptr = pin.data_ptr()
        meta_dev = self._meta[:B + 1 + total].to(self.device, non_blocking=True)
        x = torch.empty(B, T, Cc, H, W, device=self.device, dtype=torch.float32)
        dd = torch.empty(B, T, device=self.device, dtype=torch.int64)
        valid = torch.empty(B * T, device=self.device, dtype=torch.int32)
        order_a = (C.c_int * Cc)(*order)
        mean_a = None if self.mean is None else self.mean.ctypes.data_as(C.POINTER(C.c_float))
        std_a = None if self.std is None else self.std.ctypes.data_as(C.POINTER(C.c_float))
        if self.mean is not None:
            assert len(self.mean) == Cc and len(self.std) == Cc
        check(lib().c2s_collate_series(src_ptr, _SRC[dt], meta_dev.data_ptr(), meta_dev.data_ptr() + 8 * (B + 1), x.data_ptr(),
                                       dd.data_ptr(), valid.data_ptr(), B, T, Cc, Cs, H * W, order_a, mean_a, std_a,
                                       self.pad_value, stream), "collate_series")
        self._last = (meta_dev,)      # keep the device copy alive until the stream has consumed it
        return x, dd, valid


def pad_collate(batch, pad_value=0, max_size=None, device="cuda", channels_order=None, mean=None, std=None):
    """Drop-in for the (data, dates) part of the reference's collate function (src/utils.py:20-66) when the dataset hands
    over RAW series: batch = [((series [T_b,Cs,H,W], dates [T_b]), target [H,W]), ...] -> ((x, dates), y) on the device,
    the normalisation / band re-order of the dataset folded into the same pass."""
    coll = SeriesCollator(channels_order, mean, std, pad_value, device, max_size)
    series = [np.asarray(b[0][0]) for b in batch]
    dates = [np.asarray(b[0][1]) for b in batch]
    x, dd, _ = coll(series, dates)
    torch.cuda.current_stream().synchronize()          # the one-shot collator's pinned buffers die with it
    y = torch.stack([torch.as_tensor(np.asarray(b[1])) for b in batch], 0).to(device)
    return (x, dd), y
