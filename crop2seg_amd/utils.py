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
                 device="cuda", max_size: Optional[int] = None, mode: str = "zero_copy", slots: int = 2,
                 add_ndvi: bool = False, ndvi_bands: Optional[Tuple[int, int]] = None,
                 channels_like_pastis: Optional[bool] = None):
        """add_ndvi: append the dataset's NDVI channel (s2_ts_cz_crop.py:376-391,401-402): (NIR - red) / (NIR + red) of the RAW
        bands, 0 where the sum is 0 or the quotient leaves [-1, 1]; not normalised; the model then takes input_dim + 1 channels
        (train.py:315-316).  ndvi_bands = (NIR, red) as positions in the RE-ORDERED channel list; default (6, 2) with
        channels_like_pastis and (3, 0) without, as in the reference (s2_ts_cz_crop.py:384-389).  channels_like_pastis: None =
        True exactly when channels_order is the reference's PASTIS order (a custom order never silently gets bands (6, 2))."""
        assert mode in ("zero_copy", "staged")
        self.add_ndvi, self.ndvi_bands = bool(add_ndvi), ndvi_bands
        self.order = list(channels_order) if channels_order is not None else None
        self.channels_like_pastis = (self.order == CHANNELS_LIKE_PASTIS) if channels_like_pastis is None else bool(channels_like_pastis)
        self.mean = None if mean is None else np.ascontiguousarray(np.asarray(mean, dtype=np.float32))
        self.std = None if std is None else np.ascontiguousarray(np.asarray(std, dtype=np.float32))
        assert (self.mean is None) == (self.std is None), "mean and std come together"
        self.pad_value, self.device, self.max_size, self.mode = float(pad_value), torch.device(device), max_size, mode
        # `slots` staging sets (pinned raw buffer, pinned offsets/dates table, device copies, an event): the kernel (zero_copy)
        # or the H2D copies (staged) read pinned memory asynchronously, so a set is only rewritten after the event recorded
        # behind its last launch has completed -- with two sets the host can stage batch N+1 while the GPU consumes batch N
        self._slots = [dict(pinned=None, meta=None, dev_raw=None, meta_dev=None, event=None) for _ in range(max(1, int(slots)))]
        self._next = 0

    @staticmethod
    def _pin(slot: dict, nbytes: int) -> Tensor:
        if slot["pinned"] is None or slot["pinned"].numel() < nbytes:
            slot["pinned"] = torch.empty(max(nbytes, 1), dtype=torch.uint8).pin_memory()
        return slot["pinned"]

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
        ndvi_a = ndvi_b = -1
        if self.add_ndvi:
            # the reference keys the default bands on channels_like_pastis, not on whether an order is given (s2_ts_cz_crop.py:384-389)
            na, nb = self.ndvi_bands if self.ndvi_bands is not None else ((6, 2) if self.channels_like_pastis else (3, 0))
            ndvi_a, ndvi_b = order[na], order[nb]              # positions in the re-ordered list -> source channels
        Cc = len(order) + (1 if self.add_ndvi else 0)
        frame = Cs * H * W
        total = sum(lengths)
        slot = self._slots[self._next]
        self._next = (self._next + 1) % len(self._slots)
        if slot["event"] is not None:
            slot["event"].synchronize()           # the launch that last read this set's pinned memory has finished
        # stage: one pinned buffer, series back to back
        pin = self._pin(slot, total * frame * dt.itemsize)
        host = pin.numpy()[: total * frame * dt.itemsize].view(dt).reshape(total, Cs, H, W)
        off = 0
        for s in series:
            host[off:off + s.shape[0]] = s
            off += s.shape[0]
        if slot["meta"] is None or slot["meta"].numel() < B + 1 + total:
            slot["meta"] = torch.empty(B + 1 + total, dtype=torch.int64).pin_memory()
        meta = slot["meta"].numpy()
        meta[0] = 0
        meta[1:B + 1] = np.cumsum(lengths)
        meta[B + 1:B + 1 + total] = np.concatenate([np.asarray(d, dtype=np.int64) for d in dates])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        if self.mode == "staged":
            nby = total * frame * dt.itemsize
            if slot["dev_raw"] is None or slot["dev_raw"].numel() < nby:
                slot["dev_raw"] = torch.empty(nby, dtype=torch.uint8, device=self.device)
            slot["dev_raw"][:nby].copy_(pin[:nby], non_blocking=True)
            src_ptr = slot["dev_raw"].data_ptr()
        else:
            src_ptr = pin.data_ptr()
        meta_dev = slot["meta"][:B + 1 + total].to(self.device, non_blocking=True)
        x = torch.empty(B, T, Cc, H, W, device=self.device, dtype=torch.float32)
        dd = torch.empty(B, T, device=self.device, dtype=torch.int64)
        valid = torch.empty(B * T, device=self.device, dtype=torch.int32)
        order_a = (C.c_int * len(order))(*order)
        mean_a = None if self.mean is None else self.mean.ctypes.data_as(C.POINTER(C.c_float))
        std_a = None if self.std is None else self.std.ctypes.data_as(C.POINTER(C.c_float))
        if self.mean is not None:
            assert len(self.mean) == len(order) and len(self.std) == len(order)
        check(lib().c2s_collate_series_ndvi(src_ptr, _SRC[dt], meta_dev.data_ptr(), meta_dev.data_ptr() + 8 * (B + 1), x.data_ptr(),
                                            dd.data_ptr(), valid.data_ptr(), B, T, Cc, Cs, H * W, order_a, mean_a, std_a,
                                            self.pad_value, ndvi_a, ndvi_b, stream), "collate_series")
        slot["meta_dev"] = meta_dev    # the device copy stays alive until this set is reused (after its event)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        slot["event"] = ev
        return x, dd, valid


class PrefetchLoader:
    """Runs a SeriesCollator one batch ahead of the training stream.

    `batches` yields (series list, dates list, target array | None).  A worker thread copies batch N+1 into the collator's
    free pinned set (numpy copies release the GIL) and launches its collate kernel on a copy stream while the caller's
    stream is busy with batch N; the consumer side only waits on the event recorded behind that launch.  The reference
    loads with `num_workers=0` on the training thread (train.py:331-346), so its step waits for every host copy."""

    def __init__(self, batches, collator: SeriesCollator, depth: int = 1):
        import queue
        import threading
        self.coll = collator
        self._q: "queue.Queue" = queue.Queue(maxsize=max(1, depth))
        dev = collator.device
        self._dev = dev if dev.index is not None else torch.device("cuda", torch.cuda.current_device())
        self._stream = torch.cuda.Stream(device=self._dev)
        self._err: Optional[BaseException] = None
        self._stop = threading.Event()
        self._queue_mod = queue
        self._thread = threading.Thread(target=self._work, args=(iter(batches),), daemon=True)
        self._thread.start()

    def _put(self, item) -> bool:
        """Queue `item` unless the consumer went away (close()): never blocks forever on a full queue."""
        while not self._stop.is_set():
            try:
                self._q.put(item, timeout=0.1)
                return True
            except self._queue_mod.Full:
                continue
        return False

    def _work(self, it) -> None:
        try:
            torch.cuda.set_device(self._dev)
            for series, dates, target in it:
                if self._stop.is_set():
                    break
                with torch.cuda.stream(self._stream):
                    x, dd, valid = self.coll(series, dates)
                    y = None
                    if target is not None:
                        y = torch.as_tensor(np.asarray(target)).pin_memory().to(self.coll.device, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(self._stream)
                if not self._put((x, dd, valid, y, ev)):
                    break
        except BaseException as e:      # surfaced on the consumer side
            self._err = e
        finally:
            self._put(None)

    def close(self) -> None:
        """Stop the worker (a consumer that leaves the loop early calls this, or lets the object die): the worker gives up
        its queue slot, finishes the copy it is in and exits; its pinned buffers and copy stream go with the collator."""
        self._stop.set()
        try:
            while True:
                self._q.get_nowait()
        except self._queue_mod.Empty:
            pass
        if self._thread.is_alive():
            self._thread.join(timeout=5.0)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __iter__(self):
        return self

    def __next__(self):
        if self._err is not None and self._q.empty():
            self._thread.join()
            raise self._err
        item = self._q.get()
        if item is None:
            self._thread.join()
            if self._err is not None:
                raise self._err
            raise StopIteration
        x, dd, valid, y, ev = item
        cur = torch.cuda.current_stream(self.coll.device)
        cur.wait_event(ev)
        for t in (x, dd, valid, y):
            if t is not None:
                t.record_stream(cur)      # allocated on the copy stream, consumed on the caller's
        return x, dd, valid, y


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
