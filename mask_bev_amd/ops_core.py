"""Shared plumbing of the torch-facing kernel wrappers (ops*.py): raw pointers and the current stream for the C ABI,
storage-type flags, the per-kernel HIP-event timer bench.py reads, the workspace allocator."""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from . import _lib, switches
from ._lib import MaskBevHipError, check


def _ptr(t: Optional[torch.Tensor]) -> ctypes.c_void_p:
    return ctypes.c_void_p(0 if t is None else t.data_ptr())


_RAW_STREAM = getattr(torch._C, '_cuda_getCurrentRawStream', None)
_RAW_DEVICE = getattr(torch._C, '_cuda_getDevice', None)


def _stream() -> ctypes.c_void_p:
    """The current stream's handle for the C ABI.  Through torch's raw accessors when they exist: `torch.cuda.current_stream()`
    builds a Stream object behind three Python-level device look-ups — 9 us a call, and every launch of the eager sections of a
    step asks."""
    if _RAW_STREAM is not None and _RAW_DEVICE is not None:
        return ctypes.c_void_p(_RAW_STREAM(_RAW_DEVICE()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


# storage types of activations and their flag in the C ABI (MBV_DT_F32 / MBV_DT_BF16 / MBV_DT_F16, maskbev_hip.h)
_ACT_DTYPES = (torch.float32, torch.bfloat16, torch.float16)
_LO_DTYPES = (torch.bfloat16, torch.float16)
_DT_FLAG = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def _dt_flag(dtype: torch.dtype) -> int:
    try:
        return _DT_FLAG[dtype]
    except KeyError:
        raise MaskBevHipError(f'mask_bev_amd kernels take f32, bf16 or fp16 activations, got {dtype}') from None


def lo_dtype() -> torch.dtype:
    """The 16-bit type of the current autocast region (bf16 outside one)."""
    if torch.is_autocast_enabled('cuda'):
        dt = torch.get_autocast_dtype('cuda')
        if dt in _LO_DTYPES:
            return dt
    return torch.bfloat16


def _need_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise MaskBevHipError('mask_bev_amd kernels need ROCm device tensors (no CPU fallback); got a '
                                  f'{t.device} tensor')


class KernelTimer:
    """Optional HIP-event timing of the dominant kernel of a C-ABI call (used by bench.py's roofline leg).
    Events are recorded by the library itself on the launch stream, right around that one kernel."""

    def __init__(self):
        self.enabled = False
        self.records = {}          # name -> list of (start_event, stop_event)

    def events(self, name: str):
        if not self.enabled:
            return None, None, ctypes.c_void_p(0), ctypes.c_void_p(0)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()                 # forces creation of the underlying hipEvent_t; re-recorded by the library
        b.record()
        self.records.setdefault(name, []).append((a, b))
        return a, b, ctypes.c_void_p(a.cuda_event), ctypes.c_void_p(b.cuda_event)

    def span(self, name: str):
        """Context manager: HIP events on torch's current stream around a C-ABI call that launches exactly one
        kernel on that stream (K11's optimizer step)."""
        timer = self

        class _Span:
            def __enter__(self_inner):
                self_inner.ev = None
                if timer.enabled:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record()
                    self_inner.ev = (a, b)
                return self_inner

            def __exit__(self_inner, *exc):
                if self_inner.ev is not None:
                    self_inner.ev[1].record()
                    timer.records.setdefault(name, []).append(self_inner.ev)
                return False

        return _Span()

    def summary_ms(self):
        torch.cuda.synchronize()
        return {k: [a.elapsed_time(b) for a, b in v] for k, v in self.records.items()}

    def reset(self):
        self.records = {}


TIMER = KernelTimer()


def _workspace(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=device)


# every name of this module — the underscore helpers included — is part of the package-internal surface `ops` re-exports
__all__ = [_n for _n in list(globals()) if not _n.startswith('__')]
