"""Host-side helpers shared by the two classes: dtype handling, tensor checks, per-device plans."""

from __future__ import annotations

import ctypes
import threading
import weakref

import torch

from . import _lib

_DTYPE_NAMES = {
    "float32": torch.float32, "fp32": torch.float32, "float": torch.float32,
    "float64": torch.float64, "fp64": torch.float64, "double": torch.float64,
    "bfloat16": torch.bfloat16, "bf16": torch.bfloat16,
    "float16": torch.float16, "fp16": torch.float16, "half": torch.float16,
}


def as_torch_dtype(dtype):
    """Accept torch dtypes, numpy dtypes / type objects and strings where the reference takes a tf.DType."""
    if isinstance(dtype, torch.dtype):
        return dtype
    name = getattr(dtype, "name", None) or getattr(dtype, "__name__", None) or str(dtype)
    name = name.replace("torch.", "").replace("tf.", "").lower()
    if name in _DTYPE_NAMES:
        return _DTYPE_NAMES[name]
    raise TypeError("unsupported dtype %r" % (dtype,))


# AC_F32 / AC_F64 / AC_BF16 of include/audiocodec_amd.h (AC_F16 = 3: the filter bank only, MDCT_DTYPE_IDS)
DTYPE_IDS = {torch.float32: 0, torch.float64: 1, torch.bfloat16: 2}
MDCT_DTYPE_IDS = {**DTYPE_IDS, torch.float16: 3}


def require_hip_compute_dtype(compute_dtype, who, filter_bank=False):
    """float32 (wave-level / LDS-FFT / generic kernels), float64 (float64 kernels and constants) and bfloat16
    (bfloat16 tensors, float32 arithmetic) have HIP kernels; the filter bank also takes float16 tensors (float32 arithmetic),
    as the reference's does (mdctransformer.py:327-344); anything else is refused."""
    ids = MDCT_DTYPE_IDS if filter_bank else DTYPE_IDS
    if compute_dtype not in ids:
        raise NotImplementedError(
            "%s: the HIP kernels serve compute_dtype float32, float64 and bfloat16%s (got %s); "
            "there is no CPU or other-precision fallback" % (who, ", float16" if filter_bank else "", compute_dtype))
    return ids[compute_dtype]


def precompute_id(precompute_dtype, who):
    """AC_F64 / AC_F32 for the reference's ``precompute_dtype`` keyword (mdctransformer.py:13-14,31-35;
    psychoacoustic.py:14-15,61-69): the arithmetic type the constant tables are computed in on the host."""
    dt = as_torch_dtype(precompute_dtype)
    if dt not in (torch.float64, torch.float32):
        raise NotImplementedError("%s: constants are pre-computed in float64 (the reference's default) or float32, got %s"
                                  % (who, dt))
    return dt, DTYPE_IDS[dt]


def require_float32(compute_dtype, what):
    if compute_dtype != torch.float32:
        raise NotImplementedError("%s is implemented for compute_dtype float32 only (got %s)" % (what, compute_dtype))


def check_device_tensor(t, name, dtype, ndim):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
    if t.dtype != dtype:
        raise ValueError("%s has dtype %s but compute_dtype is %s (no implicit cast, as in the reference)"
                         % (name, t.dtype, dtype))
    if t.dim() != ndim:
        raise ValueError("%s must have %d dimensions, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_cuda:
        raise RuntimeError("%s lives on %s: this package only runs on a ROCm device tensor (device='cuda'); "
                           "there is no CPU fallback" % (name, t.device))
    t = t.contiguous()
    if t.data_ptr() % 16 != 0:      # a view that starts inside an allocation: the kernels want 16-byte aligned rows
        t = t.clone()
    return t


class _NoSwitch:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_SWITCH = _NoSwitch()


def on_device(device):
    """``with on_device(dev):`` makes ``dev`` current for the call; free when it already is (the common case -- a
    ``torch.cuda.device`` context costs several microseconds, which matters for streaming-sized launches)."""
    idx = device.index
    if idx is None or idx == torch.cuda.current_device():
        return _NO_SWITCH
    return torch.cuda.device(device)


def stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def ptr(t):
    return ctypes.c_void_p(t.data_ptr())


class PlanCache:
    """One native plan per device index, created on first use and destroyed with the owner."""

    def __init__(self, owner, create, destroy):
        self._create = create
        self._plans = {}
        self._lock = threading.Lock()
        plans = self._plans
        weakref.finalize(owner, PlanCache._cleanup, plans, destroy)

    @staticmethod
    def _cleanup(plans, destroy):
        for handle in plans.values():
            try:
                destroy(handle)
            except Exception:
                pass
        plans.clear()

    def get(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        handle = self._plans.get(idx)
        if handle is None:
            with self._lock:
                handle = self._plans.get(idx)
                if handle is None:
                    out = ctypes.c_void_p()
                    _lib.check(self._create(idx, ctypes.byref(out)))
                    handle = out
                    self._plans[idx] = handle
        return handle
