"""Where the tensors the library allocates live in the MI355X's HBM.

The fused encode writes ``X`` and ``thr`` side by side, the decode reads ``X`` and writes the PCM; when the two tensors a
kernel streams side by side sit in stretches of VRAM of the same class the kernel runs 10-15 % slower (DESIGN.md section 3,
"placement": one class takes ~5.5 TB/s of row-per-wave writes, two take 6.9).  A caller that brings its own output tensors
(``encode_into`` / ``decode_into`` / the C ABI) decides that itself; for the tensors the reference's API makes the LIBRARY
allocate -- what ``transform``, ``global_masking_threshold``, ``inverse_transform``, ``AudioCodec.encode`` / ``decode``
return -- the library decides, once per process and device:

* the first ``AudioCodec.encode`` whose spectrum is too large to live in the 256 MiB Infinity Cache creates an
  ``ac_workspace`` (C ABI): region A for spectra, region B for thresholds and PCM, region B placed by timing the encode
  kernel on at most 8 candidate allocations (untouched 12 GiB spacers between the tries, every loser and every spacer back
  with the driver before the call returns; ~0.1 s, once);
* from then on results are carved out of the two regions (``ac_workspace_alloc_dlpack``: DLPack tensors that own their
  extent and give it back when the last view dies; an extent is reused only for work on the stream its last tenant was
  allocated for, the rule torch's caching allocator applies to its blocks); a request the regions have no room for gets a
  plain ``torch.empty`` -- never an error, never a copy.

Two things torch's allocator does for its own blocks and a DLPack tensor's storage does not get from it:

* **other streams**: ``Tensor.record_stream`` does not reach these tensors.  A consumer that uses a result on ANOTHER stream
  and may drop its last reference before that work has run calls ``placement.record_stream(tensor, stream)`` instead (the
  same contract: the pool makes the extent's next tenant wait for that stream's work);
* **graph capture**: under ``torch.cuda.graph`` nothing is carved out of the pool (a captured address must stay valid for
  every replay, and building the pool synchronises the device): results are plain ``torch.empty`` tensors from the capture's
  private pool, as before this module existed.

The pool is built only for the configuration its effect was measured on -- float32 mono / stereo through the fused
wave-level encode (filters_n 1024 / 2048); every other configuration gets plain allocations (``audiocodec_amd.Workspace``
remains for callers who want to probe theirs explicitly).

Memory held: the two regions, sized for two generations of the first large request (so that a loop that rebinds its
results never runs dry), capped at ``AC_PLACEMENT_MAX_GIB`` (default 16) -- ``report()`` states it, ``release()`` gives it
back.  ``AC_NO_PLACEMENT=1`` switches the whole mechanism off.  Results do not depend on any of this.
"""

from __future__ import annotations

import ctypes
import os
import threading

import torch

from . import _host, _lib

REGION_SPECTRA, REGION_OTHER = 0, 1
MIN_BYTES = 128 << 20            # X and thr smaller than this fit the 256 MiB Infinity Cache together: placement does not matter for them
_lock = threading.Lock()
_pools = {}                      # device index -> _Pool, or False when creation failed / was declined


def enabled():
    return os.environ.get("AC_NO_PLACEMENT", "") in ("", "0")


class _Pool:
    def __init__(self, handle, device, dims, report):
        self.handle, self.device, self.dims, self.report = handle, device, dims, report

    def alloc(self, region, shape):
        lib = _lib.load()
        shp = (ctypes.c_int64 * len(shape))(*[int(v) for v in shape])
        stream = torch.cuda.current_stream(self.device).cuda_stream
        mt = lib.ac_workspace_alloc_dlpack(self.handle, int(region), len(shape), shp, ctypes.c_void_p(stream))
        if not mt:
            return None
        return _tensor_from_managed(mt)


_PyCapsule_New = ctypes.pythonapi.PyCapsule_New
_PyCapsule_New.restype = ctypes.py_object
_PyCapsule_New.argtypes = [ctypes.c_void_p, ctypes.c_char_p, ctypes.c_void_p]
_DELETER = ctypes.CFUNCTYPE(None, ctypes.c_void_p)


class _DLManagedHead(ctypes.Structure):   # enough of DLManagedTensor to find its deleter (dlpack.h v0.8 layout, 64-bit)
    _fields_ = [("data", ctypes.c_void_p), ("device_type", ctypes.c_int32), ("device_id", ctypes.c_int32),
                ("ndim", ctypes.c_int32), ("dtype_code", ctypes.c_uint8), ("dtype_bits", ctypes.c_uint8),
                ("dtype_lanes", ctypes.c_uint16), ("shape", ctypes.c_void_p), ("strides", ctypes.c_void_p),
                ("byte_offset", ctypes.c_uint64), ("manager_ctx", ctypes.c_void_p), ("deleter", ctypes.c_void_p)]


def _tensor_from_managed(mt):
    """torch tensor that owns the DLManagedTensor* `mt` (its deleter -- native code -- runs when the storage dies)."""
    try:
        cap = _PyCapsule_New(mt, b"dltensor", None)
        return torch.from_dlpack(cap)
    except Exception:
        head = _DLManagedHead.from_address(mt)
        _DELETER(head.deleter)(mt)           # nobody consumed the capsule: give the extent back
        raise


def pool(device):
    """The pool of ``device`` (None when there is none -- yet, or at all)."""
    p = _pools.get(device.index if device.index is not None else torch.cuda.current_device())
    return p or None


def ensure(codec, x_shape, device):
    """Called by ``AudioCodec.encode``: creates the device's pool on the first request large enough to need one."""
    if not enabled() or codec.compute_dtype != torch.float32 or torch.cuda.is_current_stream_capturing():
        return None
    # (only where the class effect was measured: the fused wave-level encode on mono / stereo rows; elsewhere the search's stop
    # criterion -- a rate of THAT kernel -- can never be met and the probe would run its full length inside a user's call)
    if x_shape[2] > 2 or not (codec.mdct.is_fast(device) and codec.psy.is_fast(device)):
        return None
    idx = device.index if device.index is not None else torch.cuda.current_device()
    p = _pools.get(idx)
    if p is not None:
        return p or None
    B, S, C = x_shape
    N = codec.filters_n
    K = S // N
    if B < 1 or K < 1 or 4 * B * (K + 1) * N * C < MIN_BYTES:
        return None
    with _lock:
        p = _pools.get(idx)
        if p is not None:
            return p or None
        lib = _lib.load()
        cap_gib = float(os.environ.get("AC_PLACEMENT_MAX_GIB", "16"))
        per_copy = 4.0 * B * ((K + 1) * N * C * 2 + (K + 2) * N * C) / 2 ** 30     # X + thr + decoded PCM
        copies = 2 if (2 * per_copy + 4.0 * B * K * N * C / 2 ** 30) <= cap_gib else 1
        if copies * per_copy > cap_gib or torch.cuda.mem_get_info(device)[0] / 2 ** 30 < 2 * copies * per_copy + 4:
            _pools[idx] = False       # would not fit the cap (or the device): plain allocations from here on
            return None
        handle = ctypes.c_void_p()
        with _host.on_device(device):
            st = lib.ac_workspace_create(codec.mdct._plan(device), codec.psy._plan(device), B, K, C, copies, 8, 96.0,
                                         _host.stream_ptr(device), ctypes.byref(handle))
        if st != _lib.AC_OK:
            _pools[idx] = False
            return None
        tries, chosen, spacer = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        ms = (ctypes.c_float * 16)()
        a, b, na, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        lib.ac_workspace_report(handle, ctypes.byref(tries), ctypes.byref(chosen), ms, ctypes.byref(spacer))
        lib.ac_workspace_regions(handle, ctypes.byref(a), ctypes.byref(na), ctypes.byref(b), ctypes.byref(nb))
        rep = {"sized_for": {"batches_n": B, "blocks_n": K, "channels_n": C, "filters_n": N, "generations": copies},
               "bytes_held": int(na.value + nb.value), "region_spectra_bytes": int(na.value), "region_other_bytes": int(nb.value),
               "tries": tries.value, "chosen_try": chosen.value, "encode_ms_by_try": [round(float(ms[i]), 4) for i in range(tries.value)],
               "spacer_GiB_during_search": spacer.value, "cap_GiB": cap_gib}
        p = _pools[idx] = _Pool(handle, device, (B, K, C, N), rep)
        return p


def empty(region, shape, dtype, device):
    """A tensor for a result of the library: from the device's pool when there is one with room, else ``torch.empty``."""
    if dtype == torch.float32 and not torch.cuda.is_current_stream_capturing():
        p = pool(device)
        if p is not None:
            t = p.alloc(region, shape)
            if t is not None:
                return t
    return torch.empty(shape, dtype=dtype, device=device)


def record_stream(tensor, stream):
    """The pool's ``Tensor.record_stream``: ``tensor`` (a result of the library, or a view that starts where it starts) is
    also used by work on ``stream``; its memory will not be rewritten before that work has run.  A tensor that does not come
    from the pool gets torch's own ``record_stream``."""
    p = pool(tensor.device) if tensor.is_cuda else None
    sp = stream.cuda_stream if hasattr(stream, "cuda_stream") else int(stream)
    if p is not None:
        st = _lib.load().ac_workspace_record_stream(p.handle, ctypes.c_void_p(tensor.untyped_storage().data_ptr()), ctypes.c_void_p(sp))
        if st == _lib.AC_OK:
            return
    tensor.record_stream(stream)


def report(device=None):
    """What the pool of ``device`` holds and how its placement was found (None: no pool)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    p = pool(dev)
    if p is None:
        return None
    return dict(p.report, live_tensors=int(_lib.load().ac_workspace_live(p.handle)))


def release(device=None):
    """Gives the pool's memory back (at once, or when the last tensor carved out of it dies) and forgets the pool: the next
    large ``encode`` builds a new one."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    with _lock:
        p = _pools.pop(idx, None)
    if p:
        _lib.load().ac_workspace_destroy(p.handle)
