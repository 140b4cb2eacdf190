"""Caller-owned buffers for one encode / decode batch shape, placed for the MI355X's HBM (the Python face of the C ABI's
``ac_workspace_*``; see ``include/audiocodec_amd.h`` and DESIGN.md section 3, "placement").

The fused encode writes the spectrum ``X`` and the threshold ``thr`` side by side, the decode reads ``X`` and writes the
PCM; when the two tensors a kernel streams side by side lie in stretches of VRAM of the same class the kernel runs 10-15 %
slower.  ``ac_workspace_create`` keeps ``X``, the tonality and ``x`` in one allocation and tries a few allocations for
(``thr``, decoded PCM), timing the encode kernel on each, with untouched spacers between the tries; it keeps the fastest
and returns everything else to the driver.  Nothing about the kernels or their results changes; only where the tensors live.

Opt-in, for callers of ``encode_into`` / ``decode_into``.  The tensors ``AudioCodec.encode`` / ``decode`` allocate
themselves are placed by the same mechanism without any of this (``audiocodec_amd/placement.py``).
"""

from __future__ import annotations

import ctypes

import torch

from . import _host, _lib, placement


class Workspace:
    """``x [B, K*N, C]``, ``X [B, K+1, N, C]``, ``t [B, K+1, 1, C]``, ``thr`` like ``X``, ``xh [B, (K+2)*N, C]`` (float32).

    :param max_tries: allocations tried for (``thr``, ``xh``), at most 16; all but the chosen one are back with the driver
                      when the constructor returns.  The search stops at the first candidate that reaches the two-class
                      rate on every copy; if none does, a second region A from further along is tried against them
    :param span_gib:  upper bound on the untouched spacer memory held for a moment between the tries (0: no spacers)
    :param tune:      False: one allocation each (``max_tries = 1``)
    """

    def __init__(self, codec, batches_n, blocks_n, channels_n, max_tries=8, span_gib=96.0, tune=True, device=None):
        _host.require_float32(codec.compute_dtype, "Workspace")
        self.codec = codec
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        B, K, C, N = int(batches_n), int(blocks_n), int(channels_n), codec.filters_n
        lib = self._lib = _lib.load()
        tries = max(1, min(16, int(max_tries))) if tune else 1
        handle = ctypes.c_void_p()
        with _host.on_device(self.device):
            _lib.check(lib.ac_workspace_create(codec.mdct._plan(self.device), codec.psy._plan(self.device), B, K, C, 1, tries,
                                               float(span_gib), _host.stream_ptr(self.device), ctypes.byref(handle)))
        self._handle = handle
        pool = placement._Pool(handle, self.device, (B, K, C, N), None)
        # carved in the layout of ac_workspace_buffers: region A = [X | t | x], region B = [thr | xh]; each tensor owns its
        # extent, and the regions are freed when the workspace AND every tensor are gone
        self.X = pool.alloc(placement.REGION_SPECTRA, (B, K + 1, N, C))
        self.t = pool.alloc(placement.REGION_SPECTRA, (B, K + 1, 1, C))
        self.x = pool.alloc(placement.REGION_SPECTRA, (B, K * N, C))
        self.thr = pool.alloc(placement.REGION_OTHER, (B, K + 1, N, C))
        self.xh = pool.alloc(placement.REGION_OTHER, (B, (K + 2) * N, C))
        if any(v is None for v in (self.X, self.t, self.x, self.thr, self.xh)):
            raise RuntimeError("internal: workspace regions too small for their tensors")
        n, chosen, spacer = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
        ms = (ctypes.c_float * 16)()
        a, b, na, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
        lib.ac_workspace_report(handle, ctypes.byref(n), ctypes.byref(chosen), ms, ctypes.byref(spacer))
        lib.ac_workspace_regions(handle, ctypes.byref(a), ctypes.byref(na), ctypes.byref(b), ctypes.byref(nb))
        self.report = {"bytes_a": int(na.value), "bytes_b": int(nb.value), "tries": n.value, "tuned": tries > 1,
                       "encode_ms_by_try": [float(ms[i]) for i in range(n.value)], "chosen_try": chosen.value,
                       "spacer_GiB": spacer.value}

    def close(self):
        if getattr(self, "_handle", None) is not None:
            self._lib.ac_workspace_destroy(self._handle)   # (the regions go when the last tensor carved out of them does)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
