"""Caller-owned buffers for one encode / decode batch, placed for the MI355X's HBM.

Measured on MI355X (tools/placement_*.py): the fused encode kernel writes two streams of the same
length side by side (spectrum ``X`` and threshold ``thr``), the decode kernel reads ``X`` and writes the PCM.  The 288 GB
of VRAM fall into stretches of 8 ... 64 GiB that belong to a few classes (a map taken by moving ``thr`` across a 224 GiB
allocation shows at least three); when the two tensors a kernel streams side by side lie in stretches of the same class,
the kernel runs slower -- encode 0.57 ms against 0.50 ms on the bench workload, decode 0.37 against 0.355 -- and offsets
inside a stretch (2 MiB ... several GiB) make no difference.  Which stretch an allocation lands in is the driver's
business, so a :class:`Workspace` allocates a row of equal chunks, keeps ``x``, ``X`` and the tonality in the first, times
the encode kernel with ``thr`` in each of the others, keeps the chunk that ran fastest for ``thr`` and the decoded PCM and
returns the rest to the allocator.  Nothing about the kernels or their results changes; only where the caller's tensors
live.
"""

from __future__ import annotations

import numpy as np
import torch

from . import _host

_GIB = 1 << 30


class Workspace:
    """``x [B, K*N, C]``, ``X [B, K+1, N, C]``, ``t [B, K+1, 1, C]``, ``thr`` like ``X``, ``xh [B, (K+2)*N, C]`` (float32).

    :param span_gib: how much memory the probing may allocate for a moment (chunks of the size the tensors need, back to
                     back; afterwards two chunks stay); 0 or ``tune=False``: two chunks, no probing
    """

    def __init__(self, codec, batches_n, blocks_n, channels_n, span_gib=112.0, tune=True, device=None):
        _host.require_float32(codec.compute_dtype, "Workspace")
        self.codec = codec
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        B, K, C, N = int(batches_n), int(blocks_n), int(channels_n), codec.filters_n
        self.shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "thr": (B, K + 1, N, C),
                       "xh": (B, (K + 2) * N, C)}
        words = {k: max(int(np.prod(s)), 1) for k, s in self.shapes.items()}
        pad = (1 << 21) // 4                                     # every tensor starts on a 2 MiB boundary of its chunk
        self._words = {k: (w + pad - 1) // pad * pad for k, w in words.items()}
        need_a = self._words["x"] + self._words["X"] + self._words["t"]
        need_b = self._words["thr"] + self._words["xh"]
        chunk_words = max(need_a, need_b)
        chunk_gib = chunk_words * 4 / _GIB
        n = 2
        if tune and span_gib > 0 and B * K > 0:
            free = torch.cuda.mem_get_info(self.device)[0] / _GIB
            n = max(2, min(64, int(min(span_gib, 0.8 * free) / chunk_gib)))   # (small batches: at most 64 chunks)
        self.report = {"chunk_GiB": chunk_gib, "chunks_probed": n, "tuned": False}
        chunks = []
        try:
            for _ in range(n):
                chunks.append(torch.empty(chunk_words, dtype=torch.float32, device=self.device))
        except RuntimeError:                                      # less memory than mem_get_info promised: use what we got
            pass
        if len(chunks) < 2:
            raise RuntimeError("Workspace: not enough device memory for two chunks of %.2f GiB" % chunk_gib)
        self._a = chunks[0]
        self._carve_a()
        best = 1
        if len(chunks) > 2:
            # the library's own probe (ac_probe_placement: median of three timed encode launches per candidate)
            import ctypes
            from . import _lib
            lib = _lib.load()
            n = len(chunks) - 1
            cands = (ctypes.c_void_p * n)(*[c.data_ptr() for c in chunks[1:]])
            ms = (ctypes.c_float * n)()
            arg = ctypes.c_int(0)
            with torch.cuda.device(self.device):
                _lib.check(lib.ac_probe_placement(codec.mdct._plan(self.device), codec.psy._plan(self.device),
                                                  _host.ptr(self.x), _host.ptr(self.X), _host.ptr(self.t), cands, n, B, K, C,
                                                  _host.stream_ptr(self.device), ctypes.byref(arg), ms))
            times = [float(v) for v in ms]
            best = 1 + int(arg.value)
            self.report.update({"tuned": True, "encode_ms_by_chunk": times, "chosen_chunk": best,
                                "chunk_addresses": ["%#x" % c.data_ptr() for c in chunks]})
        self._b = chunks[best]
        self._carve_b(self._b)
        del chunks
        torch.cuda.empty_cache()                                  # the chunks that were only probed go back to the driver

    def _carve_a(self):
        o = 0
        for k in ("X", "t", "x"):
            setattr(self, k, self._a[o: o + int(np.prod(self.shapes[k]))].view(self.shapes[k]))
            o += self._words[k]

    def _carve_b(self, chunk):
        o = 0
        for k in ("thr", "xh"):
            setattr(self, k, chunk[o: o + int(np.prod(self.shapes[k]))].view(self.shapes[k]))
            o += self._words[k]
