"""``encode()`` / ``decode()`` and the streaming overlap-add wrappers.

The reference has no ``encode``/``decode``; BASELINE.json's north star names them as the API surface,
and SURVEY.md section 0 item 3 defines them as thin compositions of the reference's methods:

    encode(x)  = transform -> tonality -> global_masking_threshold   (one fused HIP pass over the PCM)
    decode(X)  = inverse_transform
"""

from __future__ import annotations

import ctypes

import torch

from . import _host, _lib, placement
from .mdctransformer import MDCTransformer
from .psychoacoustic import PsychoacousticModel


class AudioCodec:
    """MDCT analysis + psychoacoustic masking (encode) and MDCT synthesis (decode)."""

    def __init__(self, sample_rate=48000, filters_n=1024, bark_bands_n=64, alpha=0.6, window_type="vorbis",
                 compute_dtype=torch.float32, spreading=None, precompute_dtype=torch.float64):
        self.mdct = MDCTransformer(filters_n, window_type=window_type, compute_dtype=compute_dtype,
                                   precompute_dtype=precompute_dtype)
        self.psy = PsychoacousticModel(sample_rate, filter_bands_n=filters_n, bark_bands_n=bark_bands_n,
                                       alpha=alpha, compute_dtype=compute_dtype, precompute_dtype=precompute_dtype,
                                       spreading=spreading)
        self.filters_n = int(filters_n)
        self.compute_dtype = self.mdct.compute_dtype
        self._lib = _lib.load()

    def encode_launches(self, channels_n=2, device=None):
        """How many kernel launches :meth:`encode` takes for float32 tensors of ``channels_n`` channels on ``device``
        (``ac_encode_launches``): 1 = the fused kernels (filters_n 64 ... 2048 in powers of two, mono / stereo), 2 = transform +
        one masking-model pass, 3 = the generic kernels."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return int(self._lib.ac_encode_launches(self.mdct._plan(dev), self.psy._plan(dev), int(channels_n)))

    def encode(self, x, drown=0.0):
        """x [B, K*N, C] -> (X [B,K+1,N,C], tonality [B,K+1,1,C], threshold [B,K+1,N,C]).

        ``x`` is float PCM in [-1, 1] (the reference's convention) or ``torch.int16`` PCM (extension: x = pcm / 32768
        is applied inside the kernel's loads, half the PCM bytes).  When ``x`` requires a gradient the same three results
        come from the differentiable entry points in sequence (transform, tonality, threshold: each carries its backward
        pass), as the reference's op chain is differentiable in a training graph; otherwise one fused launch."""
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            X = self.mdct.transform(x)
            t = self.psy.tonality(X)
            return X, t, self.psy.global_masking_threshold(X, t, drown)
        pcm16 = isinstance(x, torch.Tensor) and x.dtype == torch.int16
        x = _host.check_device_tensor(x, "x", torch.int16 if pcm16 else self.compute_dtype, 3)
        B, S, C = x.shape
        N = self.filters_n
        if S % N != 0:
            raise ValueError("samples_n (%d) is not a multiple of filters_n (%d)" % (S, N))
        K = S // N
        # the library allocates what it returns, and decides where: spectra and thresholds in different classes of VRAM
        # (audiocodec_amd/placement.py; plain torch.empty for small batches, other dtypes, AC_NO_PLACEMENT=1)
        placement.ensure(self, (B, S, C), x.device)
        X = placement.empty(placement.REGION_SPECTRA, (B, K + 1, N, C), self.compute_dtype, x.device)
        t = torch.empty((B, K + 1, 1, C), dtype=self.compute_dtype, device=x.device)
        thr = placement.empty(placement.REGION_OTHER, (B, K + 1, N, C), self.compute_dtype, x.device)
        self.encode_into(x, X, t, thr, drown)
        return X, t, thr

    def placement_report(self, device=None):
        """How the tensors :meth:`encode` / :meth:`decode` return are placed on ``device`` (``placement.report``)."""
        return placement.report(device)

    def encode_ex(self, x, drown=0.0, noise_seed=None, db_norm=False):
        """:meth:`encode` with its element-wise tail computed in the same pass (``ac_encode_fused_ex``):

        :param noise_seed: an int: also return ``X + thr * Normal(0, 1/6)`` -- the values
                           ``psy.add_noise(X, thr, seed=noise_seed)`` gives (reference ``psychoacoustic.py:150-167``)
        :param db_norm:    also return ``psy.amplitude_to_dB_norm(X)`` (``psychoacoustic.py:87-100``)
        :return: ``(X, tonality, threshold, noisy or None, db_norm or None)``
        """
        _host.require_float32(self.compute_dtype, "encode_ex")
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():   # the differentiable composition
            X, t, thr = self.encode(x, drown)
            return (X, t, thr, self.psy.add_noise(X, thr, seed=noise_seed) if noise_seed is not None else None,
                    self.psy.amplitude_to_dB_norm(X) if db_norm else None)
        x = _host.check_device_tensor(x, "x", self.compute_dtype, 3)
        B, S, C = x.shape
        N = self.filters_n
        if S % N != 0:
            raise ValueError("samples_n (%d) is not a multiple of filters_n (%d)" % (S, N))
        K = S // N
        X = torch.empty((B, K + 1, N, C), dtype=x.dtype, device=x.device)
        t = torch.empty((B, K + 1, 1, C), dtype=x.dtype, device=x.device)
        thr = torch.empty_like(X)
        noisy = torch.empty_like(X) if noise_seed is not None else None
        dbn = torch.empty_like(X) if db_norm else None
        flags = (1 if noisy is not None else 0) | (2 if dbn is not None else 0)
        with _host.on_device(x.device):
            _lib.check(self._lib.ac_encode_fused_ex(
                self.mdct._plan(x.device), self.psy._plan(x.device), _host.ptr(x), _host.ptr(X), _host.ptr(t), _host.ptr(thr),
                float(drown), flags, _host.ptr(noisy) if noisy is not None else None,
                _host.ptr(dbn) if dbn is not None else None, (int(noise_seed) if noise_seed is not None else 0) & (2 ** 64 - 1),
                B, K, C, _host.stream_ptr(x.device)))
        return X, t, thr, noisy, dbn

    def _check_io(self, t, name, shape, dtype, device):
        """Caller-owned tensors go to the kernels as raw pointers: shape, dtype, device and contiguity must be exact."""
        if not isinstance(t, torch.Tensor):
            raise TypeError("%s must be a torch.Tensor, got %s" % (name, type(t).__name__))
        if not t.is_cuda:
            raise RuntimeError("%s lives on %s: this package only runs on ROCm device tensors" % (name, t.device))
        if device is not None and t.device != device:
            raise ValueError("%s lives on %s but the input lives on %s" % (name, t.device, device))
        if t.dtype != dtype:
            raise ValueError("%s has dtype %s, expected %s" % (name, t.dtype, dtype))
        if shape is not None and tuple(t.shape) != tuple(shape):
            raise ValueError("%s must have shape %s, got %s" % (name, tuple(shape), tuple(t.shape)))
        if not t.is_contiguous():
            raise ValueError("%s must be contiguous (got strides %s for shape %s)" % (name, t.stride(), tuple(t.shape)))
        if t.data_ptr() % 16 != 0:
            raise ValueError("%s starts at an address that is not 16-byte aligned (a view into the middle of an "
                             "allocation?)" % name)
        if t.requires_grad and torch.is_grad_enabled():
            raise ValueError("%s requires a gradient: results written into caller-owned tensors carry none -- use "
                             "encode() / decode(), which are differentiable" % name)
        return t

    def encode_into(self, x, X, t, thr, drown=0.0):
        """Same as :meth:`encode` into caller-owned output tensors (no allocation in the timed path).  Every tensor
        must be a contiguous device tensor of exactly the shape :meth:`encode` would return; ``ValueError`` otherwise."""
        if not isinstance(x, torch.Tensor) or x.dim() != 3:
            raise ValueError("x must be a [batches_n, samples_n, channels_n] tensor")
        pcm16 = x.dtype == torch.int16
        if pcm16 and self.compute_dtype != torch.float32:
            raise ValueError("16-bit PCM input needs compute_dtype float32")
        self._check_io(x, "x", None, torch.int16 if pcm16 else self.compute_dtype, None)
        B, S, C = x.shape
        N = self.filters_n
        if S % N != 0:
            raise ValueError("samples_n (%d) is not a multiple of filters_n (%d)" % (S, N))
        K = S // N
        self._check_io(X, "X", (B, K + 1, N, C), self.compute_dtype, x.device)
        self._check_io(t, "t", (B, K + 1, 1, C), self.compute_dtype, x.device)
        self._check_io(thr, "thr", (B, K + 1, N, C), self.compute_dtype, x.device)
        if self.compute_dtype != torch.float32:
            # float64: the three typed entry points in sequence; bfloat16: the fused wave-level kernel where it applies
            with _host.on_device(x.device):
                _lib.check(self._lib.ac_encode_fused_typed(
                    self.mdct._plan(x.device), self.psy._plan(x.device), _host.ptr(x), _host.ptr(X), _host.ptr(t),
                    _host.ptr(thr), float(drown), self.mdct._dtype_id, B, K, C, _host.stream_ptr(x.device)))
            return
        fn = self._lib.ac_encode_fused_pcm16 if pcm16 else self._lib.ac_encode_fused
        with _host.on_device(x.device):
            _lib.check(fn(self.mdct._plan(x.device), self.psy._plan(x.device), _host.ptr(x), _host.ptr(X), _host.ptr(t),
                          _host.ptr(thr), float(drown), B, K, C, _host.stream_ptr(x.device)))

    def workspace(self, batches_n, blocks_n, channels_n, **kwargs):
        """Caller-owned ``x, X, t, thr, xh`` tensors for batches of this shape, placed for the MI355X's HBM (see
        :class:`audiocodec_amd.workspace.Workspace`: on plain allocations the two kernels run up to 15 % slower when the
        tensors they stream side by side happen to share a class of VRAM stretches).  Use with :meth:`encode_into` /
        :meth:`decode_into`."""
        from .workspace import Workspace
        return Workspace(self, batches_n, blocks_n, channels_n, **kwargs)

    def stream(self, batches_n, channels_n, device=None):
        """A :class:`StreamingMDCT` over this codec's filter bank and masking model (chunked ``encode`` / ``decode`` with
        device-resident overlap state)."""
        return StreamingMDCT(self.mdct, batches_n, channels_n, device=device, psy=self.psy)

    def decode(self, X, pcm16=False):
        """X [B, K', N, C] -> x [B, (K'+1)*N, C]; ``pcm16=True`` returns ``torch.int16`` PCM
        (clamp(round(32768 x)) applied inside the kernel's stores)."""
        if not pcm16:
            return self.mdct.inverse_transform(X)
        X = _host.check_device_tensor(X, "X", self.compute_dtype, 4)
        B, Kp, N, C = X.shape
        if N != self.filters_n:
            raise ValueError("axis 2 of X (%d) != filters_n (%d)" % (N, self.filters_n))
        x = torch.empty((B, (Kp + 1) * N, C), dtype=torch.int16, device=X.device)
        self.decode_into(X, x)
        return x

    def decode_into(self, X, x):
        """:meth:`decode` into a caller-owned PCM tensor ``x [B, (K'+1)*N, C]`` (``torch.int16`` selects 16-bit PCM);
        same exactness rules as :meth:`encode_into`."""
        if not isinstance(X, torch.Tensor) or X.dim() != 4:
            raise ValueError("X must be a [batches_n, blocks_n, filters_n, channels_n] tensor")
        self._check_io(X, "X", None, self.compute_dtype, None)
        B, Kp, N, C = X.shape
        if N != self.filters_n:
            raise ValueError("axis 2 of X (%d) != filters_n (%d)" % (N, self.filters_n))
        pcm16 = isinstance(x, torch.Tensor) and x.dtype == torch.int16
        if pcm16 and self.compute_dtype != torch.float32:
            raise ValueError("16-bit PCM output needs compute_dtype float32")
        self._check_io(x, "x", (B, (Kp + 1) * N, C), torch.int16 if pcm16 else self.compute_dtype, X.device)
        if self.compute_dtype != torch.float32:
            with _host.on_device(X.device):
                _lib.check(self._lib.ac_mdct_inverse_typed(self.mdct._plan(X.device), _host.ptr(X), _host.ptr(x),
                                                           self.mdct._dtype_id, B, Kp, C, _host.stream_ptr(X.device)))
            return
        fn = self._lib.ac_mdct_inverse_pcm16 if pcm16 else self._lib.ac_mdct_inverse
        with _host.on_device(X.device):
            _lib.check(fn(self.mdct._plan(X.device), _host.ptr(X), _host.ptr(x), B, Kp, C, _host.stream_ptr(X.device)))


class StreamingMDCT:
    """Chunked analysis / synthesis with device-resident overlap state (BASELINE config 5).

    Feeding consecutive chunks of ``k`` blocks gives, frame for frame, the one-shot ``transform`` /
    ``inverse_transform`` result: analysis frame ``i`` of a chunk pairs block ``i`` with block ``i-1``
    (the stored last block of the previous chunk for ``i = 0``); synthesis block ``i`` overlap-adds
    frame ``i`` with the stored aliased half of the previous frame.  With a masking model (``psy``)
    :meth:`encode_chunk` also returns tonality and threshold of the chunk's frames -- bit for bit what the one-shot
    ``AudioCodec.encode`` gives for those frames.
    """

    def __init__(self, mdct: MDCTransformer, batches_n, channels_n, device=None, psy: PsychoacousticModel = None):
        if mdct.compute_dtype not in (torch.float32, torch.float64, torch.bfloat16):
            raise NotImplementedError("streaming overlap-add serves compute_dtype float32 and float64 (every size) and bfloat16 "
                                      "(the wave-level kernels: filters_n 1024 / 2048, mono / stereo); got %s" % mdct.compute_dtype)
        if psy is not None:
            if psy.compute_dtype != mdct.compute_dtype:
                raise ValueError("psy.compute_dtype (%s) != mdct.compute_dtype (%s)" % (psy.compute_dtype, mdct.compute_dtype))
            if psy.filter_bands_n != mdct.filters_n:
                raise ValueError("psy.filter_bands_n (%d) != mdct.filters_n (%d)" % (psy.filter_bands_n, mdct.filters_n))
        self.mdct, self.psy = mdct, psy
        self.B, self.C = int(batches_n), int(channels_n)
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self._lib = _lib.load()
        handle = ctypes.c_void_p()
        with _host.on_device(self.device):
            _lib.check(self._lib.ac_stream_create(mdct._plan(self.device), self.B, self.C, ctypes.byref(handle)))
        self._handle = handle
        self._graphs = {}       # run(..., graph=True): captured calls by input buffers and arguments

    def close(self):
        self._graphs = {}       # (the graphs hold launches that address the stream's state)
        if self._handle is not None:
            self._lib.ac_stream_destroy(self._handle)
            self._handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        with _host.on_device(self.device):
            _lib.check(self._lib.ac_stream_reset(self._handle, _host.stream_ptr(self.device)))

    def _check_chunk(self, x, name, ndim):
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            raise ValueError("%s requires a gradient: the streaming calls keep state on the device and are not "
                             "differentiable -- use MDCTransformer.transform / inverse_transform" % name)
        x = _host.check_device_tensor(x, name, self.mdct.compute_dtype, ndim)
        if x.device != self.device:
            raise ValueError("%s lives on %s but the stream's state lives on %s" % (name, x.device, self.device))
        return x

    def _out(self, out, name, shape, like):
        if out is None:
            return torch.empty(shape, dtype=like.dtype, device=like.device)
        if not isinstance(out, torch.Tensor) or tuple(out.shape) != tuple(shape) or out.dtype != like.dtype \
                or out.device != like.device or not out.is_contiguous() or out.data_ptr() % 16 != 0:
            raise ValueError("%s must be a contiguous, 16-byte aligned %s tensor of shape %s on %s"
                             % (name, like.dtype, tuple(shape), like.device))
        return out

    def _stream(self, stream):
        return _host.stream_ptr(self.device) if stream is None else ctypes.c_void_p(stream.cuda_stream)

    def transform_chunk(self, x_chunk, out=None, stream=None):
        """x_chunk [B, k*N, C] -> X [B, k, N, C].  ``stream``: a ``torch.cuda.Stream`` to enqueue on (default: the
        current stream)."""
        x = self._check_chunk(x_chunk, "x_chunk", 3)
        B, S, C = x.shape
        N = self.mdct.filters_n
        if (B, C) != (self.B, self.C) or S % N != 0:
            raise ValueError("x_chunk must be [%d, k*%d, %d], got %s" % (self.B, N, self.C, tuple(x.shape)))
        k = S // N
        X = self._out(out, "out", (B, k, N, C), x)
        with _host.on_device(x.device):
            _lib.check(self._lib.ac_stream_encode_typed(self._handle, None, _host.ptr(x), _host.ptr(X), None, None, 0.0,
                                                        self.mdct._dtype_id, k, self._stream(stream)))
        return X

    def encode_chunk(self, x_chunk, drown=0.0, out=None, stream=None):
        """x_chunk [B, k*N, C] -> (X [B, k, N, C], tonality [B, k, 1, C], threshold [B, k, N, C]) of the chunk's frames
        (``ac_stream_encode``: MDCT, tonality and masking threshold in one launch where the wave-level kernels apply).
        ``out``: optional tuple of three tensors to write into."""
        if self.psy is None:
            raise ValueError("this stream was created without a masking model (psy=...)")
        x = self._check_chunk(x_chunk, "x_chunk", 3)
        B, S, C = x.shape
        N = self.mdct.filters_n
        if (B, C) != (self.B, self.C) or S % N != 0:
            raise ValueError("x_chunk must be [%d, k*%d, %d], got %s" % (self.B, N, self.C, tuple(x.shape)))
        k = S // N
        o = out if out is not None else (None, None, None)
        X = self._out(o[0], "out[0]", (B, k, N, C), x)
        t = self._out(o[1], "out[1]", (B, k, 1, C), x)
        thr = self._out(o[2], "out[2]", (B, k, N, C), x)
        with _host.on_device(x.device):
            _lib.check(self._lib.ac_stream_encode_typed(self._handle, self.psy._plan(x.device), _host.ptr(x), _host.ptr(X),
                                                        _host.ptr(t), _host.ptr(thr), float(drown), self.mdct._dtype_id, k,
                                                        self._stream(stream)))
        return X, t, thr

    def run(self, x, blocks_per_chunk, masking=True, synthesis=True, drown=0.0, graph=False):
        """A long device-resident signal ``x [1, K*N, C]`` (or a list of chunk tensors ``[B, k*N, C]``) through the stream
        in chunks of ``blocks_per_chunk`` blocks with one library call (``ac_stream_run``: launches issued from C; with
        ``synthesis`` and chunks small enough to be latency-bound, the analysis of chunk i + 1 and the synthesis of chunk i
        share one launch).  Returns ``(X, t, thr, xhat)`` -- tensors ``[1, K, ...]`` for
        a tensor input, lists of per-chunk tensors for a list input; ``t`` / ``thr`` are None without ``masking``,
        ``xhat`` is None without ``synthesis``.  The last chunk of a tensor input may be shorter.

        ``graph=True``: the launches of the call are captured into a HIP graph the first time these input buffers (same
        addresses, shapes and arguments) are seen, and replayed from then on -- the gaps between the dependent launches
        go (one stereo clip in chunks of 256 blocks: 7.9 us per chunk against 11.9).  A replay reads the CURRENT contents
        of the same input buffers and overwrites the output tensors of the first call, which are returned again."""
        _host.require_float32(self.mdct.compute_dtype, "StreamingMDCT.run (use the chunk calls for bfloat16 / float64 streams)")
        if graph:
            return self._run_graph(x, blocks_per_chunk, masking, synthesis, drown)
        k, N = int(blocks_per_chunk), self.mdct.filters_n
        if masking and self.psy is None:
            raise ValueError("this stream was created without a masking model (psy=...)")
        if k < 1:
            raise ValueError("blocks_per_chunk must be positive")
        as_list = isinstance(x, (list, tuple))
        if as_list:
            xs = [self._check_chunk(c, "x[%d]" % i, 3) for i, c in enumerate(x)]
            for c in xs:
                if tuple(c.shape) != (self.B, k * N, self.C):
                    raise ValueError("every chunk must be [%d, %d, %d], got %s" % (self.B, k * N, self.C, tuple(c.shape)))
            groups = [(xs, k)]
            like = xs[0] if xs else None
        else:
            xt = self._check_chunk(x, "x", 3)
            B, S, C = xt.shape
            if B != 1 or self.B != 1 or C != self.C or S % N != 0:
                raise ValueError("a tensor input must be [1, K*%d, %d] on a stream of one clip (pass a list of chunks "
                                 "otherwise), got %s" % (N, self.C, tuple(xt.shape)))
            K = S // N
            full, rest = K // k, K % k
            like = xt
            if (k * N * C * 4) % 16 != 0 and K > k:
                # chunk boundaries inside one tensor would not be 16-byte aligned (filters_n * channels_n odd multiples of
                # 2): per-chunk tensors through the list form, concatenated afterwards
                parts = [xt[:, i * k * N:(i + 1) * k * N].clone() for i in range(full)]
                o = self.run(parts, k, masking=masking, synthesis=synthesis, drown=drown) if full else ([], [], [], [])
                if rest:
                    r = self.run([xt[:, full * k * N:].clone()], rest, masking=masking, synthesis=synthesis, drown=drown)
                    o = tuple((a or []) + (b or []) if (a is not None or b is not None) else None for a, b in zip(o, r))
                cat = lambda ts: torch.cat(ts, dim=1) if ts else None   # noqa: E731
                return cat(o[0]), cat(o[1]) if masking else None, cat(o[2]) if masking else None, \
                    cat(o[3]) if synthesis else None
            Xall = torch.empty((1, K, N, C), dtype=xt.dtype, device=xt.device)
            tall = torch.empty((1, K, 1, C), dtype=xt.dtype, device=xt.device) if masking else None
            thrall = torch.empty_like(Xall) if masking else None
            xhall = torch.empty_like(xt) if synthesis else None
            groups = []
            if full:
                groups.append(([xt[:, i * k * N:(i + 1) * k * N] for i in range(full)], k))
            if rest:
                groups.append(([xt[:, full * k * N:]], rest))
        outs = ([], [], [], [])
        pos = 0
        for chunks, kk in groups:
            n = len(chunks)
            if n == 0:
                continue
            if as_list:
                Xc = [torch.empty((self.B, kk, N, self.C), dtype=like.dtype, device=like.device) for _ in range(n)]
                tc = [torch.empty((self.B, kk, 1, self.C), dtype=like.dtype, device=like.device) for _ in range(n)] if masking else None
                thc = [torch.empty_like(v) for v in Xc] if masking else None
                xhc = [torch.empty_like(c) for c in chunks] if synthesis else None
            else:
                Xc = [Xall[:, pos + i * kk: pos + (i + 1) * kk] for i in range(n)]
                tc = [tall[:, pos + i * kk: pos + (i + 1) * kk] for i in range(n)] if masking else None
                thc = [thrall[:, pos + i * kk: pos + (i + 1) * kk] for i in range(n)] if masking else None
                xhc = [xhall[:, (pos + i * kk) * N: (pos + (i + 1) * kk) * N] for i in range(n)] if synthesis else None
            arr = lambda ts: (ctypes.c_void_p * n)(*[v.data_ptr() for v in ts]) if ts is not None else None   # noqa: E731
            with _host.on_device(self.device):
                _lib.check(self._lib.ac_stream_run(self._handle, self.psy._plan(self.device) if masking else None, n, kk,
                                                   arr(chunks), arr(Xc), arr(tc), arr(thc), arr(xhc), float(drown),
                                                   _host.stream_ptr(self.device)))
            for o, v in zip(outs, (Xc, tc, thc, xhc)):
                if v is not None:
                    o.extend(v)
            pos += n * kk
        if as_list:
            return outs[0], (outs[1] if masking else None), (outs[2] if masking else None), (outs[3] if synthesis else None)
        return Xall, tall, thrall, xhall

    def _run_graph(self, x, blocks_per_chunk, masking, synthesis, drown):
        chunks = list(x) if isinstance(x, (list, tuple)) else [x]
        for c in chunks:
            self._check_chunk(c, "x", 3)
        key = (tuple((c.data_ptr(), tuple(c.shape)) for c in chunks), isinstance(x, (list, tuple)), int(blocks_per_chunk),
               bool(masking), bool(synthesis), float(drown))
        hit = self._graphs.get(key)
        # a captured call addresses the stream's HOME state buffers; chunk calls (and reset) since the last run() may have
        # left the current state in the other buffer of the pair: move it home first (no device work when it is there)
        with _host.on_device(self.device):
            _lib.check(self._lib.ac_stream_settle(self._handle, _host.stream_ptr(self.device)))
        if hit is None:
            with _host.on_device(self.device):
                self.mdct._plan(self.device)                       # plans are built outside the capture
                if masking:
                    if self.psy is None:
                        raise ValueError("this stream was created without a masking model (psy=...)")
                    self.psy._plan(self.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    outs = self.run(x, blocks_per_chunk, masking=masking, synthesis=synthesis, drown=drown)
            while len(self._graphs) >= 8:                         # a graph keeps its outputs alive: a handful at most
                self._graphs.pop(next(iter(self._graphs)))
            hit = self._graphs[key] = (g, outs, chunks)            # (the inputs stay alive with their graph)
        hit[0].replay()
        return hit[1]

    def inverse_chunk(self, X_chunk, out=None, stream=None):
        """X_chunk [B, k, N, C] -> x [B, k*N, C]."""
        X = self._check_chunk(X_chunk, "X_chunk", 4)
        B, k, N, C = X.shape
        if (B, C, N) != (self.B, self.C, self.mdct.filters_n):
            raise ValueError("X_chunk must be [%d, k, %d, %d], got %s" % (self.B, self.mdct.filters_n, self.C,
                                                                         tuple(X.shape)))
        x = self._out(out, "out", (B, k * N, C), X)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_stream_inverse_typed(self._handle, _host.ptr(X), _host.ptr(x), self.mdct._dtype_id, k,
                                                         self._stream(stream)))
        return x
