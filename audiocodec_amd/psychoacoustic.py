"""Psychoacoustic masking model on MI355X.

Drop-in for ``audiocodec.psychoacoustic.PsychoacousticModel`` (reference
``audiocodec/psychoacoustic.py:13-339``): same constructor keywords, methods, layouts and
attributes, on ``torch`` ROCm tensors.  Constants are pre-computed in float64 by the native
library (host side), the per-frame model runs in hand-written HIP kernels.
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _host, _lib, placement


class _TonalityFn(torch.autograd.Function):
    """Differentiable ``tonality`` (the reference is differentiated by TensorFlow inside a training graph,
    ``psychoacoustic.py:311``); the backward pass is the explicit adjoint kernel ``ac_tonality_backward``."""

    @staticmethod
    def forward(ctx, X, model):
        ctx.model = model
        ctx.save_for_backward(X)
        return model._tonality(X)

    @staticmethod
    def backward(ctx, gt):
        (X,) = ctx.saved_tensors
        return ctx.model._tonality_backward(X, gt.contiguous()), None


class _ThresholdFn(torch.autograd.Function):
    """Differentiable ``global_masking_threshold`` w.r.t. the amplitudes and the tonality (``ac_mask_threshold_backward``)."""

    @staticmethod
    def forward(ctx, X, t, model, drown):
        ctx.model, ctx.drown = model, drown
        ctx.save_for_backward(X, t)
        return model._threshold(X, t, drown)

    @staticmethod
    def backward(ctx, gthr):
        X, t = ctx.saved_tensors
        gX, gt = ctx.model._threshold_backward(X, t, ctx.drown, gthr.contiguous())
        return gX, gt, None, None


class _AddNoiseFn(torch.autograd.Function):
    """Differentiable ``add_noise`` (a plain differentiable op chain in the reference, ``psychoacoustic.py:150-167``):
    out = X + thr * n with n ~ Normal(0, 1/6) fixed by the seed, so d out / d X = 1 and d out / d thr = n; the second is
    the same kernel run on (0, grad_out) under the same seed."""

    @staticmethod
    def forward(ctx, X, thr, model, seed):
        ctx.model, ctx.seed = model, seed
        return model._add_noise(X, thr, seed)

    @staticmethod
    def backward(ctx, g):
        g = g.contiguous()
        gX = g if ctx.needs_input_grad[0] else None
        gthr = ctx.model._add_noise(None, g, ctx.seed) if ctx.needs_input_grad[1] else None
        return gX, gthr, None, None


class _DbFn(torch.autograd.Function):
    """Differentiable ``amplitude_to_dB`` / ``amplitude_to_dB_norm`` (``psychoacoustic.py:71-100``): the adjoint kernel
    ``ac_amplitude_to_db_backward`` (zero inside the clamp at ``_INTENSITY_EPS``)."""

    @staticmethod
    def forward(ctx, a, model, norm):
        ctx.model, ctx.norm = model, norm
        ctx.save_for_backward(a)
        return model._elementwise_db(a, norm)

    @staticmethod
    def backward(ctx, g):
        (a,) = ctx.saved_tensors
        return ctx.model._db_backward(a, g.contiguous(), ctx.norm), None, None


class PsychoacousticModel:
    SPREADING = {"f32": 0, "bf16_mfma": 1, "bf16x2_mfma": 2}

    def __init__(self, sample_rate, filter_bands_n=1024, bark_bands_n=64, alpha=0.6,
                 compute_dtype=torch.float32, precompute_dtype=torch.float64, spreading=None):
        """Same signature as the reference (``psychoacoustic.py:14-15``) plus one extension:

        :param spreading: form of the band x band product with the spreading matrix (``psychoacoustic.py:205-207``) in
                          the wave-level kernels (include/audiocodec_amd.h, AC_SPREAD_*): ``"f32"`` vector-ALU
                          multiply-adds, ``"bf16_mfma"`` bfloat16 operands on the matrix cores (thresholds within 5e-3),
                          ``"bf16x2_mfma"`` split-bfloat16 operands on the matrix cores (thresholds within the 1e-4
                          parity bar).  None = the library's default: ``"bf16x2_mfma"`` where the wave-level kernels
                          serve the model (filter_bands_n 1024 / 2048, 64 Bark bands), ``"f32"`` otherwise (the
                          AC_SPREAD tuning hook overrides).  Asking for a matrix-core form elsewhere is an error

        :raises TypeError: when compute_dtype is not float64, float32 or bfloat16 (``:42-43``).  float32 runs the
                           wave-level kernels; float64 runs float64 kernels on float64 constants (the on-device
                           cross-check); bfloat16 means bfloat16 tensors with float32 arithmetic inside.  Autograd is
                           float32 only
        """
        self.alpha = alpha
        self.sample_rate = sample_rate
        self.bark_bands_n = int(bark_bands_n)
        self.filter_bands_n = int(filter_bands_n)
        compute_dtype = _host.as_torch_dtype(compute_dtype)
        if compute_dtype not in (torch.float64, torch.float32, torch.bfloat16):
            raise TypeError("compute_dtype of PsychoacousticModel should be float64, float32 or bfloat16")
        self.compute_dtype = compute_dtype
        self._dtype_id = _host.require_hip_compute_dtype(compute_dtype, "PsychoacousticModel")
        self.precompute_dtype, self._pre_id = _host.precompute_id(precompute_dtype, "PsychoacousticModel")
        self._lib = _lib.load()

        N, M = self.filter_bands_n, self.bark_bands_n
        # constants computed in the precompute dtype (``:61-69``), then held in the compute dtype, as the reference holds
        # them (float64: unrounded; bfloat16: float32 here, the kernels' arithmetic type)
        np_t = np.float64 if compute_dtype == torch.float64 else np.float32
        W = np.empty((N, M), dtype=np.float64)
        W_inv = np.empty((M, N), dtype=np.float64)
        S = np.empty((M, M), dtype=np.float64)
        quiet = np.empty((M,), dtype=np.float64)
        scalars = np.empty((4,), dtype=np.float64)
        fp = ctypes.POINTER(ctypes.c_double)
        _lib.check(self._lib.ac_psy_tables_host_pre(
            N, M, float(sample_rate), float(alpha), self._pre_id, W.ctypes.data_as(fp), W_inv.ctypes.data_as(fp),
            S.ctypes.data_as(fp), quiet.ctypes.data_as(fp), scalars.ctypes.data_as(fp)))
        W, W_inv, S, quiet = (v.astype(np_t) for v in (W, W_inv, S, quiet))
        self._dB_MAX = torch.tensor(120.0, dtype=compute_dtype)                 # :52
        self._INTENSITY_EPS = torch.tensor(1e-14, dtype=compute_dtype)          # :56
        self._dB_MIN = torch.tensor(scalars[3], dtype=compute_dtype)            # :58  (= -20 dB)
        self.max_frequency = torch.tensor(scalars[0], dtype=self.precompute_dtype)      # :61
        self.max_bark = torch.tensor(scalars[1], dtype=self.precompute_dtype)           # :62
        self.bark_band_width = torch.tensor(scalars[2], dtype=self.precompute_dtype)    # :63
        self.W = torch.from_numpy(W)                                            # :66
        self.W_inv = torch.from_numpy(W_inv)                                    # :67
        self.quiet_threshold_intensity = torch.from_numpy(quiet).reshape(1, 1, M, 1)   # :68
        self.spreading_matrix = torch.from_numpy(S)                             # :69

        sr, al, lib, pre = float(sample_rate), float(alpha), self._lib, self._pre_id
        if spreading is not None and spreading not in self.SPREADING:
            raise ValueError("spreading must be one of %s" % sorted(self.SPREADING))
        self.spreading = spreading
        mode = -1 if spreading is None else self.SPREADING[spreading]   # -1: the library's default for the plan
        create = lambda dev, out: lib.ac_psy_plan_create_pre(N, M, sr, al, dev, mode, pre, out)   # noqa: E731
        self._plans = _host.PlanCache(self, create, lib.ac_psy_plan_destroy)

    def _plan(self, device):
        return self._plans.get(device)

    def is_fast(self, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return bool(self._lib.ac_psy_plan_is_fast(self._plans.get(dev)))

    def tier(self, device=None):
        """2: wave-level kernels fused into the encode; 1: wave-level kernels for general band layouts; 0: generic kernels
        (``ac_psy_plan_tier``)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return int(self._lib.ac_psy_plan_tier(self._plans.get(dev)))

    def plan_spreading(self, device=None):
        """The form of the spreading product the plan on ``device`` runs (a key of ``SPREADING``)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        mode = self._lib.ac_psy_plan_spreading(self._plans.get(dev))
        return {v: k for k, v in self.SPREADING.items()}[mode]

    # ---- element-wise utilities -----------------------------------------------------------------
    def _elementwise_db(self, mdct_amplitude, norm):
        a = mdct_amplitude
        if not isinstance(a, torch.Tensor):
            raise TypeError("mdct_amplitude must be a torch.Tensor")
        if a.dtype != self.compute_dtype:
            raise ValueError("mdct_amplitude has dtype %s but compute_dtype is %s" % (a.dtype, self.compute_dtype))
        if not a.is_cuda:
            raise RuntimeError("mdct_amplitude lives on %s: no CPU fallback" % a.device)
        a = a.contiguous()
        out = torch.empty_like(a)
        with _host.on_device(a.device):
            _lib.check(self._lib.ac_amplitude_to_db_typed(_host.ptr(a), _host.ptr(out), a.numel(), int(norm),
                                                          self._dtype_id, _host.stream_ptr(a.device)))
        return out

    def _db_backward(self, a, g, norm):
        # the kernel walks three flat arrays in the logical (row-major) order: one contiguous copy of each input, held until
        # the launch has been enqueued, and a contiguous result (a permuted view would otherwise get a scrambled gradient)
        a_c, g_c = a.contiguous(), g.contiguous()
        ga = torch.empty(a.shape, dtype=a.dtype, device=a.device)
        with _host.on_device(a.device):
            _lib.check(self._lib.ac_amplitude_to_db_backward(_host.ptr(a_c), _host.ptr(g_c), _host.ptr(ga),
                                                             a_c.numel(), int(norm), _host.stream_ptr(a.device)))
        return ga

    def _db(self, mdct_amplitude, norm):
        a = mdct_amplitude
        if isinstance(a, torch.Tensor) and a.requires_grad and torch.is_grad_enabled():
            _host.require_float32(self.compute_dtype, "the backward pass of amplitude_to_dB")
            return _DbFn.apply(a, self, norm)
        return self._elementwise_db(a, norm)

    def amplitude_to_dB(self, mdct_amplitude):
        """``amplitude_to_dB`` (``psychoacoustic.py:71-85``): [-1,1] amplitude -> dB in [_dB_MIN, _dB_MAX].
        Differentiable (float32)."""
        return self._db(mdct_amplitude, False)

    def amplitude_to_dB_norm(self, mdct_amplitude):
        """``amplitude_to_dB_norm`` (``psychoacoustic.py:87-100``): dB scale normalised to [0, 1].  Differentiable
        (float32)."""
        return self._db(mdct_amplitude, True)

    # ---- per-frame model -------------------------------------------------------------------------
    def _check_spectrum(self, X, name="mdct_amplitudes"):
        X = _host.check_device_tensor(X, name, self.compute_dtype, 4)
        if X.shape[2] != self.filter_bands_n:
            raise ValueError("axis 2 of %s (%d) != filter_bands_n (%d)" % (name, X.shape[2], self.filter_bands_n))
        return X

    def tonality(self, mdct_amplitudes):
        """``tonality`` (``psychoacoustic.py:102-120``): [B, K, N, C] -> [B, K, 1, C] in [0, 1]."""
        if isinstance(mdct_amplitudes, torch.Tensor) and mdct_amplitudes.requires_grad and torch.is_grad_enabled():
            return _TonalityFn.apply(mdct_amplitudes, self)   # (every compute_dtype: ac_tonality_backward_typed)
        return self._tonality(mdct_amplitudes)

    def _tonality(self, mdct_amplitudes):
        X = self._check_spectrum(mdct_amplitudes)
        B, F, N, C = X.shape
        t = torch.empty((B, F, 1, C), dtype=X.dtype, device=X.device)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_tonality_typed(self._plans.get(X.device), _host.ptr(X), _host.ptr(t),
                                                   self._dtype_id, B, F, C, _host.stream_ptr(X.device)))
        return t

    def _tonality_backward(self, X, gt):
        X = self._check_spectrum(X)
        B, F, N, C = X.shape
        gX = torch.empty_like(X)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_tonality_backward_typed(self._plans.get(X.device), _host.ptr(X), _host.ptr(gt), _host.ptr(gX),
                                                            self._dtype_id, B, F, C, _host.stream_ptr(X.device)))
        return gX

    def global_masking_threshold(self, mdct_amplitudes, tonality_per_block, drown=0.0):
        """``global_masking_threshold`` (``psychoacoustic.py:122-148``): -> [B, K, N, C], strictly positive."""
        needs_grad = any(isinstance(v, torch.Tensor) and v.requires_grad for v in (mdct_amplitudes, tonality_per_block))
        if needs_grad and torch.is_grad_enabled():
            return _ThresholdFn.apply(mdct_amplitudes, tonality_per_block, self, float(drown))
        return self._threshold(mdct_amplitudes, tonality_per_block, drown)

    def _threshold_backward(self, X, t, drown, gthr):
        X = self._check_spectrum(X)
        B, F, N, C = X.shape
        t = t.contiguous()
        gX = torch.empty_like(X)
        gt = torch.empty_like(t)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_mask_threshold_backward_typed(self._plans.get(X.device), _host.ptr(X), _host.ptr(t),
                                                                  float(drown), _host.ptr(gthr), _host.ptr(gX), _host.ptr(gt),
                                                                  self._dtype_id, B, F, C, _host.stream_ptr(X.device)))
        return gX, gt

    def _threshold(self, mdct_amplitudes, tonality_per_block, drown=0.0):
        X = self._check_spectrum(mdct_amplitudes)
        B, F, N, C = X.shape
        t = _host.check_device_tensor(tonality_per_block, "tonality_per_block", self.compute_dtype, 4)
        if tuple(t.shape) != (B, F, 1, C):
            raise ValueError("tonality_per_block must have shape %s, got %s" % ((B, F, 1, C), tuple(t.shape)))
        if t.device != X.device:
            raise ValueError("mdct_amplitudes and tonality_per_block live on different devices")
        thr = placement.empty(placement.REGION_OTHER, tuple(X.shape), X.dtype, X.device)   # (see placement.py)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_mask_threshold_typed(self._plans.get(X.device), _host.ptr(X), _host.ptr(t),
                                                         float(drown), _host.ptr(thr), self._dtype_id, B, F, C,
                                                         _host.stream_ptr(X.device)))
        return thr

    def add_noise(self, mdct_amplitudes, masking_threshold, seed=None):
        """``add_noise`` (``psychoacoustic.py:150-167``): X + thr * Normal(0, 1/6).

        The generator is counter-based (seed, element index); the stream differs from TensorFlow's, so
        parity is statistical (mean 0, sigma = thr / 6).  ``seed=None`` draws one from torch's generator.
        Differentiable with respect to both inputs (float32): the noise is a constant of the seed.
        """
        X = self._check_spectrum(mdct_amplitudes)
        thr = _host.check_device_tensor(masking_threshold, "masking_threshold", self.compute_dtype, 4)
        if thr.shape != X.shape or thr.device != X.device:
            raise ValueError("masking_threshold must match mdct_amplitudes in shape and device")
        if seed is None:
            seed = int(torch.randint(0, 2 ** 62, (1,)).item())
        seed = int(seed) & (2 ** 64 - 1)
        if (X.requires_grad or thr.requires_grad) and torch.is_grad_enabled():
            _host.require_float32(self.compute_dtype, "the backward pass of add_noise")
            return _AddNoiseFn.apply(X, thr, self, seed)
        return self._add_noise(X, thr, seed)

    def _add_noise(self, X, thr, seed):
        """X may be None (zeros): thr * Normal(0, 1/6) under the same seed."""
        out = torch.empty_like(thr)
        with _host.on_device(thr.device):
            _lib.check(self._lib.ac_add_noise_typed(_host.ptr(X) if X is not None else None, _host.ptr(thr), _host.ptr(out),
                                                    thr.numel(), seed, self._dtype_id, _host.stream_ptr(thr.device)))
        return out

    # ---- Bark scale (host precompute helpers, psychoacoustic.py:333-339) ------------------------------
    def freq2bark(self, frequencies):
        """Empirical Bark scale (``:333-335``)."""
        return 6.0 * torch.asinh(torch.as_tensor(frequencies, dtype=torch.float64) / 600.0)

    def bark2freq(self, bark_band):
        """Empirical Bark scale (``:337-339``)."""
        return 600.0 * torch.sinh(torch.as_tensor(bark_band, dtype=torch.float64) / 6.0)
