"""MDCT analysis / synthesis filter bank on MI355X.

Drop-in for ``audiocodec.mdctransformer.MDCTransformer`` (reference
``audiocodec/mdctransformer.py:12-368``): same constructor keywords, method names, tensor layouts
and attribute names; tensors are ``torch`` tensors on a ROCm device instead of ``tf.Tensor``.
The arithmetic is done by hand-written HIP kernels behind the C ABI of ``libaudiocodec_amd.so``;
PyTorch only owns the device memory and the stream.
"""

from __future__ import annotations

import ctypes

import numpy as np
import torch

from . import _host, _lib, placement


class _TransformFn(torch.autograd.Function):
    """Differentiable ``transform`` (the reference is differentiated by TensorFlow when it sits in a training graph).  The
    analysis bank T is linear and the DCT-IV symmetric, so T^T is the synthesis kernel run on the TRANSPOSED fold
    coefficients (``ac_mdct_plan_adjoint``): ``T^T g = inverse_transform_adjoint(g)[:, N:-N] / (4 N)``, one launch.  For
    Princen-Bradley windows computed in float64 the adjoint plan equals the plan (F^-1 = F^T); for the rectangular window
    (``mdctransformer.py:209-229``: 2x2 blocks [[1, 1], [1, 0]]) and float32-precomputed constants it does not."""

    @staticmethod
    def forward(ctx, x, mdct):
        ctx.mdct = mdct
        return mdct._transform(x)

    @staticmethod
    def backward(ctx, gX):
        m = ctx.mdct
        N = m.filters_n
        gx = m._inverse(gX.contiguous(), adjoint=True)[:, N:-N] / (4.0 * N)
        return gx, None


class _InverseFn(torch.autograd.Function):
    """Differentiable ``inverse_transform``: ``S^T g = 4 N * transform_adjoint(g)[:, 1:-1]`` (see ``_TransformFn``)."""

    @staticmethod
    def forward(ctx, X, mdct):
        ctx.mdct = mdct
        return mdct._inverse(X)

    @staticmethod
    def backward(ctx, gx):
        m = ctx.mdct
        gX = m._transform(gx.contiguous(), adjoint=True)[:, 1:-1] * (4.0 * m.filters_n)
        return gX, None


class MDCTransformer:
    def __init__(self, filters_n=1024, window_type="vorbis", compute_dtype=torch.float32,
                 precompute_dtype=torch.float64):
        """Same signature as the reference (``mdctransformer.py:13-14``).

        :param filters_n:        number of filter bands (must be even, ``:26``)
        :param window_type:      'sine', 'vorbis' (default); any other string or None selects the
                                 rectangular window (``:199-211``; the reference crashes on None)
        :param compute_dtype:    dtype of inputs and outputs: float32 (the wave-level kernels), float64 (everything in
                                 float64, constants included: the on-device float64 cross-check), bfloat16 or float16 (2-byte
                                 tensors, float32 arithmetic inside: the reference up-casts them inside its DCT-IV, ``:327-344``;
                                 float16 results beyond 65504 become infinity as a cast makes them).  ``transform`` and
                                 ``inverse_transform`` are differentiable in every one of them (the reference's op chain is,
                                 ``:31-35``); streaming: float32, float64, bfloat16
        :param precompute_dtype: arithmetic type the window / fold constants are computed in on the host before they are
                                 cast to float32 tables (``:14,31-35,58-59``): float64 (default) or float32 -- the latter
                                 rounds every constant and operation to float32 in the reference's order, the cancellation
                                 at ``:218-221`` included.  How close that is to the reference: its dense ``H`` / ``H_inv``
                                 from the reference's own source run with float32 precompute over numpy (a float32 LAPACK
                                 LU inverse of the full ``F`` where this library inverts 2x2 blocks in closed form) are met
                                 to 3e-7 / 5e-7 absolute, the reference's TensorFlow known-answer vector (which stems from
                                 such a revision) to 5e-8; parity with a TensorFlow float32 run is not demonstrated beyond
                                 that vector.  The fold blocks are then no rotations and the kernels carry all four
                                 coefficients per block
        """
        assert (filters_n % 2) == 0, "number of filters used in mdct transformation needs to be even"
        self.filters_n = int(filters_n)
        self.window_type = window_type
        self.compute_dtype = _host.as_torch_dtype(compute_dtype)
        self.precompute_dtype, self._pre_id = _host.precompute_id(precompute_dtype, "MDCTransformer")
        self._dtype_id = _host.require_hip_compute_dtype(self.compute_dtype, "MDCTransformer", filter_bank=True)
        self._window = _lib.window_id(window_type)
        self._lib = _lib.load()
        self._H = None
        self._H_inv = None
        n, w, pre, lib = self.filters_n, self._window, self._pre_id, self._lib
        self._plans = _host.PlanCache(self, lambda dev, out: lib.ac_mdct_plan_create_pre(n, w, pre, dev, out),
                                      lib.ac_mdct_plan_destroy)
        # the transposed bank, for the backward passes (built on first use)
        plans = self._plans
        self._adjoint_plans = _host.PlanCache(
            self, lambda dev, out: lib.ac_mdct_plan_adjoint(plans.get(torch.device("cuda", dev)), out), lib.ac_mdct_plan_destroy)

    # ---- dense polyphase matrices, for attribute parity only (mdctransformer.py:58-59) -------------
    def _dense(self):
        if self._H is None:
            n = self.filters_n
            H = np.empty((2, n, n), dtype=np.float32)
            Hi = np.empty((2, n, n), dtype=np.float32)
            fp = ctypes.POINTER(ctypes.c_float)
            _lib.check(self._lib.ac_mdct_dense_matrices_host_pre(n, self._window, self._pre_id, H.ctypes.data_as(fp),
                                                                 Hi.ctypes.data_as(fp)))
            self._H, self._H_inv = torch.from_numpy(H), torch.from_numpy(Hi)
        return self._H, self._H_inv

    @property
    def H(self):
        """[2, N, N] analysis polyphase matrix (2N non-zeros); built lazily, never used by the kernels."""
        return self._dense()[0]

    @property
    def H_inv(self):
        """[2, N, N] synthesis polyphase matrix; built lazily, never used by the kernels."""
        return self._dense()[1]

    def fold_coefficients(self):
        """The 8 x N/2 non-zeros of F and F^-1 in float64 (a1..a4, s1..s4; see include/audiocodec_amd.h)."""
        h = self.filters_n // 2
        coef = np.empty((8, h), dtype=np.float64)
        _lib.check(self._lib.ac_mdct_fold_coefficients_host_pre(
            self.filters_n, self._window, self._pre_id, coef.ctypes.data_as(ctypes.POINTER(ctypes.c_double))))
        return coef

    def is_fast(self, device=None):
        """True when this size runs the wave-level FFT kernels (else the generic O(N^2) kernels)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return bool(self._lib.ac_mdct_plan_is_fast(self._plans.get(dev)))

    def tier(self, channels_n=2, device=None):
        """Which kernels serve float32 tensors of ``channels_n`` channels: 3 the wave-level kernels, 2 a compile-time instance
        of the LDS-FFT tier, 1 the tier's run-time forms, 0 the O(N^2) kernels (``ac_mdct_plan_tier``)."""
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        return int(self._lib.ac_mdct_plan_tier(self._plans.get(dev), int(channels_n)))

    # ---- analysis ------------------------------------------------------------------------------------
    def transform(self, x):
        """MDCT analysis filter bank (reference ``transform``, ``mdctransformer.py:62-125``).

        :param x: signal in -1..1, ``[batches_n, samples_n, channels_n]``, ``samples_n`` a multiple of
                  ``filters_n`` (the reference's reshape raises otherwise, ``:287,295``; here ValueError)
        :return:  ``[batches_n, blocks_n + 1, filters_n, channels_n]`` amplitudes in ]-1, 1[
        """
        if isinstance(x, torch.Tensor) and x.requires_grad and torch.is_grad_enabled():
            return _TransformFn.apply(x, self)   # (every compute_dtype: the adjoint plan runs the same typed kernels)
        return self._transform(x)

    def _transform(self, x, adjoint=False):
        x = _host.check_device_tensor(x, "x", self.compute_dtype, 3)
        B, S, C = x.shape
        N = self.filters_n
        if S % N != 0:
            raise ValueError("samples_n (%d) is not a multiple of filters_n (%d)" % (S, N))
        K = S // N
        X = placement.empty(placement.REGION_SPECTRA, (B, K + 1, N, C), x.dtype, x.device)   # (see placement.py)
        with _host.on_device(x.device):
            _lib.check(self._lib.ac_mdct_forward_typed((self._adjoint_plans if adjoint else self._plans).get(x.device), _host.ptr(x), _host.ptr(X),
                                                       self._dtype_id, B, K, C, _host.stream_ptr(x.device)))
        return X

    # ---- synthesis -----------------------------------------------------------------------------------
    def inverse_transform(self, mdct_amplitudes):
        """MDCT synthesis filter bank (reference ``inverse_transform``, ``mdctransformer.py:128-153``).

        :param mdct_amplitudes: ``[batches_n, blocks_n, filters_n, channels_n]``
        :return:                ``[batches_n, (blocks_n + 1) * filters_n, channels_n]``
        """
        if isinstance(mdct_amplitudes, torch.Tensor) and mdct_amplitudes.requires_grad and torch.is_grad_enabled():
            return _InverseFn.apply(mdct_amplitudes, self)
        return self._inverse(mdct_amplitudes)

    def _inverse(self, mdct_amplitudes, adjoint=False):
        X = _host.check_device_tensor(mdct_amplitudes, "mdct_amplitudes", self.compute_dtype, 4)
        B, Kp, N, C = X.shape
        if N != self.filters_n:
            raise ValueError("axis 2 of mdct_amplitudes (%d) != filters_n (%d)" % (N, self.filters_n))
        x = placement.empty(placement.REGION_OTHER, (B, (Kp + 1) * N, C), X.dtype, X.device)
        with _host.on_device(X.device):
            _lib.check(self._lib.ac_mdct_inverse_typed((self._adjoint_plans if adjoint else self._plans).get(X.device), _host.ptr(X), _host.ptr(x),
                                                       self._dtype_id, B, Kp, C, _host.stream_ptr(X.device)))
        return x

    # native handle for the fused / streaming entry points
    def _plan(self, device):
        return self._plans.get(device)
