"""numpy/scipy stand-in for the handful of TensorFlow ops the reference calls.

TEST INFRASTRUCTURE ONLY, build-owned code.  TensorFlow is not installable in
this image (no wheel, no network).  Putting this package ahead on
``PYTHONPATH`` lets ``oracle/gen_golden.py`` import and execute the reference's
*unmodified* modules from ``/root/reference`` so that their own control flow,
constants and quirks produce the golden vectors under ``tests/golden/``; numpy
does the element arithmetic, ``scipy.fft.dct`` stands in for ``tf.signal.dct``
and a 2-tap einsum for ``tf.nn.convolution``.  It is NOT TensorFlow: kernel-level
rounding of the real library is not reproduced (see oracle/audiocodec_oracle.py,
"Parity pin status").  Only the semantics listed in SURVEY.md Appendix B are
implemented; anything else raises AttributeError on purpose.
"""

import contextlib
import types

import numpy as _np
import scipy.fft as _fft

float16 = _np.dtype("float16")
float32 = _np.dtype("float32")
float64 = _np.dtype("float64")
int32 = _np.dtype("int32")
int64 = _np.dtype("int64")


class _BFloat16:
    """Only has to exist for the ``compute_dtype in [...]`` check (psychoacoustic.py:42)."""
    name = "bfloat16"

    def __repr__(self):
        return "tf.bfloat16"


bfloat16 = _BFloat16()
Tensor = _np.ndarray
DType = _np.dtype


def function(fn=None, **_kw):
    if fn is None:
        return lambda f: f
    return fn


@contextlib.contextmanager
def name_scope(_name):
    yield


def _f(x):
    """A bare Python float becomes a float32 tensor in TF (e.g. ``tf.sqrt(2.)``)."""
    if type(x) is float:
        return _np.float32(x)
    return x


def cast(x, dtype):
    return _np.asarray(x).astype(dtype)[()]


def constant(value, dtype=None, shape=None):
    a = _np.asarray(value, dtype=dtype if dtype is not None else (float32 if isinstance(value, float) else None))
    if shape is not None:
        a = _np.broadcast_to(a, shape).copy()
    return a[()]


def convert_to_tensor(value, dtype=None):
    return _np.asarray(value, dtype=dtype)


def shape(x):
    return _np.asarray(_np.shape(x))


def range(start, limit=None, delta=1, dtype=None):  # noqa: A001
    if limit is None:
        start, limit = 0, start
    if dtype is None:
        dtype = float32 if any(isinstance(v, float) for v in (start, limit, delta)) else int32
    return _np.arange(start, limit, delta).astype(dtype)


def linspace(start, stop, num):
    dt = _np.asarray(start).dtype
    if dt.kind != "f":
        dt = float32
    start = _np.asarray(start, dtype=dt)
    stop = _np.asarray(stop, dtype=dt)
    if num == 1:
        return _np.asarray([start], dtype=dt)
    step = (stop - start) / dt.type(num - 1)
    out = (start + step * _np.arange(num).astype(dt)).astype(dt)
    out[-1] = stop
    return out


def zeros(shape, dtype=float32):
    return _np.zeros(_np.asarray(shape).astype(int).reshape(-1).tolist() if not isinstance(shape, int) else shape,
                     dtype=dtype)


def ones(shape, dtype=float32):
    return _np.ones(_np.asarray(shape).astype(int).reshape(-1).tolist() if not isinstance(shape, int) else shape,
                    dtype=dtype)


def reshape(x, shape):
    return _np.reshape(x, [int(s) for s in _np.asarray(shape).reshape(-1)])


def transpose(x, perm=None):
    return _np.transpose(x, perm)


def expand_dims(x, axis):
    return _np.expand_dims(x, axis)


def reverse(x, axis):
    return _np.flip(x, axis=tuple(axis))


def concat(values, axis):
    return _np.concatenate(values, axis=axis)


def stack(values, axis=0):
    return _np.stack(values, axis=axis)


def pad(x, paddings):
    return _np.pad(x, _np.asarray(paddings).astype(int).tolist())


def broadcast_to(x, shape):
    return _np.broadcast_to(x, shape)


def einsum(eq, *ops):
    return _np.einsum(eq, *ops)


def map_fn(fn, elems):
    return _np.stack([fn(e) for e in elems])


def sin(x):
    return _np.sin(_f(x))


def exp(x):
    return _np.exp(_f(x))


def sqrt(x):
    return _np.sqrt(_f(x))


def pow(x, y):  # noqa: A001
    return _np.power(_f(x), y)


def asinh(x):
    return _np.arcsinh(_f(x))


def sinh(x):
    return _np.sinh(_f(x))


def abs(x):  # noqa: A001
    return _np.abs(x)


def maximum(x, y):
    return _np.maximum(x, y)


def minimum(x, y):
    return _np.minimum(x, y)


def divide(x, y):
    return _np.divide(x, y)


def clip_by_value(t, clip_value_min, clip_value_max):
    return _np.clip(t, clip_value_min, clip_value_max)


def _reduce(fn):
    def red(x, axis=None, keepdims=False):
        x = _np.asarray(x)
        return fn(x, axis=axis, keepdims=keepdims, **({"dtype": x.dtype} if fn in (_np.mean, _np.sum) else {}))
    return red


reduce_mean = _reduce(_np.mean)
reduce_sum = _reduce(_np.sum)
reduce_max = _reduce(_np.max)


def _log(x):
    return _np.log(_f(x))


math = types.SimpleNamespace(log=_log, exp=exp, pow=pow, sqrt=sqrt, maximum=maximum, minimum=minimum)


def _diag(v):
    return _np.diag(_np.asarray(v))


linalg = types.SimpleNamespace(diag=_diag, inv=_np.linalg.inv)


def _dct(x, type=2, n=None, axis=-1, norm=None):  # noqa: A002
    return _fft.dct(_np.asarray(x), type=type, n=n, axis=axis, norm=norm).astype(_np.asarray(x).dtype)


signal = types.SimpleNamespace(dct=_dct)


def _convolution(input, filters, padding="VALID"):  # noqa: A002
    """1-D VALID cross-correlation: out[b,w,o] = sum_k sum_c in[b,w+k,c] * f[k,c,o]."""
    assert padding == "VALID"
    a = _np.asarray(input)
    f = _np.asarray(filters)
    taps = f.shape[0]
    wout = a.shape[1] - taps + 1
    out = None
    for k in _np.arange(taps):
        term = _np.einsum("bwc,co->bwo", a[:, k:k + wout, :], f[k])
        out = term if out is None else out + term
    return out


nn = types.SimpleNamespace(convolution=_convolution)

_rng = _np.random.default_rng(12345)


def _normal(shape, mean=0.0, stddev=1.0, dtype=float32):
    return (mean + stddev * _rng.standard_normal(tuple(shape))).astype(dtype)


def _uniform(shape, minval=0.0, maxval=1.0, dtype=float32):
    return _rng.uniform(minval, maxval, tuple(shape)).astype(dtype)


random = types.SimpleNamespace(normal=_normal, uniform=_uniform)


class _Errors:
    class InvalidArgumentError(ValueError):
        pass


errors = _Errors()
