"""ctypes binding of ``libaudiocodec_amd.so`` (the C ABI declared in ``include/audiocodec_amd.h``).

There is no fallback: if the shared library is missing or a call fails the product raises.
"""

from __future__ import annotations

import ctypes
import os
import threading
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("AUDIOCODEC_AMD_LIB") or os.path.join(_HERE, "lib", "libaudiocodec_amd.so")

AC_OK = 0
AC_EINVAL, AC_EHIP, AC_ENOMEM, AC_ENODEV, AC_EUNSUPPORTED = -1, -2, -3, -4, -5
WINDOW_IDS = {"vorbis": 0, "sine": 1}   # anything else -> 2 (rectangular), mdctransformer.py:199-211
WINDOW_RECT = 2

# name -> (restype, argtypes); mirrors include/audiocodec_amd.h (+ the test hook of audiocodec_amd_testing.h) one to one
PROTOTYPES = {
    "ac_version": (c_int, []),
    "ac_last_error": (c_char_p, []),
    "ac_set_force_generic": (c_int, [c_int]),
    "ac_testing_runs_image": (c_int, [c_int, c_int, c_double, c_double, c_int, POINTER(ctypes.c_uint32), c_int, POINTER(c_int)]),
    "ac_mdct_fold_coefficients_host": (c_int, [c_int, c_int, POINTER(c_double)]),
    "ac_mdct_dense_matrices_host": (c_int, [c_int, c_int, POINTER(c_float), POINTER(c_float)]),
    "ac_psy_tables_host": (c_int, [c_int, c_int, c_double, c_double, POINTER(c_float), POINTER(c_float),
                                   POINTER(c_float), POINTER(c_float), POINTER(c_double)]),
    "ac_psy_tables_host_f64": (c_int, [c_int, c_int, c_double, c_double, POINTER(c_double), POINTER(c_double),
                                       POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "ac_mdct_fold_coefficients_host_pre": (c_int, [c_int, c_int, c_int, POINTER(c_double)]),
    "ac_mdct_dense_matrices_host_pre": (c_int, [c_int, c_int, c_int, POINTER(c_float), POINTER(c_float)]),
    "ac_psy_tables_host_pre": (c_int, [c_int, c_int, c_double, c_double, c_int, POINTER(c_double), POINTER(c_double),
                                       POINTER(c_double), POINTER(c_double), POINTER(c_double)]),
    "ac_mdct_plan_create_pre": (c_int, [c_int, c_int, c_int, c_int, POINTER(c_void_p)]),
    "ac_psy_plan_create_pre": (c_int, [c_int, c_int, c_double, c_double, c_int, c_int, c_int, POINTER(c_void_p)]),
    "ac_mdct_plan_adjoint": (c_int, [c_void_p, POINTER(c_void_p)]),
    "ac_mdct_plan_create": (c_int, [c_int, c_int, c_int, POINTER(c_void_p)]),
    "ac_mdct_plan_destroy": (c_int, [c_void_p]),
    "ac_psy_plan_create": (c_int, [c_int, c_int, c_double, c_double, c_int, POINTER(c_void_p)]),
    "ac_psy_plan_destroy": (c_int, [c_void_p]),
    "ac_mdct_plan_is_fast": (c_int, [c_void_p]),
    "ac_mdct_plan_tier": (c_int, [c_void_p, c_int]),
    "ac_psy_plan_is_fast": (c_int, [c_void_p]),
    "ac_psy_plan_tier": (c_int, [c_void_p]),
    "ac_psy_plan_create_ex": (c_int, [c_int, c_int, c_double, c_double, c_int, c_int, POINTER(c_void_p)]),
    "ac_psy_plan_spreading": (c_int, [c_void_p]),
    "ac_mdct_forward_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_mdct_inverse_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_tonality_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_mask_threshold_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_int, c_int, c_int, c_int,
                                        c_void_p]),
    "ac_tonality_backward_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_mask_threshold_backward_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int,
                                                 c_int, c_void_p]),
    "ac_workspace_create": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_double, c_void_p, POINTER(c_void_p)]),
    "ac_workspace_buffers": (c_int, [c_void_p, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                                     POINTER(c_void_p)]),
    "ac_workspace_regions": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t), POINTER(c_void_p), POINTER(c_size_t)]),
    "ac_workspace_report": (c_int, [c_void_p, POINTER(c_int), POINTER(c_int), POINTER(c_float), POINTER(c_double)]),
    "ac_workspace_destroy": (c_int, [c_void_p]),
    "ac_workspace_alloc_dlpack": (c_void_p, [c_void_p, c_int, c_int, POINTER(ctypes.c_int64), c_void_p]),
    "ac_workspace_live": (ctypes.c_long, [c_void_p]),
    "ac_workspace_record_stream": (c_int, [c_void_p, c_void_p, c_void_p]),
    "ac_probe_placement": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, POINTER(c_void_p), c_int, c_int, c_int,
                                   c_int, c_void_p, POINTER(c_int), POINTER(c_float)]),
    "ac_encode_fused_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int,
                                      c_int, c_int, c_void_p]),
    "ac_stream_encode_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_int, c_int, c_void_p]),
    "ac_stream_inverse_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ac_amplitude_to_db_typed": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_int, c_void_p]),
    "ac_add_noise_typed": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint64, c_int, c_void_p]),
    "ac_mdct_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_mdct_inverse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_tonality": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_mask_threshold": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_tonality_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ac_mask_threshold_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_void_p, c_void_p, c_void_p, c_int,
                                           c_int, c_int, c_void_p]),
    "ac_encode_launches": (c_int, [c_void_p, c_void_p, c_int]),
    "ac_encode_fused": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_int,
                                c_int, c_void_p]),
    "ac_encode_fused_ex": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p,
                                   c_void_p, c_uint64, c_int, c_int, c_int, c_void_p]),
    "ac_amplitude_to_db_backward": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ac_mdct_forward_pcm16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_mdct_inverse_pcm16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ac_encode_fused_pcm16": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int,
                                      c_int, c_int, c_void_p]),
    "ac_stream_create": (c_int, [c_void_p, c_int, c_int, POINTER(c_void_p)]),
    "ac_stream_reset": (c_int, [c_void_p, c_void_p]),
    "ac_stream_destroy": (c_int, [c_void_p]),
    "ac_stream_forward": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ac_stream_inverse": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ac_stream_settle": (c_int, [c_void_p, c_void_p]),
    "ac_stream_run": (c_int, [c_void_p, c_void_p, c_int, c_int, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p),
                              POINTER(c_void_p), POINTER(c_void_p), c_float, c_void_p]),
    "ac_stream_encode": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_float, c_int, c_void_p]),
    "ac_amplitude_to_db": (c_int, [c_void_p, c_void_p, c_size_t, c_int, c_void_p]),
    "ac_add_noise": (c_int, [c_void_p, c_void_p, c_void_p, c_size_t, c_uint64, c_void_p]),
}

_lock = threading.Lock()
_lib = None


class AudioCodecError(RuntimeError):
    """A libaudiocodec_amd call returned a non-zero status."""

    def __init__(self, status, message):
        super().__init__("libaudiocodec_amd status %d: %s" % (status, message))
        self.status = status


def load():
    """Load (once) and return the ctypes handle.  Raises if the library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH) and not os.environ.get("AUDIOCODEC_AMD_LIB"):
            _build_in_tree()
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "%s not found -- build it with `python -c 'import __graft_entry__ as g; g.build()'` or "
                "`make -C audiocodec_amd/csrc`.  There is no CPU fallback.%s"
                % (LIB_PATH, ("\n--- output of the in-tree build attempt ---\n" + _build_log) if _build_log else ""))
        lib = ctypes.CDLL(LIB_PATH)
        for name, (restype, argtypes) in PROTOTYPES.items():
            fn = getattr(lib, name)          # AttributeError here = header/library mismatch
            fn.restype = restype
            fn.argtypes = argtypes
        if lib.ac_version() < 171:
            raise ImportError("libaudiocodec_amd too old: %d" % lib.ac_version())
        _lib = lib
    return _lib


_build_log = ""


def _build_in_tree():
    """The library is built in-tree by ``__graft_entry__.build()`` / ``make -C audiocodec_amd/csrc``; when it is missing
    (a fresh checkout) try that once -- hipcc cross-compiles for gfx950 without a GPU.  Ranks importing at the same time
    serialise on a file lock; the build links into a scratch directory and renames the library into place, so nobody can
    dlopen a half-written file.  A failure is not hidden: load() raises ImportError with the compiler's output."""
    global _build_log
    import fcntl
    import shutil
    import subprocess
    import tempfile
    # under a profiler (rocprofv3 preloads a library that initialises the GPU in every child) the make / hipcc / clang
    # children would exec from GPU-initialised processes, which this pool forbids: refuse, do not build
    preload = [k for k in os.environ if k.startswith(("ROCP", "ROCPROFILER", "ROCTRACER"))]
    if preload or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        _build_log = ("not building under a profiler (%s set): run `make -C audiocodec_amd/csrc` (or __graft_entry__.build()) "
                      "before the profiled command" % ", ".join(preload or ["LD_PRELOAD"]))
        return
    libdir = os.path.join(_HERE, "lib")
    os.makedirs(libdir, exist_ok=True)
    with open(os.path.join(libdir, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(LIB_PATH):          # another process built it while this one waited
                return
            tmp = tempfile.mkdtemp(prefix="build_", dir=libdir)
            try:
                r = subprocess.run(["make", "-C", os.path.join(_HERE, "csrc"), "-j4", "OUTDIR=" + tmp],
                                   stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1800)
                _build_log = r.stdout[-4000:]
                if r.returncode == 0:
                    os.replace(os.path.join(tmp, "libaudiocodec_amd.so"), LIB_PATH)
            except (OSError, subprocess.SubprocessError) as e:
                _build_log = "%s: %s" % (type(e).__name__, e)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)


def check(status):
    """Turn a non-zero status into an exception (ValueError for AC_EINVAL, like a shape error)."""
    if status == AC_OK:
        return
    msg = load().ac_last_error()
    msg = msg.decode("utf-8", "replace") if msg else ""
    if status == AC_EINVAL:
        raise ValueError(msg)
    raise AudioCodecError(status, msg)


def window_id(window_type):
    """Reference semantics (mdctransformer.py:199-211): case-insensitive 'sine' / 'vorbis', else rectangular.
    ``None`` selects the rectangular window (the reference raises AttributeError; documented fix)."""
    if isinstance(window_type, str):
        return WINDOW_IDS.get(window_type.lower(), WINDOW_RECT)
    if window_type is None:
        return WINDOW_RECT
    raise TypeError("window_type must be a string or None")
