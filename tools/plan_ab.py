"""A/B of radix plans of the LDS-FFT tier: library variants built with another AC_WAVE_CT_SIZES list (tools/build_variant.sh
NAME '-DAC_WAVE_CT_SIZES=AC_WAVE_CT(800, 64, 10, 10, 4, 0) ...') against the product library, transform / inverse on the
SAME tensors in one process (placement differs from process to process by more than most plans do).
    python tools/plan_ab.py 800,1152 name=path.so [name=path.so ...]      (B clips of 10 s stereo; env B, default 256)"""
import ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from audiocodec_amd import _lib as L
sizes = [int(v) for v in sys.argv[1].split(",")]
B, C = int(os.environ.get("B", 256)), 2
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
libs = []
for v in sys.argv[2:]:
    name, path = v.split("=", 1)
    lib = ctypes.CDLL(os.path.abspath(path))
    for fn_name, (restype, argtypes) in L.PROTOTYPES.items():
        if hasattr(lib, fn_name):
            fn = getattr(lib, fn_name); fn.restype = restype; fn.argtypes = argtypes
    libs.append((name, lib))
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t_: ctypes.c_void_p(t_.data_ptr())
for N in sizes:
    K = 480000 // N
    x = torch.rand((B, K * N, C), device=dev) * 2 - 1
    X = torch.empty((B, K + 1, N, C), device=dev); xh = torch.empty((B, (K + 2) * N, C), device=dev)
    plans, ref = [], None
    for name, lib in libs:
        mp = ctypes.c_void_p()
        assert lib.ac_mdct_plan_create(N, 0, 0, ctypes.byref(mp)) == 0
        plans.append(mp)
        assert lib.ac_mdct_forward(mp, P(x), P(X), B, K, C, stream) == 0
        assert lib.ac_mdct_inverse(mp, P(X), P(xh), B, K + 1, C, stream) == 0
        torch.cuda.synchronize()
        cur = (X.clone(), xh.clone())
        if ref is None: ref = cur
        else:
            dX = float((cur[0] - ref[0]).abs().max() / ref[0].abs().max()); dx = float((cur[1] - ref[1]).abs().max())
            assert dX < 1e-5 and dx < 1e-5, (name, N, dX, dx)
    res = {name: ([], []) for name, _ in libs}
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:
        libs[0][1].ac_mdct_forward(plans[0], P(x), P(X), B, K, C, stream); torch.cuda.synchronize()
    for r in range(5):
        for (name, lib), mp in zip(libs, plans):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            for _ in range(2):
                lib.ac_mdct_forward(mp, P(x), P(X), B, K, C, stream); lib.ac_mdct_inverse(mp, P(X), P(xh), B, K + 1, C, stream)
            ev[0].record()
            for _ in range(5): lib.ac_mdct_forward(mp, P(x), P(X), B, K, C, stream)
            ev[1].record()
            for _ in range(5): lib.ac_mdct_inverse(mp, P(X), P(xh), B, K + 1, C, stream)
            ev[2].record(); torch.cuda.synchronize()
            res[name][0].append(ev[0].elapsed_time(ev[1]) / 5); res[name][1].append(ev[1].elapsed_time(ev[2]) / 5)
    print("N %5d  " % N + "   ".join("%s %.3f / %.3f" % (name, np.median(res[name][0]), np.median(res[name][1])) for name, _ in libs), flush=True)
    for (name, lib), mp in zip(libs, plans): lib.ac_mdct_plan_destroy(mp)
    del x, X, xh
