#!/usr/bin/env python
"""encode() of the LDS-FFT tier, one fused launch against transform + masking kernel, over the sizes with an instance
(B clips of ~2.5 s stereo): the table behind the policy of wave_encode_fuses (profiles/r4/lds_fft_fused_encode_sweep.txt)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd
B, C = int(os.environ.get("B", 256)), int(os.environ.get("C", 2))
sizes = [int(a) for a in os.environ.get("SIZES", "").split(",") if a] or [
    288, 320, 384, 480, 500, 540, 576, 600, 640, 648, 720, 768, 800, 864, 900, 960, 972, 1000, 1080, 1152, 1200, 1280, 1296, 1440, 1500, 1536, 1600, 1620,
    1728, 1800, 1920, 1944, 2000, 2160, 2304, 2400, 2500, 2560, 2592, 2700, 2880, 2916, 3000, 3072, 3200, 3240, 3456, 3600, 3840, 3888, 4000, 4096]
def timeit(fn, n=10):
    t_end = time.perf_counter() + 0.05
    while time.perf_counter() < t_end:
        for _ in range(4): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("filters_n  fused ms  two launches ms  ratio   (B = %d, C = %d)" % (B, C))
for N in sizes:
    K = max(4, int(os.environ.get("SAMPLES", 480000)) // N)
    codec = audiocodec_amd.AudioCodec(48000, N)
    x = torch.rand((B, K * N, C), device="cuda") * 2 - 1
    X = torch.empty((B, K + 1, N, C), device="cuda"); t = torch.empty((B, K + 1, 1, C), device="cuda"); thr = torch.empty_like(X)
    os.environ["AC_LDS_WAVE_NOFUSE"] = "2"    # (2: the policy table off, every instance fused)
    a = timeit(lambda: codec.encode_into(x, X, t, thr))
    os.environ["AC_LDS_WAVE_NOFUSE"] = "1"
    b = timeit(lambda: codec.encode_into(x, X, t, thr))
    del os.environ["AC_LDS_WAVE_NOFUSE"]
    print("%8d  %8.3f  %8.3f  %6.3f" % (N, a, b, a / b), flush=True)
