#!/usr/bin/env python
"""Timing of the backward passes on the bench workload (design aid)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd
N, B, K, C = 1024, 256, 468, 2
codec = audiocodec_amd.AudioCodec(48000, N)
x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
X = codec.mdct.transform(x)
t = codec.psy.tonality(X)
g = torch.rand_like(X)
gt = torch.rand_like(t)
def timeit(fn, n=10):
    import time
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.1:      # settle the device (DESIGN_LOG.md 5a)
        fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
nb = X.numel() * 4
for name, fn, passes in [("threshold backward", lambda: codec.psy._threshold_backward(X, t, 0.0, g), 3),
                         ("tonality backward", lambda: codec.psy._tonality_backward(X, gt), 2),
                         ("transform backward", lambda: codec.mdct._inverse(g), 2)]:
    ms = timeit(fn)
    print("%-22s %8.3f ms  %6.0f GB/s" % (name, ms, passes * nb / ms / 1e6))
