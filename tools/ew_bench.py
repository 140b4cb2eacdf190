import os, sys, torch
sys.path.insert(0, os.getcwd())
import audiocodec_amd
p = audiocodec_amd.PsychoacousticModel(48000)
X = torch.rand(256, 469, 1024, 2, device="cuda") - 0.5
thr = torch.rand_like(X) * 0.1
def timeit(fn, n=20):
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
print("add_noise   %.3f ms  %.0f GB/s" % ((t := timeit(lambda: p.add_noise(X, thr, seed=1))), 3 * nb / t / 1e6))
print("dB          %.3f ms  %.0f GB/s" % ((t := timeit(lambda: p.amplitude_to_dB(X))), 2 * nb / t / 1e6))
print("dB_norm     %.3f ms  %.0f GB/s" % ((t := timeit(lambda: p.amplitude_to_dB_norm(X))), 2 * nb / t / 1e6))
out = torch.empty_like(X)
print("torch add (2R:1W) %.3f ms  %.0f GB/s" % ((t := timeit(lambda: torch.add(X, thr, out=out))), 3 * nb / t / 1e6))
print("torch abs (1R:1W) %.3f ms  %.0f GB/s" % ((t := timeit(lambda: torch.abs(X, out=out))), 2 * nb / t / 1e6))
