import os, sys, torch
sys.path.insert(0, os.getcwd())
import audiocodec_amd
from audiocodec_amd import _lib
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for N in [int(v) for v in os.environ.get("SIZES", "960,480,240,120,576,1920,1536,192,4096,32,16").split(",")]:
    C = int(os.environ.get("C", 2)); B = int(os.environ.get("B", 128 // C))
    K = 468 * 1024 // N
    x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
    m = audiocodec_amd.MDCTransformer(N)
    X = m.transform(x)
    tf, ti = timeit(lambda: m.transform(x)), timeit(lambda: m.inverse_transform(X))
    frames = B * C * K
    print("N %5d  transform %7.3f ms %6.0f GB/s   inverse %7.3f ms %6.0f GB/s" % (N, tf, 8 * N * frames / tf / 1e6, ti, 8 * N * frames / ti / 1e6))
