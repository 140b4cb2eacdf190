import os, sys, torch
sys.path.insert(0, os.getcwd())
import audiocodec_amd
N, B, K, C = 1024, 512, 468, 1
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev); thr = torch.empty_like(X)
codecs = {m: audiocodec_amd.AudioCodec(48000, N, spreading=m) for m in ("f32", "bf16x2_mfma", "bf16_mfma")}
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for m, c in codecs.items(): c.encode_into(x, X, t, thr)
for rnd in range(3):
    for m, c in codecs.items():
        ms = timeit(lambda: c.encode_into(x, X, t, thr))
        print("round %d  mono B=512  %-12s encode %.4f ms  %.0f GB/s  frac %.3f" % (rnd, m, ms, 12292 * B * C * K / ms / 1e6, 12292 * B * C * K / ms / 1e6 / 8000))
ref = torch.empty_like(thr); codecs["f32"].encode_into(x, X, t, ref); codecs["bf16x2_mfma"].encode_into(x, X, t, thr)
print("thr max rel dev bf16x2 vs f32: %.3g" % float(((thr - ref).abs() / ref).max()))
