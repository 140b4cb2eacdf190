"""How long an idle gap re-triggers the start-of-load transient (MI355X): 120 settled steps, a gap, 40 steps timed per step."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
import numpy as np, torch, audiocodec_amd
dev = torch.device("cuda", 0)
N, B, K, C = 1024, 256, 468, 2
codec = audiocodec_amd.AudioCodec(48000, N)
x = bench.make_clips(torch, dev, 0, B, K)
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
def steps(n, timed=False):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)] if timed else None
    for i in range(n):
        if timed: ev[i][0].record()
        codec.encode_into(x, X, t, thr)
        if timed: ev[i][1].record()
        codec.decode_into(X, xh)
        if timed: ev[i][2].record()
    torch.cuda.synchronize()
    if timed:
        return [e[0].elapsed_time(e[1]) for e in ev], [e[1].elapsed_time(e[2]) for e in ev]
for gap_ms in (0.0, 0.1, 0.3, 1.0, 2.0, 5.0, 20.0):
    steps(150)
    if gap_ms > 0:
        t0 = time.perf_counter()
        while (time.perf_counter() - t0) * 1e3 < gap_ms: pass
    enc, dec = steps(40, True)
    print("gap %5.1f ms  enc mean(first 20) %.3f  all: %s" % (gap_ms, float(np.mean(enc[:20])), " ".join("%.3f" % v for v in enc[:24])))
    print("              dec mean(first 20) %.3f  step mean %.3f" % (float(np.mean(dec[:20])), float(np.mean(enc[:20]) + np.mean(dec[:20]))))
# settle by a lighter kernel? a torch copy loop then the steps
for _ in range(2):
    time.sleep(1.0)
    for i in range(150): xh[:, :K * N].copy_(x)
    torch.cuda.synchronize()
    enc, dec = steps(40, True)
    print("after 150 torch copies: enc", " ".join("%.3f" % v for v in enc[:24]))
# encode-only and decode-only steady state, for reference
time.sleep(1.0); steps(150)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(101)]
for i in range(100):
    ev[i].record(); codec.encode_into(x, X, t, thr)
ev[100].record(); torch.cuda.synchronize()
print("encode only steady", np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(100)]))
for i in range(100):
    ev[i].record(); codec.decode_into(X, xh)
ev[100].record(); torch.cuda.synchronize()
print("decode only steady", np.mean([ev[i].elapsed_time(ev[i + 1]) for i in range(100)]))
