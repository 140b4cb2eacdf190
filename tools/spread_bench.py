#!/usr/bin/env python
"""The three forms of the spreading product (f32 VALU, bf16 MFMA, split-bf16 MFMA) side by side on the bench workload:
threshold deviation from the f32 form and kernel times of the fused encode and the stand-alone threshold (design aid)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N = int(os.environ.get("N", 1024))
B, K, C = int(os.environ.get("B", 256)), int(os.environ.get("K", 468 * 1024 // N)), 2
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); ref = torch.empty_like(X)


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for rounds in range(int(os.environ.get("ROUNDS", 2))):
    for mode in ("f32", "bf16_mfma", "bf16x2_mfma"):
        codec = audiocodec_amd.AudioCodec(48000, N, spreading=mode)
        codec.encode_into(x, X, t, thr)
        assert codec.psy.plan_spreading() == mode
        if mode == "f32":
            ref.copy_(thr)
        dev_rel = float(((thr - ref).abs() / ref).max())
        thr2 = codec.psy.global_masking_threshold(X, t)
        same = bool(torch.equal(thr2, thr))
        e = timeit(lambda: codec.encode_into(x, X, t, thr))
        g = timeit(lambda: codec.psy.global_masking_threshold(X, t))
        print("N=%d %-12s max rel dev of thr vs f32 %.3g   fused==unfused %s   encode %.3f ms   threshold %.3f ms"
              % (N, mode, dev_rel, same, e, g), flush=True)
