#!/usr/bin/env python
"""Step time of the bench step (encode + decode) against time since the GPU became busy: how long do the clocks take to settle?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
codec = audiocodec_amd.AudioCodec(48000, N)
codec.encode_into(x, X, t, thr); codec.decode_into(X, xh); torch.cuda.synchronize()
import time; time.sleep(float(os.environ.get("IDLE", 1.0)))
n = int(os.environ.get("STEPS", 800))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
ev[0].record()
for i in range(n):
    codec.encode_into(x, X, t, thr); codec.decode_into(X, xh); ev[i + 1].record()
torch.cuda.synchronize()
ts = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(n)])
cum = np.cumsum(ts)
for lo in range(0, n, 50):
    print("steps %3d-%3d  t = %6.1f ms   %.4f ms/step  (%.1f M frames/s)" % (lo, lo + 49, cum[lo], ts[lo:lo + 50].mean(), B * C * K / ts[lo:lo + 50].mean() / 1e3), flush=True)
