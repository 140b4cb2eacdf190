#!/usr/bin/env python
"""Encode / decode interleaved per chunk of clips (so that decode finds the chunk's X in the 256 MiB Infinity Cache)
against whole-batch encode then decode, on the bench workload (design aid)."""
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


def step(chunk):
    for b in range(0, B, chunk):
        e = min(B, b + chunk)
        codec.encode_into(x[b:e], X[b:e], t[b:e], thr[b:e])
        codec.decode_into(X[b:e], xh[b:e])


def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts)), float(np.min(ts))


for r in range(2):
    for chunk in (256, 128, 64, 32, 16, 8, 4):
        med, mn = timeit(lambda: step(chunk))
        print("chunk %3d clips: step %.3f ms (min %.3f)  %.1f M frames/s" % (chunk, med, mn, B * C * K / med / 1e3), flush=True)
