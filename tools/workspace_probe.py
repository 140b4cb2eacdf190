#!/usr/bin/env python
"""Workspace placement tuning on the bench workload: report + encode / decode times against plain torch allocations."""
import os, sys, json, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)


def med(fn, n=10):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


x = torch.rand((B, K * N, C), device=dev) * 2 - 1
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
print("plain torch tensors: encode %.4f  decode %.4f" % (med(lambda: codec.encode_into(x, X, t, thr)), med(lambda: codec.decode_into(X, xh))))
t0 = time.perf_counter()
ws = audiocodec_amd.Workspace(codec, B, K, C, span_gib=float(os.environ.get("SPAN", 112)))
torch.cuda.synchronize()
print("workspace built in %.2f s" % (time.perf_counter() - t0))
print(json.dumps(ws.report))
ws.x.copy_(x)
e = med(lambda: codec.encode_into(ws.x, ws.X, ws.t, ws.thr)); d = med(lambda: codec.decode_into(ws.X, ws.xh))
print("workspace:           encode %.4f  decode %.4f   round trip %.2e" % (e, d, float((ws.xh[:, N:-N] - ws.x).abs().max())))
print("plain torch tensors: encode %.4f  decode %.4f" % (med(lambda: codec.encode_into(x, X, t, thr)), med(lambda: codec.decode_into(X, xh))))
