#!/usr/bin/env python
"""Run one entry point a few times on the bench workload (profiling driver)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd
what = sys.argv[1] if len(sys.argv) > 1 else "encode"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N = int(os.environ.get("N", 1024))
B, K, C = int(os.environ.get("B", 256)), int(os.environ.get("K", 468 * 1024 // N)), int(os.environ.get("C", 2))
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
codec = audiocodec_amd.AudioCodec(48000, N)
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
codec.encode_into(x, X, t, thr)
torch.cuda.synchronize()
for _ in range(reps):
    if what == "encode": codec.encode_into(x, X, t, thr)
    elif what == "transform": codec.mdct.transform(x)
    elif what == "inverse": codec.decode_into(X, xh)
    elif what == "psy": codec.psy.global_masking_threshold(X, t)
torch.cuda.synchronize()
