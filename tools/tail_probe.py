#!/usr/bin/env python
"""Does the last partial round of workgroups cost visible time?  ns per frame of encode / decode for neighbouring batch sizes."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, K, C = 1024, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)


def med(fn, n=15):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for rnd in range(2):
    for B in (256, 254, 250, 246, 240, 236, 230, 224):
        x = torch.rand((B, K * N, C), device=dev) * 2 - 1
        X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
        thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
        e = med(lambda: codec.encode_into(x, X, t, thr)); d = med(lambda: codec.decode_into(X, xh))
        fr = B * C * K
        enc_wg = (B * (K + 1) + 15) // 16; dec_wg = (B * ((K + 2 + 2) // 3) + 3) // 4
        print("B %3d  encode %.3f ms %.3f ns/frame (%5d WGs = %.2f rounds)   decode %.3f ms %.3f ns/frame (%5d WGs = %.2f rounds)"
              % (B, e, e * 1e6 / fr, enc_wg, enc_wg / 768, d, d * 1e6 / fr, dec_wg, dec_wg / 768), flush=True)
        del x, X, t, thr, xh
