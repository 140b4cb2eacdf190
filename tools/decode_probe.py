#!/usr/bin/env python
"""Decode (and encode) time for one N on Workspace-placed tensors; used to sweep AC_SEGLEN per filters_n."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N = int(os.environ.get("N", 2048))
B, K, C = 256, 468 * 1024 // N, int(os.environ.get("C", 2))
codec = audiocodec_amd.AudioCodec(48000, N)
ws = codec.workspace(B, K, C)
ws.x.copy_(torch.rand(ws.x.shape, device="cuda") * 2 - 1)


def med(fn, n=20):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


e = med(lambda: codec.encode_into(ws.x, ws.X, ws.t, ws.thr)); d = med(lambda: codec.decode_into(ws.X, ws.xh))
print("N=%d C=%d AC_SEGLEN=%s  encode %.4f ms  decode %.4f ms" % (N, C, os.environ.get("AC_SEGLEN", "default"), e, d))
