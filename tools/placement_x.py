#!/usr/bin/env python
"""With X and thr already in different classes of VRAM: does the place of the PCM the encode kernel reads matter?
(and of the PCM the decode kernel writes)"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
words = int(1.0 * (1 << 30)) // 4
chunks = [torch.empty(words, dtype=torch.float32, device=dev) for _ in range(int(os.environ.get("CHUNKS", 100)))]
view = lambda c, k: c[: int(np.prod(shapes[k]))].view(shapes[k])   # noqa: E731
src = torch.rand(shapes["x"], device=dev) * 2 - 1


def med(fn, n=5):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


X, t = view(chunks[0], "X"), torch.empty(shapes["t"], device=dev)
x0 = view(chunks[1], "x"); x0.copy_(src)
times = [med(lambda: codec.encode_into(x0, X, t, view(chunks[j], "thr"))) for j in range(2, len(chunks))]
print("thr in chunk j (x in chunk 1):", " ".join("%.3f" % v for v in times))
jb = 2 + int(np.argmin(times))
thr = view(chunks[jb], "thr")
row = []
for k in range(1, len(chunks)):
    if k == jb:
        row.append("  -  "); continue
    xk = view(chunks[k], "x"); xk.copy_(src)
    row.append("%.3f" % med(lambda: codec.encode_into(xk, X, t, thr)))
print("thr in chunk %d; x in chunk k = 1..: " % jb + " ".join(row))
codec.encode_into(x0, X, t, thr)
row = []
for k in range(1, len(chunks)):
    if k == jb:
        row.append("  -  "); continue
    row.append("%.3f" % med(lambda: codec.decode_into(X, view(chunks[k], "xh")[:, : (K + 2) * N])))
print("decode, xh in chunk k = 1..:        " + " ".join(row))
