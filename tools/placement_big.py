#!/usr/bin/env python
"""One very large arena: X near its start, thr at offsets across the whole arena -> map of fast / slow regions."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
GB = 1 << 30
AG = int(os.environ.get("ARENA", 224))
free, total = torch.cuda.mem_get_info()
print("free %.1f GiB of %.1f" % (free / GB, total / GB))
arena = torch.empty(AG * GB // 4, dtype=torch.float32, device=dev)
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C)}


def at(off_gib, k):
    o = int(off_gib * GB) // 4
    return arena[o: o + int(np.prod(shapes[k]))].view(shapes[k])


def med(fn, n=4):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


x, t = at(0, "x"), at(1.0, "t")
x.copy_(torch.rand(shapes["x"], device=dev) * 2 - 1)
for Xo in (2.0, float(os.environ.get("X2", 100.0))):
    X = at(Xo, "X")
    row = []
    for o in np.arange(4.0, AG - 1.0, float(os.environ.get("STEP", 4.0))):
        if abs(o - Xo) < 1.0:
            row.append("  -  ")
            continue
        thr = at(o, "thr")
        row.append("%.3f" % med(lambda: codec.encode_into(x, X, t, thr)))
    print("X at %5.1f GiB; thr at 4, 8, ...:" % Xo)
    for i in range(0, len(row), 16):
        print("   ", " ".join(row[i:i + 16]), flush=True)
