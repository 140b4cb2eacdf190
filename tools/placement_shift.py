#!/usr/bin/env python
"""Encode / decode time of the bench step with the five tensors laid out back to back inside one 28 GiB arena, against the
offset of the group inside the arena (in 2 MiB units)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
src = torch.rand(shapes["x"], device=dev) * 2 - 1
arena = torch.empty(28 * (1 << 30) // 4, dtype=torch.float32, device=dev)
U = 1 << 21


def med(fn, n=6):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def place(shift, gap):
    o, T = shift, {}
    for k, s in shapes.items():
        n = int(np.prod(s))
        T[k] = arena[o * U // 4: o * U // 4 + n].view(s)
        o += (n * 4 + U - 1) // U + gap
    return T


print("arena %#x" % arena.data_ptr())
T = place(0, 1); T["x"].copy_(src); med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
gap = int(os.environ.get("GAP", 1))
for shift in list(range(0, 12001, 500)) + [7000, 0, 7000, 0]:
    T = place(shift, gap)
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("shift %5d (x at %#x)  encode %.4f  decode %.4f" % (shift, T["x"].data_ptr(), e, d), flush=True)
