#!/usr/bin/env python
"""Structured scan of relative tensor placement: x at a 1 GiB boundary, X shifted by s, thr by 2 s (and variants)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
arena = torch.empty(8 * (1 << 30) // 4, dtype=torch.float32, device=dev)
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
src = torch.rand(shapes["x"], device=dev) * 2 - 1
slot = 1 << 30


def carve(offsets):
    return {k: arena[offsets[k] // 4: offsets[k] // 4 + int(np.prod(s))].view(s) for k, s in shapes.items()}


def med(fn, n=12):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def run(tag, sx, sX, sthr, sxh):
    offs = {"x": 0 * slot + sx, "X": 1 * slot + sX, "thr": 2 * slot + sthr, "t": 3 * slot, "xh": 4 * slot + sxh}
    T = carve(offs)
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("%-34s encode %.4f ms  decode %.4f ms" % (tag, e, d), flush=True)


run("all aligned", 0, 0, 0, 0)
for s in (256, 512, 1024, 2048, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1 << 20, 1 << 21, 1 << 22):
    run("X +s, thr +2s, xh +3s  s=%d" % s, 0, s, 2 * s, 3 * s)
run("all aligned (again)", 0, 0, 0, 0)
for s in (4096, 65536, 1 << 20):
    run("only thr +s  s=%d" % s, 0, 0, s, 0)
    run("only X +s    s=%d" % s, 0, s, 0, 0)
    run("X, thr +s (x apart)  s=%d" % s, 0, s, s, 0)
