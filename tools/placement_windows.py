#!/usr/bin/env python
"""Encode / decode time against the 4 GiB-aligned address windows the tensors of the bench step sit in (one 40 GiB arena)."""
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
W = 1 << int(os.environ.get("WBITS", 32))
arena = torch.empty(40 * (1 << 30) // 4, dtype=torch.float32, device=dev)
base = arena.data_ptr()
w0 = (base + W - 1) // W * W          # first window boundary inside the arena


def med(fn, n=6):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def at(addr, shape):
    off = (addr - base) // 4
    return arena[off: off + int(np.prod(shape))].view(shape)


def run(tag, win, inner):
    """win[k] = window index (relative to w0), inner[k] = byte offset inside the window"""
    T = {k: at(w0 + win[k] * W + inner[k], shapes[k]) for k in shapes}
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("%-44s encode %.4f  decode %.4f" % (tag, e, d), flush=True)


G = 1 << 30
print("arena %#x, first boundary %#x, window %d GiB" % (base, w0, W >> 30))
run("warm-up", dict(x=0, X=0, thr=0, t=0, xh=0), dict(x=0, X=1 * G, thr=2 * G, t=3 * G, xh=3 * G + (1 << 22)))
if W >= 4 * G:
    run("all in window 0", dict(x=0, X=0, thr=0, t=0, xh=1), dict(x=0, X=1 * G, thr=2 * G, t=3 * G, xh=0))
    run("x | X thr (two windows)", dict(x=0, X=1, thr=1, t=1, xh=2), dict(x=0, X=0, thr=1 * G, t=2 * G, xh=0))
    run("x X | thr", dict(x=0, X=0, thr=1, t=1, xh=2), dict(x=0, X=1 * G, thr=0, t=2 * G, xh=0))
    run("x thr | X", dict(x=0, X=1, thr=0, t=1, xh=2), dict(x=0, X=0, thr=1 * G, t=2 * G, xh=0))
    run("x | X | thr (three windows)", dict(x=0, X=1, thr=2, t=2, xh=3), dict(x=0, X=0, thr=0, t=2 * G, xh=0))
    run("x | X | thr, same inner offsets +1G", dict(x=0, X=1, thr=2, t=2, xh=3), dict(x=G, X=G, thr=G, t=3 * G, xh=G))
    run("x | X | thr, staggered inner offsets", dict(x=0, X=1, thr=2, t=2, xh=3), dict(x=0, X=G, thr=2 * G, t=3 * G, xh=G))
    run("decode: X and xh in one window", dict(x=1, X=0, thr=1, t=1, xh=0), dict(x=0, X=0, thr=1 * G, t=2 * G, xh=1 * G))
    run("all in window 0 (again)", dict(x=0, X=0, thr=0, t=0, xh=1), dict(x=0, X=1 * G, thr=2 * G, t=3 * G, xh=0))
    run("x | X | thr (again)", dict(x=0, X=1, thr=2, t=2, xh=3), dict(x=0, X=0, thr=0, t=2 * G, xh=0))
