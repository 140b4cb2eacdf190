#!/usr/bin/env python
"""Where inside a 44 GiB arena does the encode kernel speed up?  x ends at arena offset b, X starts at b, thr follows X
(b scanned); then x / X / thr in three different 16 GiB stretches."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
nbytes = {k: int(np.prod(s)) * 4 for k, s in shapes.items()}
src = torch.rand(shapes["x"], device=dev) * 2 - 1
GB = 1 << 30
arena = torch.empty(44 * GB // 4, dtype=torch.float32, device=dev)
U = 1 << 21


def med(fn, n=6):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def at(off, k):
    off = off // U * U
    return arena[off // 4: off // 4 + nbytes[k] // 4].view(shapes[k])


def run(tag, ox, oX, othr, oxh):
    T = dict(x=at(ox, "x"), X=at(oX, "X"), thr=at(othr, "thr"), t=at(43 * GB, "t"), xh=at(oxh, "xh"))
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("%-46s encode %.4f  decode %.4f" % (tag, e, d), flush=True)


print("arena %#x" % arena.data_ptr())
run("warm-up", 0, 1 * GB, 2 * GB, 3 * GB)
for b in np.arange(13.0, 19.01, 0.5):
    bb = int(b * GB)
    run("x ends / X starts at %.2f GiB, thr after X" % b, bb - nbytes["x"], bb, bb + nbytes["X"] + U, bb + 2 * nbytes["X"] + 2 * U)
for b in (29.5, 30.5, 31.0, 31.5, 32.0, 32.5, 33.5):
    bb = int(b * GB)
    run("x ends / X starts at %.2f GiB, thr after X" % b, bb - nbytes["x"], bb, bb + nbytes["X"] + U, bb + 2 * nbytes["X"] + 2 * U)
run("x @2, X @18, thr @34 (three stretches)", 2 * GB, 18 * GB, 34 * GB, 36 * GB)
run("x @2, X @18, thr @20", 2 * GB, 18 * GB, 20 * GB, 36 * GB)
run("x @2, X @4, thr @18", 2 * GB, 4 * GB, 18 * GB, 36 * GB)
run("x @18, X @2, thr @34", 18 * GB, 2 * GB, 34 * GB, 36 * GB)
run("decode probe: X @2, xh @18", 6 * GB, 2 * GB, 4 * GB, 18 * GB)
run("decode probe: X @2, xh @4", 6 * GB, 2 * GB, 8 * GB, 4 * GB)
