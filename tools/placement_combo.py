#!/usr/bin/env python
"""Encode time of the bench step on (a) eight independently allocated tensor sets, (b) three placements inside one big arena
allocated first / last (design aid: is there 'lucky' memory on this box, and does one big allocation get it?)."""
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


def med(fn, n=8):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


def arena_times(arena, tag):
    U = 1 << 21
    out = []
    for shift in (0, 3000, 7000):
        o, T = shift, {}
        for k, s in shapes.items():
            n = int(np.prod(s))
            T[k] = arena[o * U // 4: o * U // 4 + n].view(s)
            o += (n * 4 + U - 1) // U + 1
        T["x"].copy_(src)
        out.append(med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"])))
    print("%s arena %#x: " % (tag, arena.data_ptr()) + " ".join("%.3f" % v for v in out), flush=True)


first = torch.empty(24 * (1 << 30) // 4, dtype=torch.float32, device=dev)
arena_times(first, "first")
sets, rng = [], np.random.default_rng(1)
junk = []
for i in range(8):
    junk.append(torch.empty(int(rng.integers(1, 400)) * (1 << 20), dtype=torch.uint8, device=dev))
    sets.append({k: (src.clone() if k == "x" else torch.empty(s, device=dev)) for k, s in shapes.items()})
for rnd in range(2):
    print("sets: " + " ".join("%.3f" % med(lambda s=s: codec.encode_into(s["x"], s["X"], s["t"], s["thr"])) for s in sets), flush=True)
last = torch.empty(24 * (1 << 30) // 4, dtype=torch.float32, device=dev)
arena_times(last, "last ")
arena_times(first, "first")
