#!/usr/bin/env python
"""How much do the kernel times depend on where the tensors sit in HBM?  One arena, the five tensors of the bench step
carved out at different offsets, encode / decode timed per placement (design aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
arena = torch.empty(int(os.environ.get("ARENA_GB", 16)) * (1 << 30) // 4, dtype=torch.float32, device=dev)
base = arena.data_ptr()
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
src = torch.rand(shapes["x"], device=dev) * 2 - 1
rng = np.random.default_rng(int(os.environ.get("SEED", 0)))


def carve(offsets):
    out = {}
    for k, shp in shapes.items():
        n = int(np.prod(shp))
        out[k] = arena[offsets[k] // 4: offsets[k] // 4 + n].view(shp)
    return out


def med(fn, n=12):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


gran = int(os.environ.get("GRAN", 256))
slot = 1 << 30     # each tensor inside its own 1 GiB slot (+ a random shift), slots in random order
print("arena base %#x (mod 2 MiB: %#x)" % (base, base % (1 << 21)))
for trial in range(int(os.environ.get("TRIALS", 14))):
    order = rng.permutation(5)
    offs = {}
    for i, k in enumerate(["x", "X", "thr", "t", "xh"]):
        shift = int(rng.integers(0, (slot - 990 * (1 << 20)) // gran)) * gran if trial else 0
        offs[k] = int(order[i] if trial else i) * slot + shift
    T = carve(offs)
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("trial %2d  encode %.4f ms  decode %.4f ms   offsets mod 1 MiB (KiB): x %4d X %4d thr %4d xh %4d   slots %s"
          % (trial, e, d, offs["x"] % (1 << 20) >> 10, offs["X"] % (1 << 20) >> 10, offs["thr"] % (1 << 20) >> 10,
             offs["xh"] % (1 << 20) >> 10, "".join(str(int(o)) for o in (order if trial else range(5)))), flush=True)
