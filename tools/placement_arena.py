#!/usr/bin/env python
"""Random 2 MiB-granular placements of the bench tensors inside one arena; prints offsets (in 2 MiB units) and times."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
GB = int(os.environ.get("ARENA_GB", 24))
arena = torch.empty(GB * (1 << 30) // 4, dtype=torch.float32, device=dev)
U = 1 << 21
shapes = {"x": (B, K * N, C), "X": (B, K + 1, N, C), "thr": (B, K + 1, N, C), "t": (B, K + 1, 1, C), "xh": (B, (K + 2) * N, C)}
units = {k: (int(np.prod(s)) * 4 + U - 1) // U for k, s in shapes.items()}
src = torch.rand(shapes["x"], device=dev) * 2 - 1
rng = np.random.default_rng(int(os.environ.get("SEED", 3)))


def med(fn, n=8):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


total = GB * 512
for trial in range(int(os.environ.get("TRIALS", 28))):
    while True:   # non-overlapping random placement
        offs = {k: int(rng.integers(0, total - units[k])) for k in shapes}
        iv = sorted((offs[k], offs[k] + units[k]) for k in shapes)
        if all(iv[i][1] <= iv[i + 1][0] for i in range(4)):
            break
    T = {k: arena[offs[k] * U // 4: offs[k] * U // 4 + int(np.prod(s))].view(s) for k, s in shapes.items()}
    T["x"].copy_(src)
    e = med(lambda: codec.encode_into(T["x"], T["X"], T["t"], T["thr"]))
    d = med(lambda: codec.decode_into(T["X"], T["xh"]))
    print("enc %.4f dec %.4f  x %5d X %5d thr %5d xh %5d | X-x %6d thr-X %6d thr-x %6d xh-X %6d"
          % (e, d, offs["x"], offs["X"], offs["thr"], offs["xh"], offs["X"] - offs["x"], offs["thr"] - offs["X"],
             offs["thr"] - offs["x"], offs["xh"] - offs["X"]), flush=True)
