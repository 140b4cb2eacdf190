#!/usr/bin/env python
"""Several independently allocated sets of the bench tensors in one process: does the encode time depend on which physical
memory a set landed in?  Each set is timed in three interleaved rounds (design aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
nsets = int(os.environ.get("SETS", 8))
src = torch.rand((B, K * N, C), device=dev) * 2 - 1
sets, junk = [], []
rng = np.random.default_rng(1)
for i in range(nsets):
    junk.append(torch.empty(int(rng.integers(1, 400)) * (1 << 20), dtype=torch.uint8, device=dev))   # shift the next allocations
    s = dict(x=src.clone(), X=torch.empty((B, K + 1, N, C), device=dev), t=torch.empty((B, K + 1, 1, C), device=dev),
             thr=torch.empty((B, K + 1, N, C), device=dev), xh=torch.empty((B, (K + 2) * N, C), device=dev))
    sets.append(s)


def med(fn, n=10):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


for rnd in range(3):
    row = []
    for i, s in enumerate(sets):
        e = med(lambda: codec.encode_into(s["x"], s["X"], s["t"], s["thr"]))
        d = med(lambda: codec.decode_into(s["X"], s["xh"]))
        row.append("%d: %.3f/%.3f" % (i, e, d))
    print("round %d  " % rnd + "  ".join(row), flush=True)
print("bases:", ["%#x" % (s["X"].data_ptr() >> 21) for s in sets])
