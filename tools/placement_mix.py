#!/usr/bin/env python
"""Which tensor's allocation decides the encode time?  Independently allocated sets, then mixes of them (design aid)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
codec = audiocodec_amd.AudioCodec(48000, N)
nsets = 8
src = torch.rand((B, K * N, C), device=dev) * 2 - 1
sets, junk = [], []
rng = np.random.default_rng(1)
for i in range(nsets):
    junk.append(torch.empty(int(rng.integers(1, 400)) * (1 << 20), dtype=torch.uint8, device=dev))
    sets.append(dict(x=src.clone(), X=torch.empty((B, K + 1, N, C), device=dev), t=torch.empty((B, K + 1, 1, C), device=dev),
                     thr=torch.empty((B, K + 1, N, C), device=dev), xh=torch.empty((B, (K + 2) * N, C), device=dev)))


def med(fn, n=8):
    fn(); fn()
    ts = []
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    return float(np.median(ts))


enc = lambda a, b, c: med(lambda: codec.encode_into(sets[a]["x"], sets[b]["X"], sets[b]["t"], sets[c]["thr"]))
base = [enc(i, i, i) for i in range(nsets)]
base = [enc(i, i, i) for i in range(nsets)]
print("own sets      ", " ".join("%.3f" % v for v in base))
best, worst = int(np.argmin(base)), int(np.argmax(base))
print("best set %d, worst set %d" % (best, worst))
for name, f in (("x from j  ", lambda j: enc(j, best, best)), ("X from j  ", lambda j: enc(best, j, best)), ("thr from j", lambda j: enc(best, best, j))):
    print("best set with %s" % name, " ".join("%.3f" % f(j) for j in range(nsets)))
for s in sets:
    print("x %#x  X %#x  thr %#x" % (s["x"].data_ptr(), s["X"].data_ptr(), s["thr"].data_ptr()))
# write-only and read-only probes per tensor
for i, s in enumerate(sets):
    w = med(lambda: s["X"].fill_(1.0)); w2 = med(lambda: s["thr"].fill_(1.0)); r = med(lambda: s["x"].sum())
    print("set %d  fill X %.3f  fill thr %.3f  sum x %.3f" % (i, w, w2, r))
