#!/usr/bin/env python
"""Per-entry-point timings on the bench workload (design aid; not the contract bench)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N = int(os.environ.get("N", 1024))
B, K, C = int(os.environ.get("B", 256)), int(os.environ.get("K", 468 * 1024 // N)), int(os.environ.get("C", 2))
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
codec = audiocodec_amd.AudioCodec(48000, N)
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
frames = B * C * K

def timeit(fn, n=20):
    # (an MI355X that has idled for a few ms runs the first ~30 ms of any load up to 25 % slower, DESIGN_LOG.md 5a: keep it
    # busy for ~100 ms before the timed launches, as bench.py does)
    t_end = time.perf_counter() + 0.1
    while time.perf_counter() < t_end:
        for _ in range(8): fn()
        torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

rows = [
    ("encode_fused", lambda: codec.encode_into(x, X, t, thr), 12 * N + 4),
    ("transform", lambda: codec.mdct.transform(x), 8 * N),
    ("inverse", lambda: codec.decode_into(X, xh), 8 * N),
    ("tonality", lambda: codec.psy.tonality(X), 4 * N + 4),
    ("threshold", lambda: codec.psy.global_masking_threshold(X, t), 8 * N + 4),
    ("torch copy X (1R:1W)", lambda: thr.copy_(X), 8 * N),
    ("torch fill X (0R:1W)", lambda: thr.fill_(1.0), 4 * N),
    ("torch sum X (1R:0W)", lambda: X.sum(), 4 * N),
]
if codec.mdct.is_fast() and N >= 1024:
    pcm = (x * 32767).to(torch.int16)
    pcm_out = torch.empty((B, (K + 2) * N, C), device=dev, dtype=torch.int16)
    rows += [("encode_fused pcm16", lambda: codec.encode_into(pcm, X, t, thr), 10 * N + 4),
             ("inverse pcm16", lambda: codec.decode_into(X, pcm_out), 6 * N)]
if codec.mdct.is_fast() and C <= 2 and N >= 1024:
    cb = audiocodec_amd.AudioCodec(48000, N, compute_dtype=torch.bfloat16)
    xb = x.to(torch.bfloat16); Xb = torch.empty_like(X, dtype=torch.bfloat16); tb = torch.empty_like(t, dtype=torch.bfloat16)
    thrb = torch.empty_like(Xb); xhb = torch.empty_like(xh, dtype=torch.bfloat16)
    cb.encode_into(xb, Xb, tb, thrb)
    rows += [("encode_fused bf16", lambda: cb.encode_into(xb, Xb, tb, thrb), 6 * N + 2),
             ("inverse bf16", lambda: cb.decode_into(Xb, xhb), 4 * N),
             ("threshold bf16", lambda: cb.psy.global_masking_threshold(Xb, tb), 4 * N + 2)]
for name, fn, bpf in rows:
    ms = timeit(fn)
    print("%-24s %8.3f ms   %7.0f GB/s (algorithmic %d B/frame)" % (name, ms, bpf * frames / ms / 1e6, bpf))
