#!/usr/bin/env python
"""Sequential steps (encode, decode on one stream) against a two-stream pipeline in which decode of step i runs beside
encode of step i + 1 on double-buffered outputs (design aid)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import audiocodec_amd

N, B, K, C = 1024, 256, 468, 2
dev = torch.device("cuda")
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
bufs = [dict(X=torch.empty((B, K + 1, N, C), device=dev), t=torch.empty((B, K + 1, 1, C), device=dev),
             thr=torch.empty((B, K + 1, N, C), device=dev), xh=torch.empty((B, (K + 2) * N, C), device=dev)) for _ in range(2)]
codec = audiocodec_amd.AudioCodec(48000, N)
steps = int(os.environ.get("STEPS", 40))


def sequential():
    for i in range(steps):
        b = bufs[i & 1]
        codec.encode_into(x, b["X"], b["t"], b["thr"])
        codec.decode_into(b["X"], b["xh"])


sa, sb = torch.cuda.Stream(), torch.cuda.Stream()


def pipelined():
    enc_done = [None, None]
    dec_done = [None, None]
    for i in range(steps):
        k = i & 1
        b = bufs[k]
        with torch.cuda.stream(sa):
            if dec_done[k] is not None:
                sa.wait_event(dec_done[k])          # the decode that read this buffer set two steps ago
            codec.encode_into(x, b["X"], b["t"], b["thr"])
            enc_done[k] = torch.cuda.Event(); enc_done[k].record(sa)
        with torch.cuda.stream(sb):
            sb.wait_event(enc_done[k])
            codec.decode_into(b["X"], b["xh"])
            dec_done[k] = torch.cuda.Event(); dec_done[k].record(sb)
    torch.cuda.current_stream().wait_stream(sa)
    torch.cuda.current_stream().wait_stream(sb)


for name, fn in (("sequential", sequential), ("pipelined", pipelined), ("sequential", sequential), ("pipelined", pipelined)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%-11s %.3f ms/step  %.1f M frames/s" % (name, dt / steps * 1e3, B * C * K * steps / dt / 1e6), flush=True)
err = float((bufs[0]["xh"][:, N:-N] - x).abs().max())
print("round trip", err)
