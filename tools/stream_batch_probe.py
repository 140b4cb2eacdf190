"""Streaming with many concurrent streams: ac_stream_run on B streams, chunks of k blocks (design aid for the size limit
of the duplex launch, AC_DUPLEX_MAX_TASKS).   B=64 python tools/stream_batch_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audiocodec_amd
N, k = 1024, 256
B, n = int(os.environ.get("B", 64)), int(os.environ.get("CHUNKS", 24))
dev = torch.device("cuda", 0)
codec = audiocodec_amd.AudioCodec(48000, N)
chunks = [torch.rand((B, k * N, 2), device=dev) * 2 - 1 for _ in range(n)]
st = codec.stream(B, 2)
for masking in (False, True):
    ts = []
    for rep in range(5):
        st.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st.run(chunks, k, masking=masking)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    dt = sorted(ts)[len(ts) // 2]
    print("B %3d  masking=%-5s  %.1f us per chunk  %.1f M frames/s  (AC_DUPLEX_MAX_TASKS=%s)"
          % (B, masking, dt / n * 1e6, B * 2 * k * n / dt / 1e6, os.environ.get("AC_DUPLEX_MAX_TASKS", "default")))
