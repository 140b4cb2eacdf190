"""Streaming configs[4] under rocprofv3 --kernel-trace: per-kernel durations and the gaps between them.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp -- python tools/stream_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audiocodec_amd
N, Kt, k = 1024, 28125, 256
dev = torch.device("cuda", 0)
codec = audiocodec_amd.AudioCodec(48000, N)
xs = torch.rand((1, Kt * N, 2), device=dev) * 2 - 1
st = codec.stream(1, 2)
b = (torch.empty((1, k, N, 2), device=dev), torch.empty((1, k, 1, 2), device=dev), torch.empty((1, k, N, 2), device=dev))
xo = torch.empty((1, k * N, 2), device=dev)
chunks = [xs[:, p * N:(p + k) * N] for p in range(0, Kt - k + 1, k)]
for rep in range(3):
    st.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for xc in chunks:
        st.encode_chunk(xc, out=b); st.inverse_chunk(b[0], out=xo)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print("chain: host issue %.3f ms, total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
