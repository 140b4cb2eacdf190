"""16-bit PCM at the boundary of the several-frames-per-wave kernels: ac_mdct_forward_pcm16 / ac_mdct_inverse_pcm16 and the
encode (transform + masking model) on B = 256 stereo clips of 10 s.   python tools/pcm16_short_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audiocodec_amd
from audiocodec_amd import _lib, _host
lib = _lib.load()
def timeit(fn, n=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for N in (1024, 512, 256, 128, 64):
    B, C = 256, 2
    K = 468 * 1024 // N
    pcm = torch.randint(-32768, 32768, (B, K * N, C), device="cuda", dtype=torch.int16)
    codec = audiocodec_amd.AudioCodec(48000, N)
    plan = codec.mdct._plan(pcm.device)
    X = torch.empty(B, K + 1, N, C, device="cuda")
    out = torch.empty(B, (K + 2) * N, C, device="cuda", dtype=torch.int16)
    sp = _host.stream_ptr(pcm.device)
    fwd = lambda: _lib.check(lib.ac_mdct_forward_pcm16(plan, _host.ptr(pcm), _host.ptr(X), B, K, C, sp))
    inv = lambda: codec.decode_into(X, out)
    enc = lambda: codec.encode(pcm)
    tf, ti, te = timeit(fwd), timeit(inv), timeit(enc)
    fr = B * C * K
    print("N %4d  transform %.3f ms %5.0f GB/s   inverse %.3f ms %5.0f GB/s (6 N B/frame)   encode %.3f ms %5.0f GB/s (10 N + 4 B/frame)"
          % (N, tf, 6 * N * fr / tf / 1e6, ti, 6 * N * fr / ti / 1e6, te, (10 * N + 4) * fr / te / 1e6))
