"""The bench workload (B = 256 stereo clips x 468 blocks, N = 1024) as a pipelined round trip: encode (with the masking
model) and decode through the streaming API in chunks of k blocks -- the analysis of chunk i + 1 shares a launch with the
synthesis of chunk i, and the synthesis finds the chunk's spectrum in the 256 MiB Infinity Cache.  Every output of the
one-shot calls is produced (X, t, thr, PCM).   python tools/pipeline_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audiocodec_amd
N, B, K, C = 1024, int(os.environ.get("B", 256)), 468, 2
dev = torch.device("cuda", 0)
codec = audiocodec_amd.AudioCodec(48000, N)
x = torch.rand((B, K * N, C), device=dev) * 2 - 1
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
def wall(fn, reps=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
def one_shot():
    for _ in range(10):
        codec.encode_into(x, X, t, thr); codec.decode_into(X, xh)
dt = wall(one_shot) / 10
print("one-shot encode + decode launches        %.3f ms per batch  %.1f M frames/s" % (dt * 1e3, B * C * K / dt / 1e6))
st = codec.stream(B, C)
for k in (26, 39, 52, 78, 117, 234):
    chunks = [x[:, i * k * N:(i + 1) * k * N].contiguous() for i in range(K // k)]
    def run():
        st.reset()
        st.run(chunks, k, masking=True)
    dt = wall(run)
    print("stream pipeline, chunks of %3d blocks     %.3f ms per batch  %.1f M frames/s  (spectrum per chunk %.0f MB)"
          % (k, dt * 1e3, B * C * K / dt / 1e6, B * k * N * C * 4 / 1e6))
