"""Does a HIP graph shorten the gaps between the dependent launches of the streaming chain?  ac_stream_run over one
stereo clip (10 min, chunks of 256 blocks) issued directly against the same calls captured once into a graph and replayed
(design aid; the library itself issues plain launches).   python tools/stream_graph_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, audiocodec_amd
N, Kt, k = 1024, 28125, 256
dev = torch.device("cuda", 0)
codec = audiocodec_amd.AudioCodec(48000, N)
n = Kt // k
xs = torch.rand((1, n * k * N, 2), device=dev) * 2 - 1
st = codec.stream(1, 2)
def wall(fn, reps=7):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return sorted(ts)[len(ts) // 2]
for masking in (False, True):
    def direct():
        st.reset()
        st.run(xs, k, masking=masking)
    dt = wall(direct)
    print("masking=%-5s direct launches   %.2f us per chunk" % (masking, dt / n * 1e6))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        direct()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st.run(xs, k, masking=masking)     # (an even number of state swaps would be needed to replay this for real)
    dt = wall(lambda: g.replay())
    print("masking=%-5s graph replay      %.2f us per chunk" % (masking, dt / n * 1e6))
