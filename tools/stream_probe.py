"""Streaming configs[4] under rocprofv3 --kernel-trace: per-kernel durations and the gaps between them.
    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/sp -- python3 tools/stream_probe.py
    python3 tools/stream_probe.py --summarise gpurun_out/sp        (afterwards: durations and gaps per kernel kind)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
    import csv, glob, collections
    rows = []
    for f in glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    dur, gap = collections.defaultdict(list), collections.defaultdict(list)
    for i, (s, e, n) in enumerate(rows):
        name = n.replace("void ac::(anonymous namespace)::", "").split("(")[0]
        if "ac::" not in n:
            continue
        dur[name].append(e - s)
        if i and "ac::" in rows[i - 1][2]:
            gap[name].append(s - rows[i - 1][1])
    for name, d in dur.items():
        d.sort()
        g = sorted(gap[name]) or [0]
        print("%-48s n %6d  duration median %6.2f us (p10 %5.2f p90 %5.2f)  gap before it median %5.2f us"
              % (name, len(d), d[len(d) // 2] / 1e3, d[len(d) // 10] / 1e3, d[len(d) * 9 // 10] / 1e3, g[len(g) // 2] / 1e3))
    sys.exit(0)

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
    print("chain (encode + inverse): host issue %.3f ms, total %.3f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
n = len(chunks)
for masking in (False, True):
    for rep in range(3):
        st.reset(); torch.cuda.synchronize(); t0 = time.perf_counter()
        st.run(xs[:, :n * k * N], k, masking=masking)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("ac_stream_run masking=%s: host issue %.3f ms, total %.3f ms = %.2f us per chunk"
              % (masking, (t1 - t0) * 1e3, (t2 - t0) * 1e3, (t2 - t0) * 1e6 / n))
