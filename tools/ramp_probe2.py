"""Step time against time since the start of load (MI355X): per-step event times of the bench step over several loops
separated by idle gaps, with the GPU's own sysfs clocks sampled from a side process.  Usage: python tools/ramp_probe2.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
smi = bench.SmiSampler()
import numpy as np, torch, audiocodec_amd
dev = torch.device("cuda", 0)
N, B, K, C = 1024, 256, 468, 2
codec = audiocodec_amd.AudioCodec(48000, N)
x = bench.make_clips(torch, dev, 0, B, K)
X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
torch.cuda.synchronize()
props = torch.cuda.get_device_properties(0)
print("pci", getattr(props, "pci_bus_id", None), getattr(props, "pci_device_id", None), getattr(props, "pci_domain_id", None))
def loop(steps, label):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(steps)]
    torch.cuda.synchronize(); w0 = time.time()
    for i in range(steps):
        ev[i][0].record(); codec.encode_into(x, X, t, thr); ev[i][1].record(); codec.decode_into(X, xh); ev[i][2].record()
    torch.cuda.synchronize(); w1 = time.time()
    enc = [e[0].elapsed_time(e[1]) for e in ev]; dec = [e[1].elapsed_time(e[2]) for e in ev]
    idx = [0, 1, 2, 3, 4, 6, 8, 10, 15, 20, 25, 30, 40, 50, 75, 100, 150, 200, 300, 400, 600, 800, 999]
    print(label, "wall %.1f ms" % ((w1 - w0) * 1e3))
    print("  enc", " ".join("%d:%.3f" % (i, enc[i]) for i in idx if i < steps))
    print("  dec", " ".join("%d:%.3f" % (i, dec[i]) for i in idx if i < steps))
    return w0, w1
time.sleep(2.0)
spans = []
spans.append(loop(1000, "after 2 s idle"))
time.sleep(0.005); spans.append(loop(60, "after 5 ms idle"))
time.sleep(0.05); spans.append(loop(60, "after 50 ms idle"))
time.sleep(0.5); spans.append(loop(200, "after 500 ms idle"))
time.sleep(3.0); spans.append(loop(200, "after 3 s idle"))
# encode only / decode only after idle
time.sleep(2.0)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
for i in range(200):
    ev[i].record(); codec.encode_into(x, X, t, thr)
ev[200].record(); torch.cuda.synchronize()
e = [ev[i].elapsed_time(ev[i + 1]) for i in range(200)]
print("encode only after 2 s idle", " ".join("%d:%.3f" % (i, e[i]) for i in (0, 1, 2, 4, 8, 16, 32, 64, 128, 199)))
# a trivial copy kernel as the probe of the memory system, after idle
time.sleep(2.0)
for i in range(200):
    ev[i].record(); xh[:, :K * N].copy_(x)
ev[200].record(); torch.cuda.synchronize()
e = [ev[i].elapsed_time(ev[i + 1]) for i in range(200)]
print("torch copy 1 GB after 2 s idle", " ".join("%d:%.3f" % (i, e[i]) for i in (0, 1, 2, 4, 8, 16, 32, 64, 128, 199)))
bdf = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
st = smi.finish(spans[0][0], spans[0][1], bdf)
print(json.dumps(st))
# raw samples of the first loop for every card
rows = [json.loads(l) for l in open(smi.log)]
print("samples", len(rows), "keys", sorted(rows[0].keys()) if rows else None)
w0, w1 = spans[0]
sel = [r for r in rows if w0 - 0.02 <= r["t"] <= w0 + 0.3 and bdf in r.get("bdf", "")]
for r in sel[::3]:
    print("  t=%+.1f ms" % ((r["t"] - w0) * 1e3), {k: v for k, v in r.items() if k != "t"})
