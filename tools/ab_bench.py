"""A/B of library variants on the SAME tensors in one process (placement varies from process to process by more than most
kernel changes; see DESIGN_LOG.md 9a).  Usage (GPU box):
    python tools/ab_bench.py [--n 1024] [--clips 256] [--blocks 468] [--rounds 5] name=path.so [name=path.so ...]
Each variant: its own plans (ac_mdct_plan_create / ac_psy_plan_create), fused encode + decode on shared buffers, device
settled first, `rounds` interleaved rounds of 20 steps, median per variant; outputs of every variant compared with the
first one's (max relative deviation of X / thr / PCM)."""
import argparse, ctypes, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from audiocodec_amd import _lib as L

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=1024)
ap.add_argument("--clips", type=int, default=256)
ap.add_argument("--blocks", type=int, default=468)
ap.add_argument("--channels", type=int, default=2)
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--spread", type=int, default=-1, help="AC_SPREAD_* for ac_psy_plan_create_ex; -1 = library default")
ap.add_argument("--sets", type=int, default=1, help="independent sets of buffers (different places in VRAM)")
ap.add_argument("--workspace", action="store_true", help="one more set of buffers placed by audiocodec_amd.Workspace")
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
N, B, K, C = a.n, a.clips, a.blocks, a.channels
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)

def load(path):
    lib = ctypes.CDLL(os.path.abspath(path))
    for name, (restype, argtypes) in L.PROTOTYPES.items():
        if hasattr(lib, name):
            fn = getattr(lib, name); fn.restype = restype; fn.argtypes = argtypes
    mp, pp = ctypes.c_void_p(), ctypes.c_void_p()
    assert lib.ac_mdct_plan_create(N, 0, 0, ctypes.byref(mp)) == 0, lib.ac_last_error()
    if a.spread >= 0:
        assert lib.ac_psy_plan_create_ex(N, 64, 48000.0, 0.6, 0, a.spread, ctypes.byref(pp)) == 0, lib.ac_last_error()
    else:
        assert lib.ac_psy_plan_create(N, 64, 48000.0, 0.6, 0, ctypes.byref(pp)) == 0, lib.ac_last_error()
    return lib, mp, pp

variants = []
for v in a.variants:
    name, path = v.split("=", 1)
    variants.append((name,) + load(path))
sets = []
g = torch.Generator(device=dev).manual_seed(1)
for s in range(a.sets):
    x = torch.empty((B, K * N, C), device=dev).uniform_(-1, 1, generator=g)
    X = torch.empty((B, K + 1, N, C), device=dev); t = torch.empty((B, K + 1, 1, C), device=dev)
    thr = torch.empty_like(X); xh = torch.empty((B, (K + 2) * N, C), device=dev)
    sets.append((x, X, t, thr, xh))
if a.workspace:
    import audiocodec_amd
    ws = audiocodec_amd.Workspace(audiocodec_amd.AudioCodec(48000, N), B, K, C, device=dev)
    ws.x.copy_(sets[0][0])
    sets.append((ws.x, ws.X, ws.t, ws.thr, ws.xh))
    print("workspace set %d: %s" % (len(sets) - 1, ws.report))
stream = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
P = lambda t_: ctypes.c_void_p(t_.data_ptr())
def enc(v, bufs):
    _, lib, mp, pp = v; x, X, t, thr, xh = bufs
    st = lib.ac_encode_fused(mp, pp, P(x), P(X), P(t), P(thr), 0.0, B, K, C, stream); assert st == 0, lib.ac_last_error()
def dec(v, bufs):
    _, lib, mp, pp = v; x, X, t, thr, xh = bufs
    st = lib.ac_mdct_inverse(mp, P(X), P(xh), B, K + 1, C, stream); assert st == 0, lib.ac_last_error()
# results of every variant against the first
ref = None
for v in variants:
    enc(v, sets[0]); dec(v, sets[0]); torch.cuda.synchronize()
    cur = [sets[0][i].clone() for i in (1, 3, 4)]
    if ref is None:
        ref = cur
        err = float((cur[2][:, N:-N] - sets[0][0]).abs().max()) if K > 0 else 0.0
        print("%-12s round trip max abs err %.3g" % (v[0], err))
    else:
        dX = float((cur[0] - ref[0]).abs().max() / ref[0].abs().max())
        dthr = float(((cur[1] - ref[1]).abs() / ref[1]).max())
        dx = float((cur[2] - ref[2]).abs().max())
        print("%-12s vs %-12s  X %.3g (of peak)  thr %.3g (rel)  pcm %.3g (abs)" % (v[0], variants[0][0], dX, dthr, dx))
    del cur
del ref
# settle
t0 = time.perf_counter()
while time.perf_counter() - t0 < 0.15:
    enc(variants[0], sets[0]); dec(variants[0], sets[0]); torch.cuda.synchronize()
res = {(v[0], s): ([], []) for v in variants for s in range(len(sets))}
for r in range(a.rounds):
    for si, bufs in enumerate(sets):
        for v in variants:
            for _ in range(3):
                enc(v, bufs); dec(v, bufs)
            ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(a.steps)]
            for i in range(a.steps):
                ev[i][0].record(); enc(v, bufs); ev[i][1].record(); dec(v, bufs); ev[i][2].record()
            torch.cuda.synchronize()
            res[(v[0], si)][0].append(float(np.mean([e[0].elapsed_time(e[1]) for e in ev])))
            res[(v[0], si)][1].append(float(np.mean([e[1].elapsed_time(e[2]) for e in ev])))
fr = B * C * K
for si in range(len(sets)):
    for v in variants:
        e, d = res[(v[0], si)]
        print("set %d  %-12s encode %.4f ms (min %.4f)  decode %.4f ms (min %.4f)  step %.4f ms = %.1f M frames/s"
              % (si, v[0], np.median(e), min(e), np.median(d), min(d), np.median(e) + np.median(d),
                 fr / ((np.median(e) + np.median(d)) * 1e-3) / 1e6))
