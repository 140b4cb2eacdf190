#!/usr/bin/env python
"""transform / inverse_transform of more than two channels on the LDS-FFT tier: the team form (whole rows through LDS) against
the strided channel pairs, over the sizes with an instance and C = 3, 4, 6 -- the table behind the policy of team_pays
(profiles/r4/lds_fft_team_sweep.txt)."""
import os, re, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import audiocodec_amd
src = open(os.path.join(ROOT, "audiocodec_amd", "csrc", "ac_generic.hip")).read()
blk = src[src.index("#define AC_WAVE_CT_SIZES"):]
blk = blk[:blk.index("#endif")]
sizes = sorted({int(m.group(1)) for m in re.finditer(r"AC_WAVE_CT\((\d+),", blk)})
sizes = [int(a) for a in os.environ.get("SIZES", "").split(",") if a] or [n for n in sizes if 64 <= n <= 4096]
chans = [int(a) for a in os.environ.get("CHANNELS", "3,4,6").split(",")]
def timeit(fn, n=6):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("filters_n  C   transform: team ms  strided ms  ratio    inverse: team ms  strided ms  ratio")
for N in sizes:
    m = audiocodec_amd.MDCTransformer(N)
    for C in chans:
        B = max(1, 256 // C)
        K = max(4, 480000 // N)
        x = torch.rand((B, K * N, C), device="cuda") * 2 - 1
        X = m.transform(x)
        r = []
        for mode in ("2", "1"):
            os.environ["AC_LDS_WAVE_NOTEAM"] = mode
            r.append((timeit(lambda: m.transform(x)), timeit(lambda: m.inverse_transform(X))))
        del os.environ["AC_LDS_WAVE_NOTEAM"]
        print("%8d  %d   %8.3f  %8.3f  %6.3f    %8.3f  %8.3f  %6.3f" % (N, C, r[0][0], r[1][0], r[0][0] / r[1][0], r[0][1], r[1][1], r[0][1] / r[1][1]), flush=True)
        del x, X
