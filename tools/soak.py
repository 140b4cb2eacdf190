"""Randomised soak of the wave-level / LDS-FFT / masking tiers against the O(N^2) kernels (AC_TESTING=1): shapes, sizes,
windows, channel counts, streaming chunkings, 16-bit PCM.  Not part of the test suite (minutes, not seconds); prints the
first mismatch and exits non-zero.   AC_TESTING=1 python tools/soak.py [seconds] [seed]   (BIG=1: fewer, larger cases)"""
import os, sys, time
os.environ.setdefault("AC_TESTING", "1")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch, audiocodec_amd
from audiocodec_amd import _lib
lib = _lib.load()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
BIG = os.environ.get("BIG", "0") == "1"   # BIG=1: fewer, larger cases
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
SIZES = [64, 128, 256, 512, 1024, 2048, 960, 480, 240, 120, 576, 192, 96, 48, 32, 16, 1536, 3072, 4096, 24, 1000, 1920, 2880, 6144, 8192, 30, 90,
         500, 600, 720, 800, 1080, 2160, 3000, 7680, 108, 1152, 2304, 540, 576, 3240, 3600]
TOL, LSB = 1e-4, 1.0 / 32768
t_end, cases = time.time() + budget, 0
t_tick = time.time() + 60.0
def fail(msg):
    print("MISMATCH:", msg); sys.exit(1)
while time.time() < t_end:
    N = int(rng.choice(SIZES)); wt = str(rng.choice(["vorbis", "sine", "vorbis", "sine", "rect"]))
    pre = "float32" if rng.integers(0, 5) == 0 else "float64"     # (one case in five: float32-precomputed constants)
    B, C = int(rng.integers(1, 5)), int(rng.choice([1, 2, 3, 1, 2, 3, 4, 5, 6, 7]))
    # the launch policies half of the time forced: every LDS-FFT instance with the fused encode, every shape with a team form
    os.environ["AC_LDS_WAVE_NOFUSE"] = str(rng.choice(["0", "2"]))
    os.environ["AC_LDS_WAVE_NOTEAM"] = str(rng.choice(["0", "2"]))
    K = int(rng.integers(0, max(2, min(60, 40000 // N))))
    if BIG:   # launches of many workgroups: frames per wave > 1, strips, persistent rounds (the O(N^2) check bounds the size)
        N = int(rng.choice([64, 128, 256, 512, 1024, 2048, 960, 480]))
        B, C = int(rng.integers(1, 48)), int(rng.integers(1, 3))
        K = int(rng.integers(1, max(2, (6000 if N >= 1024 else 20000) // (B * C))))
    M = int(rng.choice([64, 48, 20])) if N >= 128 else int(rng.choice([16, 8]))
    drown = float(rng.choice([0.0, 0.3, 1.0]))
    tag = "N=%d %s pre=%s B=%d K=%d C=%d M=%d fuse=%s team=%s" % (N, wt, pre, B, K, C, M, os.environ["AC_LDS_WAVE_NOFUSE"], os.environ["AC_LDS_WAVE_NOTEAM"])
    x = torch.from_numpy(rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)).cuda()
    codec = audiocodec_amd.AudioCodec(48000, N, bark_bands_n=M, window_type=wt, precompute_dtype=pre)
    lib.ac_set_force_generic(0)
    X, t, thr = codec.encode(x, drown=drown)
    xh = codec.decode(X)
    lib.ac_set_force_generic(1)
    try:
        Xg = codec.mdct.transform(x); tg = codec.psy.tonality(X); thrg = codec.psy.global_masking_threshold(X, t, drown)
        xg = codec.decode(X)
    finally:
        lib.ac_set_force_generic(0)
    peak = Xg.abs().amax(dim=2, keepdim=True).clamp_min(1e-20)
    if float(((X - Xg).abs() / peak).max()) > TOL: fail(tag + " transform")
    if float(((t - tg).abs() - 1e-4 * tg.abs()).max()) > 1e-6: fail(tag + " tonality")
    if float(((thr - thrg).abs() / thrg).max()) > TOL: fail(tag + " threshold")
    if float((xh - xg).abs().max()) > 2e-6 * max(1.0, float(xg.abs().max())): fail(tag + " inverse")
    if K > 0 and wt != "rect" and float((xh[:, N:-N] - x).abs().max()) > LSB: fail(tag + " round trip")
    # streaming in random chunks = one shot, bit for bit (float32 path of the same tier)
    if K >= 2:
        st = codec.stream(B, C)
        cuts = sorted(set([0, K] + [int(v) for v in rng.integers(1, K, size=int(rng.integers(1, 4)))]))
        outs = [st.transform_chunk(x[:, a * N:b * N].contiguous()) for a, b in zip(cuts[:-1], cuts[1:])]
        outs.append(st.transform_chunk(torch.zeros(B, N, C, device="cuda")))
        Xs = torch.cat(outs, dim=1)
        if not torch.equal(Xs, codec.mdct.transform(x)): fail(tag + " streaming transform, cuts %s" % cuts)
        st.reset()
        outs = [st.inverse_chunk(X[:, a:b].contiguous()) for a, b in zip(cuts[:-1] , cuts[1:])] + [st.inverse_chunk(X[:, K:].contiguous())]
        if float((torch.cat(outs, dim=1) - xh[:, :(K + 1) * N]).abs().max()) > 1e-6 * max(1.0, float(xh.abs().max())): fail(tag + " streaming inverse")
        st.close()
    # ac_stream_run (duplex launches where they apply) = the chunk-by-chunk calls, bit for bit; once more as a replayed graph
    if K >= 2 and C <= 2 and N in (1024, 2048) and cases % 3 == 0:
        k = int(rng.integers(1, K + 1)); n = K // k
        masking = bool(rng.integers(0, 2))
        chunks = [x[:, i * k * N:(i + 1) * k * N].contiguous() for i in range(n)]
        sa, sb = codec.stream(B, C), codec.stream(B, C)
        Xl, tl, thl, xl = sa.run(chunks, k, masking=masking, drown=drown)
        for i, c in enumerate(chunks):
            if masking:
                Xc, tc, thc = sb.encode_chunk(c, drown=drown)
                if not (torch.equal(tc, tl[i]) and torch.equal(thc, thl[i])): fail(tag + " stream run masking, k=%d chunk %d" % (k, i))
            else:
                Xc = sb.transform_chunk(c)
            if not torch.equal(Xc, Xl[i]): fail(tag + " stream run spectrum, k=%d chunk %d" % (k, i))
            if not torch.equal(sb.inverse_chunk(Xc), xl[i]): fail(tag + " stream run synthesis, k=%d chunk %d" % (k, i))
        sa.reset()
        g1 = sa.run(chunks, k, masking=masking, drown=drown, graph=True)
        if not all(torch.equal(a, b) for a, b in zip(g1[0], Xl)) or not all(torch.equal(a, b) for a, b in zip(g1[3], xl)):
            fail(tag + " stream run as a graph, k=%d" % k)
        sa.close(), sb.close()
    # 16-bit PCM where the wave-level kernels take it
    if codec.mdct.is_fast() and (N >= 1024 or C <= 2) and K > 0 and wt != "rect" and pre == "float64":
        pcm = torch.from_numpy(rng.integers(-32768, 32768, (B, K * N, C)).astype(np.int16)).cuda()
        Xp, Xq = codec.encode(pcm)[0], codec.encode(pcm.float() / 32768.0)[0]
        # (more than two channels: PCM on the wave-level kernels' strided form, float32 on the channel-pair instances)
        if not (torch.equal(Xp, Xq) if C <= 2 else float((Xp - Xq).abs().max()) <= 2e-6 * float(Xq.abs().max())): fail(tag + " pcm16 encode")
        if not torch.equal(codec.decode(Xp, pcm16=True)[:, N:-N], pcm): fail(tag + " pcm16 round trip")
    cases += 1
    if time.time() > t_tick:      # (a run that says nothing for minutes is taken to be hung)
        print("... %d cases" % cases, flush=True)
        t_tick = time.time() + 60.0
print("soak: %d random cases in %.0f s, no mismatch" % (cases, budget))
