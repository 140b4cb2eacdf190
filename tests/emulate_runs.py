"""numpy replay of the run-structured masking model (audiocodec_amd/csrc/ac_psy_runs_dev.h) on the image the library's
host builder makes (``ac_testing_runs_image``): the same slot layout, list walks, entry sums and per-bin look-ups the
kernel performs, lane by lane, with the float32 roundings that matter (sums in float32, transcendentals through float64).
Test infrastructure: it lets the CPU suite hold the plan-time structure to the oracle without a GPU.
"""
import ctypes

import numpy as np

LAYOUT_FIELDS = ["words", "lw", "kb", "n4", "n16", "n64", "o4", "o16", "o64", "oz", "slot",
                 "off_S", "off_bc", "off_bd", "off_lst", "off_bw", "off_idx"]


def runs_image(lib, N, M, sample_rate, alpha, precompute=1):
    """(layout dict, image as uint32 array) or None when the tables lack the structure."""
    lay = (ctypes.c_int * 17)()
    st = lib.ac_testing_runs_image(N, M, float(sample_rate), float(alpha), precompute, None, 0, lay)
    if st != 0:
        return None
    L = dict(zip(LAYOUT_FIELDS, list(lay)))
    img = np.zeros(L["words"], dtype=np.uint32)
    st = lib.ac_testing_runs_image(N, M, float(sample_rate), float(alpha), precompute,
                                   img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint32)), L["words"], lay)
    assert st == 0
    return L, img


def granule_registers(N):
    return 1 if N <= 128 else 2 if N <= 256 else 4 if N <= 512 else 8 if N <= 1024 else 16 if N <= 2048 else 32


def thresholds(L, img, N, M, alpha, X, t, g, drown=0.0):
    """X [N] float32 (one frame, one signal), t its tonality, g the 2M-entry spreading prototype -> thr [N] float32."""
    f32 = np.float32
    fimg = img.view(np.float32)
    slot = np.zeros(L["slot"] // 4 + 2, dtype=np.float32)   # one float per (8-byte) v2f entry's first signal: index = byte offset / 8

    def at(off):          # value of the v2f at byte offset `off` (first signal)
        assert off % 8 == 0 and 0 <= off < L["slot"], off
        return slot[off // 8]

    def put(off, v):
        slot[off // 8] = f32(v)

    I = (X.astype(f32) * X.astype(f32)).astype(f32)
    for f in range(N):
        put(8 * f, I[f])
    # level sums: run c = the four values at src + 32 c (two 16-byte halves; g0 + g1 then .x + .z of the packed add)
    def level(src, dst, n):
        for c in range(n):
            a0, a1, a2, a3 = (at(src + 32 * c + 8 * k) for k in range(4))
            put(dst + 8 * c, f32(f32(a0 + a2) + f32(a1 + a3)))
    level(0, L["o4"], L["n4"])
    level(L["o4"], L["o16"], L["n16"])
    if L["n64"] > 0:
        level(L["o16"], L["o64"], L["n64"])
    assert at(L["oz"]) == 0.0
    R = granule_registers(N)
    G = np.zeros(64, dtype=np.float64)
    P = np.zeros(64, dtype=np.float32)
    for lane in range(64):
        bc = img[L["off_bc"] + 4 * lane: L["off_bc"] + 4 * lane + 4]
        w0, w1 = fimg[L["off_bc"] + 4 * lane + 1], fimg[L["off_bc"] + 4 * lane + 2]
        e0, e1 = int(bc[0]) & 0xffff, int(bc[0]) >> 16
        P0 = f32(at(e0) * w0)
        P1 = f32(at(e1) * w1)
        for k in range(L["lw"]):
            w = int(img[L["off_lst"] + 64 * k + lane])
            P0 = f32(P0 + at(w & 0xffff))
            P1 = f32(P1 + at(w >> 16))
        P[lane] = f32(P0 + P1)
    Q = np.where(np.arange(64) < M, np.maximum(P.astype(np.float64), 1e-14) ** alpha, 0.0)
    # sum_i Q_i S[i, j], S[i, j] = g[M - i + j]
    acc = np.zeros(64)
    for j in range(M):
        for i in range(M):
            acc[j] += Q[i] * float(g[M - i + j])
    for lane in range(64):
        quiet = float(fimg[L["off_bd"] + 4 * lane + 1])
        c1 = float(fimg[L["off_bd"] + 4 * lane])
        offset = (1.0 - drown) * (float(t) * c1 + 5.5)
        with np.errstate(divide="ignore"):
            y = max(np.log2(acc[lane]) - alpha * 0.33219280948873623 * offset, -46.506993328423076)
        G[lane] = max(2.0 ** (y / alpha), quiet)
    for lane in range(64):
        put(8 * lane, G[lane])
    entries = np.zeros((64, 2), dtype=np.float64)
    for lane in range(64):
        rho = float(fimg[L["off_bd"] + 4 * lane + 2])
        goff = int(img[L["off_bc"] + 4 * lane + 3])
        s = 0.0
        for k in range(L["kb"]):
            assert goff + 8 * k < 512
            s += float(at(goff + 8 * k)) * float(fimg[L["off_bw"] + 64 * k + lane])
        entries[lane, 0] = np.sqrt(max(G[lane] * rho, 1e-14))
        entries[lane, 1] = np.sqrt(max(s, 1e-14))
    for lane in range(64):
        put(512 + 16 * lane, entries[lane, 0])
        put(512 + 16 * lane + 8, entries[lane, 1])
    thr = np.zeros(N, dtype=np.float32)
    for i in range(R):
        for lane in range(64):
            q = 64 * i + lane
            if q < N // 2:
                w = int(img[L["off_idx"] + 64 * i + lane])
                thr[2 * q] = at(w & 0xffff)
                thr[2 * q + 1] = at(w >> 16)
    return thr
