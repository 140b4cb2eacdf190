#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ from the reference's own source.

TEST INFRASTRUCTURE ONLY.  Runs only in the build container, where
/root/reference is mounted; it is a no-op elsewhere (the GPU box has no
reference).  The reference modules are imported *unmodified* with the numpy
stand-in for TensorFlow (oracle/tf_standin) ahead on sys.path, evaluated in
float64 ("ref64": the mathematical ground truth of the reference's formulas)
and in float32 ("ref32": the reference's own rounding envelope).  Only data --
inputs and outputs -- is written; no reference source travels.

    python oracle/gen_golden.py            # rewrites tests/golden/*.npz
"""

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def _import_reference():
    sys.dont_write_bytecode = True
    sys.path.insert(0, os.path.join(HERE, "tf_standin"))
    sys.path.insert(0, REF)
    import tensorflow as tf  # the stand-in
    from audiocodec import psychoacoustic
    from audiocodec.mdctransformer import MDCTransformer
    return tf, MDCTransformer, psychoacoustic.PsychoacousticModel


def _sine_f32(amplitude, frequency, sample_rate, duration_sec):
    """tests/test_mdctransformer.py:11-15 with TF's float32 semantics (see oracle.sine_wav)."""
    t = np.arange(0, sample_rate * duration_sec, dtype=np.float32)
    phase = (np.float32(2.0 * np.pi * frequency) * t) / np.float32(sample_rate)
    return (np.float32(amplitude) * np.sin(phase, dtype=np.float32)).reshape(1, -1, 1)


def _triplets(m):
    idx = np.nonzero(m)
    return np.stack([idx[0], idx[1]], axis=1).astype(np.int32), m[idx]


def main():
    if not os.path.isdir(REF):
        print("gen_golden: %s not present -- nothing to do" % REF)
        return 0
    tf, MDCT, Psy = _import_reference()
    os.makedirs(OUT, exist_ok=True)
    f32, f64 = np.float32, np.float64

    def mdct(N, wt, dt):
        return MDCT(N, window_type=wt, compute_dtype=np.dtype(dt), precompute_dtype=tf.float64)

    def psy(sr, N, M, dt, alpha=0.6):
        return Psy(sr, filter_bands_n=N, bark_bands_n=M, alpha=alpha, compute_dtype=np.dtype(dt),
                   precompute_dtype=tf.float64)

    def save(name, **arrs):
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrs)
        print("  %-28s %7.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))

    # 1. known-answer case of test_mdct_calculation (tests/test_mdctransformer.py:39-54)
    N = 64
    x = _sine_f32(0.8, 4, 64, 4.0)
    x = x[:, : N * (x.shape[1] // N)]
    known = np.array([-0.000412722176, 0.000430465181, 0.000789350364, -0.000867388735, -0.00275337417,
                      0.0132110268, 0.0193885863, 0.156005412, -0.233544752, -0.0129148215], dtype=f64)
    save("mdct_n64_sine", x=x,
         X_ref64=mdct(N, "vorbis", f64).transform(x.astype(f64)),
         X_ref32=mdct(N, "vorbis", f32).transform(x),
         known_answer_frame1_first10=known,
         note=np.array("known_answer = literal values of tests/test_mdctransformer.py:51-52 (real-TF provenance, "
                       "float32-precompute revision: expect ~6e-7 abs vs the current fp64-precompute code)"))

    # 2. round trip of test_inverse_identity (tests/test_mdctransformer.py:19-37)
    N = 256
    x = _sine_f32(0.8, 880, 16000, 1.0)
    x = x[:, : N * (x.shape[1] // N)]
    m64, m32 = mdct(N, "vorbis", f64), mdct(N, "vorbis", f32)
    X64 = m64.transform(x.astype(f64))
    save("mdct_n256_roundtrip", x=x, X_ref64=X64, X_ref32=m32.transform(x),
         xhat_ref64=m64.inverse_transform(X64), xhat_ref32=m32.inverse_transform(m32.transform(x)))

    # 3. random stereo blocks, N = 1024 / 2048 / small odd-ish sizes, all three windows
    rng = np.random.default_rng(1)
    for N, K, wts in ((1024, 4, ("vorbis", "sine")), (2048, 2, ("vorbis",)), (16, 5, ("vorbis", "sine", "rect")),
                      (12, 3, ("vorbis",))):
        x = rng.uniform(-1, 1, (1, K * N, 2)).astype(f32)
        for wt in wts:
            m64, m32 = mdct(N, wt, f64), mdct(N, wt, f32)
            X64 = m64.transform(x.astype(f64))
            X32 = m32.transform(x)
            save("mdct_n%d_rand_%s" % (N, wt), x=x, X_ref64=X64, X_ref32=X32,
                 xhat_ref64=m64.inverse_transform(X64), xhat_ref32=m32.inverse_transform(X32))

    # dense polyphase matrices at a printable size (mdctransformer.py:37-56 shows N = 8)
    m64 = mdct(8, "vorbis", f64)
    save("mdct_n8_H", H=np.asarray(m64.H), H_inv=np.asarray(m64.H_inv))

    # 4. psychoacoustic constant tables
    for sr, N, M in ((48000, 1024, 64), (48000, 2048, 64), (32768, 64, 64), (44100, 256, 48)):
        p = psy(sr, N, M, f64)
        wi, wv = _triplets(np.asarray(p.W))
        vi, vv = _triplets(np.asarray(p.W_inv))
        save("psy_%d_%d_%d_tables" % (sr, N, M), W_idx=wi, W_val=wv, W_inv_idx=vi, W_inv_val=vv,
             S=np.asarray(p.spreading_matrix), quiet=np.asarray(p.quiet_threshold_intensity).reshape(-1),
             max_bark=f64(p.max_bark), bark_band_width=f64(p.bark_band_width), dB_MIN=f64(p._dB_MIN),
             dB_MIN_f32=f32(psy(sr, N, M, f32)._dB_MIN))

    # 5. tonality + masking threshold cases at 48 kHz / 1024 / 64
    g = np.load(os.path.join(OUT, "mdct_n1024_rand_vorbis.npz"))
    Xr = g["X_ref32"][:, 1:4]                                   # [1,3,1024,2] float32, interior frames
    Xz = np.zeros((1, 1, 1024, 2), dtype=f32)
    Xd = np.zeros((1, 1, 1024, 2), dtype=f32)
    Xd[0, 0, 100, :] = 0.5
    xs = _sine_f32(0.8, 880, 48000, 0.2)
    xs = xs[:, : 1024 * (xs.shape[1] // 1024)]
    Xs = mdct(1024, "vorbis", f32).transform(xs)
    # widely varying levels: random spectrum with a 1e-6 .. 1 envelope and one silent channel
    env = np.logspace(-6, 0, 1024).reshape(1, 1, 1024, 1)
    Xe = (rng.uniform(-1, 1, (1, 2, 1024, 2)) * env).astype(f32)
    Xe[0, 1, :, 1] = 0.0
    p64, p32 = psy(48000, 1024, 64, f64), psy(48000, 1024, 64, f32)
    arrs = {}
    for name, X in (("rand", Xr), ("zero", Xz), ("delta", Xd), ("sine", Xs), ("envelope", Xe)):
        arrs["X_" + name] = X
        for tag, p, dt in (("ref64", p64, f64), ("ref32", p32, f32)):
            Xc = X.astype(dt)
            t = p.tonality(Xc)
            arrs["t_%s_%s" % (name, tag)] = t
            for drown in (0.0, 0.5, 1.0):
                if drown != 0.0 and name not in ("rand", "delta"):
                    continue
                arrs["thr_%s_d%02d_%s" % (name, int(drown * 10), tag)] = p.global_masking_threshold(Xc, t, drown)
    save("psy_48000_1024_64_cases", **arrs)

    # 5b. the same kind of cases for the other models the wave-level masking kernels serve: N = 2048 at 48 kHz (BASELINE
    #     configs[3]) and 96 kHz, N = 1024 at 44.1 kHz.  Spectra: interior frames of a reference transform of uniform noise,
    #     all-zero, single-bin delta, 1e-6 .. 1 envelope with one silent channel; drown 0 / 0.5 / 1 on rand and delta.
    rng2 = np.random.default_rng(2)   # (its own stream: the fixtures above and below keep their values)
    for sr, Nf in ((48000, 2048), (44100, 1024), (96000, 2048)):
        gx = np.load(os.path.join(OUT, "mdct_n%d_rand_vorbis.npz" % Nf))
        Xr = gx["X_ref32"][:, 1:3]                                   # [1,2,Nf,2] float32, interior frames
        Xz = np.zeros((1, 1, Nf, 2), dtype=f32)
        Xd = np.zeros((1, 1, Nf, 2), dtype=f32)
        Xd[0, 0, 100, :] = 0.5
        env = np.logspace(-6, 0, Nf).reshape(1, 1, Nf, 1)
        Xe = (rng2.uniform(-1, 1, (1, 2, Nf, 2)) * env).astype(f32)
        Xe[0, 1, :, 1] = 0.0
        q64, q32 = psy(sr, Nf, 64, f64), psy(sr, Nf, 64, f32)
        arrs = {}
        for name, X in (("rand", Xr), ("zero", Xz), ("delta", Xd), ("envelope", Xe)):
            arrs["X_" + name] = X
            for tag, p, dt in (("ref64", q64, f64), ("ref32", q32, f32)):
                Xc = X.astype(dt)
                t = p.tonality(Xc)
                arrs["t_%s_%s" % (name, tag)] = t
                for drown in (0.0, 0.5, 1.0):
                    if drown != 0.0 and name not in ("rand", "delta"):
                        continue
                    arrs["thr_%s_d%02d_%s" % (name, int(drown * 10), tag)] = p.global_masking_threshold(Xc, t, drown)
        save("psy_%d_%d_64_cases" % (sr, Nf), **arrs)

    # 5c. BASELINE configs[0]: one 1-s mono 48 kHz clip, N = 1024 round trip: 48 000 samples
    #     truncated to 46 blocks, 0.8 sin(2 pi 880 t / 48000) + uniform(-0.1, 0.1) noise (default_rng(0))
    t48 = np.arange(46 * 1024, dtype=f64)
    x1 = (0.8 * np.sin(2.0 * np.pi * 880.0 * t48 / 48000.0)
          + np.random.default_rng(0).uniform(-0.1, 0.1, t48.shape)).astype(f32).reshape(1, -1, 1)
    m64, m32 = mdct(1024, "vorbis", f64), mdct(1024, "vorbis", f32)
    X64 = m64.transform(x1.astype(f64))
    save("mdct_n1024_mono_1s", x=x1, X_ref64=X64, X_ref32=m32.transform(x1),
         xhat_ref64=m64.inverse_transform(X64))

    # the reference's own tonality test configuration (tests/test_psychoacoustic.py:32-65): sr = N = 64
    x = _sine_f32(0.8, 4, 64, 5.0)
    Xt = mdct(64, "vorbis", f32).transform(x)
    xn = rng.uniform(-1, 1, (2, 10 * 64, 2)).astype(f32)
    Xn = mdct(64, "vorbis", f32).transform(xn)
    q64, q32 = psy(64, 64, 64, f64), psy(64, 64, 64, f32)
    arrs = {"X_tone": Xt, "X_noise": Xn}
    for name, X in (("tone", Xt), ("noise", Xn)):
        for tag, p, dt in (("ref64", q64, f64), ("ref32", q32, f32)):
            t = p.tonality(X.astype(dt))
            arrs["t_%s_%s" % (name, tag)] = t
            arrs["thr_%s_%s" % (name, tag)] = p.global_masking_threshold(X.astype(dt), t)
    save("psy_64_64_64_cases", **arrs)

    # 5d. sizes beside the powers of two and the masking model on general band layouts: filters_n = 960 (the 20-ms frame
    #     of a 48 kHz speech / music codec; mixed-radix tier), 512 and 128 (bins overlap several Bark bands): transform,
    #     round trip, tonality and thresholds of interior frames (drown 0 / 0.5), float64 and float32 reference runs
    rng3 = np.random.default_rng(3)   # (its own stream again)
    for Nf in (960, 512, 128):   # (128: eight frames per wave; appended last, the streams of the first two are unchanged)
        xr = rng3.uniform(-1, 1, (1, 5 * Nf, 2)).astype(f32)
        m64, m32 = mdct(Nf, "vorbis", f64), mdct(Nf, "vorbis", f32)
        X64 = m64.transform(xr.astype(f64))
        X32 = m32.transform(xr)
        arrs = {"x": xr, "X_ref64": X64, "X_ref32": X32, "xhat_ref64": m64.inverse_transform(X64)}
        env = np.logspace(-5, 0, Nf).reshape(1, 1, Nf, 1)
        Xe = (rng3.uniform(-1, 1, (1, 2, Nf, 2)) * env).astype(f32)
        q64, q32 = psy(48000, Nf, 64, f64), psy(48000, Nf, 64, f32)
        for name, X in (("rand", np.asarray(X32)[:, 1:4]), ("envelope", Xe)):
            arrs["Xp_" + name] = X
            for tag, pm, dt in (("ref64", q64, f64), ("ref32", q32, f32)):
                Xc = X.astype(dt)
                t = pm.tonality(Xc)
                arrs["t_%s_%s" % (name, tag)] = t
                for drown in (0.0, 0.5):
                    arrs["thr_%s_d%02d_%s" % (name, int(drown * 10), tag)] = pm.global_masking_threshold(Xc, t, drown)
        save("codec_48000_%d_64_cases" % Nf, **arrs)

    # 6. dB utilities (psychoacoustic.py:71-100)
    a = np.array([0.0, 1e-7, 1e-3, 0.5, 1.0, -0.25], dtype=f32)
    save("db_utils", a=a, dB_ref32=p32.amplitude_to_dB(a), dBn_ref32=p32.amplitude_to_dB_norm(a),
         dB_ref64=p64.amplitude_to_dB(a.astype(f64)), dBn_ref64=p64.amplitude_to_dB_norm(a.astype(f64)))
    # 7. precompute_dtype = float32 (mdctransformer.py:13-14,31-35,58-59; psychoacoustic.py:14-15,61-69): the constants built in
    #    float32 arithmetic.  The reference's TensorFlow known-answer vector (fixture 1) stems from such a revision.  A file of
    #    its own: the fixtures above keep their bytes.
    def mdct32(N, wt, dt):
        return MDCT(N, window_type=wt, compute_dtype=np.dtype(dt), precompute_dtype=tf.float32)

    def psy32(sr, N, M, dt, alpha=0.6):
        return Psy(sr, filter_bands_n=N, bark_bands_n=M, alpha=alpha, compute_dtype=np.dtype(dt), precompute_dtype=tf.float32)

    arrs = {}
    g1 = np.load(os.path.join(OUT, "mdct_n64_sine.npz"))
    m = mdct32(64, "vorbis", f32)
    arrs["n64_x"] = g1["x"]
    arrs["n64_X_ref32pre"] = m.transform(g1["x"])                       # float32 compute, float32 precompute
    arrs["n64_H"], arrs["n64_H_inv"] = np.asarray(m.H), np.asarray(m.H_inv)
    arrs["n64_H_dtype"] = np.array(str(np.asarray(m.H).dtype))
    g2 = np.load(os.path.join(OUT, "mdct_n256_roundtrip.npz"))
    m = mdct32(256, "vorbis", f32)
    X = m.transform(g2["x"])
    arrs["n256_x"], arrs["n256_X_ref32pre"], arrs["n256_xhat_ref32pre"] = g2["x"], X, m.inverse_transform(X)
    for wt in ("sine", "rect"):
        m = mdct32(16, wt, f32)
        arrs["n16_%s_H" % wt], arrs["n16_%s_H_inv" % wt] = np.asarray(m.H), np.asarray(m.H_inv)
    for sr, N, M in ((48000, 1024, 64), (32768, 64, 64)):
        p = psy32(sr, N, M, f32)
        tag = "psy_%d_%d_%d_" % (sr, N, M)
        wi, wv = _triplets(np.asarray(p.W))
        vi, vv = _triplets(np.asarray(p.W_inv))
        arrs.update({tag + "W_idx": wi, tag + "W_val": wv, tag + "W_inv_idx": vi, tag + "W_inv_val": vv,
                     tag + "S": np.asarray(p.spreading_matrix), tag + "quiet": np.asarray(p.quiet_threshold_intensity).reshape(-1),
                     tag + "max_bark": np.asarray(p.max_bark), tag + "bark_band_width": np.asarray(p.bark_band_width)})
    gp = np.load(os.path.join(OUT, "psy_48000_1024_64_cases.npz"))
    p = psy32(48000, 1024, 64, f32)
    for name in ("rand", "envelope"):
        Xc = gp["X_" + name]
        t = p.tonality(Xc)
        arrs["psy_X_" + name], arrs["psy_t_" + name] = Xc, t
        arrs["psy_thr_%s_d00" % name] = p.global_masking_threshold(Xc, t, 0.0)
    arrs["psy_thr_rand_d05"] = p.global_masking_threshold(gp["X_rand"], p.tonality(gp["X_rand"]), 0.5)
    save("precompute_float32_cases", **arrs)
    return 0


if __name__ == "__main__":
    sys.exit(main())
