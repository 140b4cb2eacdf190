"""GPU (MI355X): parity of the HIP path -- called through the C ABI -- with the oracle and the golden
vectors.  Tolerances (BASELINE.json north star / SURVEY 8(c)):
  * MDCT coefficients: max_frame(|dX|_inf / |X|_inf) <= 1e-4 and rel-L2 <= 1e-4 vs the fp64 reference values
  * tonality: |dt| <= 1e-4 |t| + 1e-6 (conftest.tonality_err <= 1);  thresholds: element-wise relative <= 1e-4
  * round trip PCM: <= 1 LSB of int16 (3.05e-5) and identical after rounding to int16
"""

import numpy as np
import pytest
import torch

from conftest import rel_elem, rel_l2, rel_peak, tonality_err

import audiocodec_amd
from audiocodec_amd import _lib
from oracle.audiocodec_oracle import MDCTOracle, PsychoOracle, sine_wav

pytestmark = pytest.mark.gpu

TOL = 1e-4
LSB = 1.0 / 32768.0


@pytest.fixture(scope="module", autouse=True)
def _require_gpu():
    assert torch.cuda.is_available(), "these tests need the MI355X (run with -m gpu on the GPU box)"
    _lib.load()
    yield
    _lib.load().ac_set_force_generic(0)


@pytest.fixture(params=["auto", "generic"])
def path(request):
    assert _lib.load().ac_set_force_generic(1 if request.param == "generic" else 0) == 0, "AC_TESTING=1 not in effect"
    yield request.param
    _lib.load().ac_set_force_generic(0)


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def host(t):
    return t.detach().cpu().numpy()


MDCT_CASES = [("mdct_n64_sine", 64, "vorbis"), ("mdct_n256_roundtrip", 256, "vorbis"),
              ("mdct_n1024_rand_vorbis", 1024, "vorbis"), ("mdct_n1024_rand_sine", 1024, "sine"),
              ("mdct_n2048_rand_vorbis", 2048, "vorbis"), ("mdct_n16_rand_vorbis", 16, "vorbis"),
              ("mdct_n16_rand_sine", 16, "sine"), ("mdct_n16_rand_rect", 16, "rect"),
              ("mdct_n12_rand_vorbis", 12, "vorbis"),
              ("mdct_n1024_mono_1s", 1024, "vorbis")]      # BASELINE configs[0]: one 1-s mono 48 kHz clip


@pytest.mark.parametrize("name,N,wt", MDCT_CASES)
def test_mdct_golden(golden, path, name, N, wt):
    g = golden(name)
    m = audiocodec_amd.MDCTransformer(N, window_type=wt)
    X = host(m.transform(dev(g["x"])))
    assert X.shape == g["X_ref64"].shape and X.dtype == np.float32
    assert rel_peak(X, g["X_ref64"]) <= TOL
    assert rel_l2(X, g["X_ref64"]) <= TOL
    if "xhat_ref64" in g:
        xh = host(m.inverse_transform(dev(g["X_ref32"])))
        assert xh.shape == g["xhat_ref64"].shape
        assert np.max(np.abs(xh - g["xhat_ref64"])) <= LSB


def test_known_answer_vector(golden, path):
    """tests/test_mdctransformer.py:39-54"""
    g = golden("mdct_n64_sine")
    X = host(audiocodec_amd.MDCTransformer(64).transform(dev(g["x"])))
    a = g["known_answer_frame1_first10"]
    assert np.all(X[0, 1, :10, 0] - a < 1e-6)
    assert np.max(np.abs(X[0, 1, :10, 0] - a)) < 1e-6          # two-sided, the bound the reference and the oracle test use


def test_known_answer_vector_with_float32_precompute(golden):
    """The reference's TensorFlow-generated vector (tests/test_mdctransformer.py:51-52) stems from a float32-precompute
    revision (mdctransformer.py:13-14,31-35, the cancellation at :218-221): with precompute_dtype=float32 the HIP path --
    the sixteen-frames-per-wave kernel in its FOLD4 form, four coefficients per fold block -- meets it to 1e-7 TWO-SIDED,
    against 6e-7 for the float64-precompute default; and meets the reference's own float32-precompute output
    (fixture X_ref32pre) to 1e-7 everywhere."""
    g, g32 = golden("mdct_n64_sine"), golden("precompute_float32_cases")
    _lib.load().ac_set_force_generic(0)
    m = audiocodec_amd.MDCTransformer(64, precompute_dtype=torch.float32)
    assert m.is_fast()                                          # wave-level kernels, not a fallback tier
    X = host(m.transform(dev(g["x"])))
    a = g["known_answer_frame1_first10"]
    print("float32 precompute: |X - TF vector| = %.2e, |X - X_ref32pre| = %.2e"
          % (np.max(np.abs(X[0, 1, :10, 0] - a)), np.max(np.abs(X - g32["n64_X_ref32pre"]))))
    assert np.max(np.abs(X[0, 1, :10, 0] - a)) <= 1e-7
    assert np.max(np.abs(X - g32["n64_X_ref32pre"])) <= 1e-7
    m256 = audiocodec_amd.MDCTransformer(256, precompute_dtype=torch.float32)
    X = m256.transform(dev(g32["n256_x"]))
    assert np.max(np.abs(host(X) - g32["n256_X_ref32pre"])) <= 2e-7
    assert np.max(np.abs(host(m256.inverse_transform(X)) - g32["n256_xhat_ref32pre"])) <= 2e-6


@pytest.mark.parametrize("N,C,wt", [(64, 1, "vorbis"), (64, 2, "sine"), (128, 2, "vorbis"), (256, 1, "vorbis"), (512, 2, "vorbis"),
                                    (1024, 2, "vorbis"), (2048, 1, "sine"), (960, 2, "vorbis"), (12, 3, "vorbis"),
                                    (64, 2, "rect"), (256, 2, "rect"), (512, 1, "rect"), (128, 3, "rect")])
def test_float32_precompute_and_rect_windows_vs_oracle(path, N, C, wt):
    """Fold blocks that are not rotations -- float32-precomputed windows and the rectangular window (mdctransformer.py:
    209-229) -- on every tier: the FOLD4 form of the several-frames-per-wave kernels (N = 64 ... 512, mono / stereo), the
    LDS-FFT tier and the O(N^2) kernels elsewhere.  Against the oracle run with the same precompute dtype, float64
    compute: coefficients to 1e-4 of the frame peak, the inverse to 1 LSB of the signal's own peak, the round trip of a
    Princen-Bradley window to 1 LSB; chunked (streaming) and one-shot results identical."""
    pre = np.float64 if wt == "rect" else np.float32
    B, K = 3, 7
    rng = np.random.default_rng(N + C)
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, precompute_dtype=torch.float32 if pre == np.float32 else torch.float64)
    if path == "auto" and N in (64, 128, 256, 512):
        assert m.is_fast()
    o = MDCTOracle(N, wt, np.float64, precompute_dtype=pre)
    X = m.transform(dev(x))
    Xo = o.transform(x.astype(np.float64))
    assert rel_peak(host(X), Xo) <= TOL and rel_l2(host(X), Xo) <= TOL
    xh = host(m.inverse_transform(X))
    xo = o.inverse_transform(Xo)
    assert np.max(np.abs(xh - xo)) <= LSB * max(1.0, np.max(np.abs(xo)))
    if wt != "rect":
        assert np.max(np.abs(xh[:, N:-N] - x)) <= LSB
    if C <= 2 or N > 64:
        st = audiocodec_amd.StreamingMDCT(m, B, C)
        parts = [st.transform_chunk(dev(x[:, a * N:b * N])) for a, b in ((0, 3), (3, 4), (4, 7))]
        assert torch.equal(torch.cat(parts, dim=1), X[:, :K])
        back = torch.cat([st.inverse_chunk(p_) for p_ in parts], dim=1)
        assert float((back - torch.from_numpy(xh[:, :K * N]).cuda()).abs().max()) <= 1e-6 * max(1.0, np.max(np.abs(xo)))
        st.close()


def test_inverse_identity_like_reference(path):
    """tests/test_mdctransformer.py:19-37"""
    N = 256
    x = sine_wav(0.8, 880, sample_rate=16000, duration_sec=1.0)
    x = x[:, : N * (x.shape[1] // N)]
    m = audiocodec_amd.MDCTransformer(N)
    xh = host(m.inverse_transform(m.transform(dev(x))))
    assert np.max(np.abs(x - xh[:, N:-N])) < 1e-5


def test_shape_like_reference(path):
    """tests/test_mdctransformer.py:56-75"""
    x = torch.randn(128, 10 * 64, 2, device="cuda")
    X = audiocodec_amd.MDCTransformer(64).transform(x)
    assert tuple(X.shape) == (128, 11, 64, 2)


@pytest.mark.parametrize("B,K,C,N", [(3, 5, 2, 1024), (2, 3, 1, 1024), (2, 4, 3, 1024), (1, 1, 2, 1024),
                                     (5, 7, 2, 256), (2, 2, 5, 64), (1, 3, 2, 2048), (2, 37, 2, 1024),
                                     (3, 9, 2, 2048), (2, 5, 1, 2048), (1, 6, 3, 2048), (2, 3, 4, 1024), (1, 2, 5, 1024),
                                     (3, 4, 1, 1024), (2, 3, 2, 512), (1, 5, 1, 4096), (2, 2, 3, 32),
                                     # filters_n whose half is 5-smooth: the mixed-radix LDS-FFT tier (Opus / MP3 sizes)
                                     (2, 5, 2, 960), (3, 4, 1, 480), (2, 7, 2, 240), (1, 9, 3, 120), (2, 3, 2, 576),
                                     (2, 4, 1, 192), (1, 2, 2, 1920), (1, 2, 2, 3072), (2, 3, 2, 30), (1, 3, 2, 1536),
                                     (2, 2, 2, 2000), (1, 4, 2, 36)])
def test_mdct_random_vs_oracle(path, B, K, C, N):
    rng = np.random.default_rng(B * 1000 + K * 10 + C)
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    m = audiocodec_amd.MDCTransformer(N)
    o = MDCTOracle(N, "vorbis", np.float64)
    X = host(m.transform(dev(x)))
    Xo = o.transform(x.astype(np.float64))
    assert rel_peak(X, Xo) <= TOL and rel_l2(X, Xo) <= TOL
    xh = host(m.inverse_transform(dev(X)))
    assert xh.shape == (B, (K + 2) * N, C)
    assert np.max(np.abs(xh[:, N:-N] - x)) <= LSB
    assert np.array_equal(np.round(xh[:, N:-N] * 32767), np.round(x * 32767)) or \
        np.max(np.abs(np.round(xh[:, N:-N] * 32767) - np.round(x * 32767))) <= 1
    # head and tail blocks of the round trip are the aliased halves; compare with the oracle too
    xo = o.inverse_transform(Xo)
    assert np.max(np.abs(xh - xo)) <= LSB


def _smooth_half(N):
    h = N // 2
    for r in (2, 3, 5):
        while h % r == 0:
            h //= r
    return h == 1


# every filters_n the 16-byte kernels of the LDS-FFT tier serve (float32 stereo rows, filters_n % 4 == 0 up to 8192 with a
# 5-smooth half, beside the powers of two of the wave-level kernels): one compile-time instance each (ac_generic.hip
# AC_WAVE_CT_SIZES; a frame per group of lanes inside a wave up to 1024, per workgroup of two / four / eight waves and in
# place above, up to 8192 -- 7500 has no plan of four passes and stays on the O(N^2) kernels); the reference takes any even filters_n (mdctransformer.py:26)
WAVE16_SIZES = [N for N in range(16, 8193, 4) if _smooth_half(N) and N not in (64, 128, 256, 512, 1024, 2048, 7500)]


@pytest.mark.parametrize("C", [2, 1, 3])
@pytest.mark.parametrize("N", WAVE16_SIZES)
def test_lds_fft_wave_16_byte_kernels_every_size(N, C):
    """Every instance in its three row layouts (stereo; mono: two signals per complex pair, the last pair of the odd batch
    half empty; three channels: pairs (0, 1) and (2, -)), against the fp64 oracle: analysis, synthesis incl. the aliased
    head / tail blocks, and the round trip to 1 LSB."""
    rng = np.random.default_rng(N + C)
    assert audiocodec_amd.MDCTransformer(N).tier(C) == 2   # (a silent fall to the run-time forms would pass the numerics)
    B, K = 3, (70 if N <= 128 else 37 if N <= 480 else 11 if N <= 2048 else 6 if N <= 4096 else 3)
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    m = audiocodec_amd.MDCTransformer(N)
    o = MDCTOracle(N, "vorbis", np.float64)
    X = host(m.transform(dev(x)))
    Xo = o.transform(x.astype(np.float64))
    assert rel_peak(X, Xo) <= TOL and rel_l2(X, Xo) <= TOL
    xh = host(m.inverse_transform(dev(X)))
    assert np.max(np.abs(xh[:, N:-N] - x)) <= LSB
    assert np.max(np.abs(xh - o.inverse_transform(Xo))) <= LSB


@pytest.mark.parametrize("N,C", [(960, 2), (480, 1), (1920, 2), (48, 1)])
def test_lds_fft_tier_rows_off_the_16_byte_grid(N, C):
    """A view whose first element is not on a 16-byte boundary: the C ABI wants 16-byte aligned tensors (AC_REQUIRE_ALIGNED),
    the package hands it an aligned copy; same values."""
    B, K = 2, 9
    g = torch.Generator(device="cuda").manual_seed(N)
    flat = torch.empty(B * K * N * C + 4, device="cuda").uniform_(-1, 1, generator=g)
    x_off = flat[1:1 + B * K * N * C].view(B, K * N, C)
    assert x_off.is_contiguous() and x_off.data_ptr() % 8 == 4
    x = x_off.clone()
    m = audiocodec_amd.MDCTransformer(N)
    X, X_off = m.transform(x), m.transform(x_off)
    assert float((X - X_off).abs().max()) <= 2e-6 * float(X.abs().max())
    Xf = torch.empty(X.numel() + 4, device="cuda")
    X_view = Xf[1:1 + X.numel()].view(X.shape)
    X_view.copy_(X)
    y, y_off = m.inverse_transform(X), m.inverse_transform(X_view)
    assert float((y - y_off).abs().max()) <= 2e-6


@pytest.mark.parametrize("N,C,window", [(64, 3, "vorbis"), (128, 5, "sine"), (256, 3, "vorbis"), (512, 6, "vorbis"), (64, 3, "rect"),
                                        (256, 4, "rect"), (512, 3, "rect"), (1024, 2, "rect"), (2048, 2, "rect"), (2048, 3, "rect")])
def test_lds_fft_wave_16_byte_kernels_at_the_powers_of_two(N, C, window):
    """What the wave-level kernels leave at their own sizes -- the rectangular window at 1024 / 2048, more than two channels
    below 1024 -- runs instances of the LDS-FFT tier as well (mdctransformer.py:26, 199-211)."""
    rng = np.random.default_rng(N + C)
    m = audiocodec_amd.MDCTransformer(N, window)
    assert m.tier(C) == 2
    B, K = 2, 9
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    o = MDCTOracle(N, window, np.float64)
    X = host(m.transform(dev(x)))
    Xo = o.transform(x.astype(np.float64))
    assert rel_peak(X, Xo) <= TOL and rel_l2(X, Xo) <= TOL
    assert np.max(np.abs(host(m.inverse_transform(dev(X))) - o.inverse_transform(Xo))) <= LSB


def test_lds_fft_wave_16_byte_kernels_strip_lengths(tmp_path):
    """A frame's bits do not depend on where in a strip it falls (small batches run strips of one frame, large ones up to
    32: AC_LDS_WAVE_STRIP forces a length), nor on where a stream is cut: one-shot and chunked analysis agree bit for bit at
    every strip length, the synthesis to rounding."""
    import os, subprocess, sys
    from conftest import ROOT
    code = ("import sys, numpy as np, torch, audiocodec_amd\n"
            "out = {}\n"
            "for N, C in ((960, 2), (480, 1), (120, 2), (1920, 2), (4096, 1), (32, 2)):\n"
            "    g = torch.Generator(device='cuda').manual_seed(N)\n"
            "    K = 21\n"
            "    x = torch.empty(3, K * N, C, device='cuda').uniform_(-1, 1, generator=g)\n"
            "    m = audiocodec_amd.MDCTransformer(N)\n"
            "    X = m.transform(x)\n"
            "    st = audiocodec_amd.StreamingMDCT(m, 3, C)\n"
            "    Xs = torch.cat([st.transform_chunk(x[:, a * N:b * N].contiguous()) for a, b in ((0, 5), (5, 6), (6, 19), (19, 21))]\n"
            "                   + [st.transform_chunk(torch.zeros(3, N, C, device='cuda'))], dim=1)\n"
            "    assert torch.equal(Xs, X), ('chunked != one-shot', N, C)\n"
            "    out['X%d' % N] = X.cpu().numpy()\n"
            "    out['y%d' % N] = m.inverse_transform(X).cpu().numpy()\n"
            "np.savez(sys.argv[1], **out)\n")
    got = {}
    for strip in ("0", "1", "4", "32"):
        f = str(tmp_path / ("strip%s.npz" % strip))
        env = dict(os.environ)
        env.pop("AC_LDS_WAVE_STRIP", None)
        if strip != "0":
            env["AC_LDS_WAVE_STRIP"] = strip
        r = subprocess.run([sys.executable, "-c", code, f], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got[strip] = np.load(f)
    for strip in ("1", "4", "32"):
        for k in got["0"].files:
            a, b = got["0"][k], got[strip][k]
            if k.startswith("X"):
                assert np.array_equal(a, b), (k, strip)
            else:
                assert np.max(np.abs(a - b)) <= 1e-6, (k, strip)


def test_lds_fft_wave_16_byte_kernels_every_size_inside_a_strip(tmp_path):
    """test_lds_fft_wave_16_byte_kernels_every_size runs small batches, i.e. strips of one frame; here every instance (both row
    layouts) with strips of five frames -- the carried fold / aliased half -- against the strips-of-one results, which that
    test holds to the oracle."""
    import json, os, subprocess, sys
    from conftest import ROOT
    code = ("import sys, json, numpy as np, torch, audiocodec_amd\n"
            "out = {}\n"
            "for N in json.loads(sys.argv[2]):\n"
            "    for C in (2, 1, 5):\n"
            "        g = torch.Generator(device='cuda').manual_seed(N + C)\n"
            "        x = torch.empty(3, 7 * N, C, device='cuda').uniform_(-1, 1, generator=g)\n"
            "        m = audiocodec_amd.MDCTransformer(N)\n"
            "        X = m.transform(x)\n"
            "        out['X%d_%d' % (N, C)] = X.cpu().numpy()\n"
            "        out['y%d_%d' % (N, C)] = m.inverse_transform(X).cpu().numpy()\n"
            "np.savez(sys.argv[1], **out)\n")
    got = {}
    for strip in ("1", "5"):
        f = str(tmp_path / ("strip%s.npz" % strip))
        r = subprocess.run([sys.executable, "-c", code, f, json.dumps(WAVE16_SIZES)], cwd=ROOT,
                           env=dict(os.environ, AC_LDS_WAVE_STRIP=strip), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        got[strip] = np.load(f)
    for k in got["1"].files:
        a, b = got["1"][k], got["5"][k]
        if k.startswith("X"):
            assert np.array_equal(a, b), k
        else:
            assert np.max(np.abs(a - b)) <= 1e-6, k


def test_lds_fft_wave_16_byte_kernels_run_time_form(tmp_path):
    """The same kernels with the size as a run-time argument (AC_LDS_WAVE_NOCT=1, the A/B reference of the compile-time
    instances) agree with the instances to float32 rounding (the compiler contracts the two forms differently)."""
    import os, subprocess, sys
    from conftest import ROOT
    sizes = (960, 480, 120, 48, 16)
    code = ("import sys, numpy as np, torch, audiocodec_amd\n"
            "out = {}\n"
            "for N in %r:\n"
            "    g = torch.Generator(device='cuda').manual_seed(N)\n"
            "    x = torch.empty(2, 35 * N, 2, device='cuda').uniform_(-1, 1, generator=g)\n"
            "    m = audiocodec_amd.MDCTransformer(N)\n"
            "    X = m.transform(x)\n"
            "    out['X%%d' %% N] = X.cpu().numpy()\n"
            "    out['y%%d' %% N] = m.inverse_transform(X).cpu().numpy()\n"
            "np.savez(sys.argv[1], **out)\n" % (sizes,))
    got = []
    for noct in ("0", "1"):
        f = str(tmp_path / ("noct%s.npz" % noct))
        r = subprocess.run([sys.executable, "-c", code, f], cwd=ROOT, env=dict(os.environ, AC_LDS_WAVE_NOCT=noct),
                           capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got.append(np.load(f))
    for N in sizes:
        for k in ("X%d" % N, "y%d" % N):
            a, b = got[0][k], got[1][k]
            assert np.max(np.abs(a - b)) <= 2e-6 * np.max(np.abs(a)), k


def test_int16_pcm_round_trip_exact(path):
    """PCM that came from int16 survives encode -> decode bit-exactly after re-quantisation."""
    rng = np.random.default_rng(7)
    pcm = rng.integers(-32768, 32768, (4, 6 * 1024, 2), dtype=np.int16)
    x = (pcm.astype(np.float32) / 32768.0)
    m = audiocodec_amd.MDCTransformer(1024)
    xh = host(m.inverse_transform(m.transform(dev(x))))[:, 1024:-1024]
    assert np.max(np.abs(xh - x)) <= LSB
    assert np.array_equal(np.round(xh * 32768.0).astype(np.int32), pcm.astype(np.int32))


@pytest.mark.parametrize("N,C", [(960, 2), (960, 1), (120, 3), (4096, 2), (8192, 1), (16, 2)])
def test_empty_and_edge_shapes_on_the_lds_fft_instances(N, C):
    """No samples, no clips, one block, one clip through the instances of the LDS-FFT tier (strips of one frame, a lone pair
    of a mono batch, frames that are all aliased halves)."""
    m = audiocodec_amd.MDCTransformer(N)
    X = m.transform(torch.zeros(2, 0, C, device="cuda"))
    assert tuple(X.shape) == (2, 1, N, C) and float(X.abs().max()) == 0.0
    assert tuple(m.transform(torch.zeros(0, 2 * N, C, device="cuda")).shape) == (0, 3, N, C)
    y = m.inverse_transform(torch.zeros(1, 0, N, C, device="cuda"))
    assert tuple(y.shape) == (1, N, C) and float(y.abs().max()) == 0.0
    rng = np.random.default_rng(N + C)
    o = MDCTOracle(N, "vorbis", np.float64)
    for B, K in ((1, 1), (1, 2), (5, 1)):
        x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
        Xo = o.transform(x.astype(np.float64))
        Xg = m.transform(dev(x))
        assert rel_peak(host(Xg), Xo) <= TOL
        assert np.max(np.abs(host(m.inverse_transform(Xg)) - o.inverse_transform(Xo))) <= LSB


def test_empty_and_edge_shapes(path):
    m = audiocodec_amd.MDCTransformer(64)
    X = m.transform(torch.zeros(2, 0, 2, device="cuda"))
    assert tuple(X.shape) == (2, 1, 64, 2) and float(X.abs().max()) == 0.0
    assert tuple(m.transform(torch.zeros(0, 128, 2, device="cuda")).shape) == (0, 3, 64, 2)
    x = m.inverse_transform(torch.zeros(1, 0, 64, 1, device="cuda"))
    assert tuple(x.shape) == (1, 64, 1) and float(x.abs().max()) == 0.0
    with pytest.raises(ValueError):
        m.transform(torch.zeros(1, 100, 1, device="cuda"))
    with pytest.raises(ValueError):
        m.transform(torch.zeros(1, 128, 1, device="cuda", dtype=torch.float64))
    with pytest.raises(ValueError):
        m.inverse_transform(torch.zeros(1, 2, 32, 1, device="cuda"))
    # non-contiguous input is accepted (made contiguous), result identical
    xs = torch.rand(2, 256, 4, device="cuda")[:, :, ::2]
    assert torch.equal(m.transform(xs), m.transform(xs.contiguous()))


def test_linearity_and_shift_properties(path):
    """Size-independent properties: linearity and block-shift covariance of the analysis bank."""
    N = 1024
    m = audiocodec_amd.MDCTransformer(N)
    x1 = torch.rand(2, 8 * N, 2, device="cuda") * 2 - 1
    x2 = torch.rand(2, 8 * N, 2, device="cuda") * 2 - 1
    lhs = m.transform(0.5 * x1 - 0.25 * x2)
    rhs = 0.5 * m.transform(x1) - 0.25 * m.transform(x2)
    assert float((lhs - rhs).abs().max()) < 2e-6
    shifted = torch.cat([torch.zeros(2, N, 2, device="cuda"), x1], dim=1)
    Xs = m.transform(shifted)
    assert float(Xs[:, 0].abs().max()) == 0.0
    assert torch.equal(Xs[:, 1:], m.transform(x1))


# ---- psychoacoustic model ---------------------------------------------------------------------------

@pytest.mark.parametrize("cfg,sr,N,M", [("psy_48000_1024_64_cases", 48000, 1024, 64),
                                        ("psy_64_64_64_cases", 64, 64, 64),
                                        ("psy_48000_2048_64_cases", 48000, 2048, 64),     # BASELINE configs[3]'s model
                                        ("psy_44100_1024_64_cases", 44100, 1024, 64),
                                        ("psy_96000_2048_64_cases", 96000, 2048, 64)])
def test_psy_golden(golden, path, cfg, sr, N, M):
    """tonality / global_masking_threshold against values the reference's own code produced (oracle/gen_golden.py);
    with path = auto the models at N = 1024 / 2048 run the wave-level kernels"""
    g = golden(cfg)
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    if path == "auto" and N in (1024, 2048):
        assert p.is_fast()
    for key in [k for k in g if k.startswith("X_")]:
        name = key[2:]
        X = dev(g[key].astype(np.float32))
        t_ref = g["t_%s_ref64" % name]
        t = host(p.tonality(X))
        assert t.shape == t_ref.shape
        assert tonality_err(t, t_ref) <= 1.0, name
        for k2 in [k for k in g if k.startswith("thr_" + name) and k.endswith("ref64")]:
            mid = k2[len("thr_" + name):-len("ref64")].strip("_")
            drown = int(mid[1:]) / 10.0 if mid else 0.0
            thr = host(p.global_masking_threshold(X, dev(t_ref.astype(np.float32)), drown))
            assert rel_elem(thr, g[k2]) <= TOL, (name, k2)
            assert thr.min() >= 1e-7 * (1 - 1e-6)


@pytest.mark.parametrize("N", [960, 512, 128])
def test_codec_golden_beside_the_powers_of_two(golden, path, N):
    """filters_n = 960 (mixed-radix LDS-FFT tier) and 512 (two frames per wave), each with the masking model for general
    band layouts, against values the reference's own code produced (oracle/gen_golden.py 5d)"""
    g = golden("codec_48000_%d_64_cases" % N)
    codec = audiocodec_amd.AudioCodec(48000, N)
    X = host(codec.mdct.transform(dev(g["x"])))
    assert rel_peak(X, g["X_ref64"]) <= TOL and rel_l2(X, g["X_ref64"]) <= TOL
    xh = host(codec.mdct.inverse_transform(dev(g["X_ref32"])))
    assert np.max(np.abs(xh - g["xhat_ref64"])) <= LSB
    if path == "auto":
        assert codec.psy.tier() == 1
    for name in ("rand", "envelope"):
        Xp = dev(g["Xp_" + name])
        t_ref = g["t_%s_ref64" % name]
        assert tonality_err(host(codec.psy.tonality(Xp)), t_ref) <= 1.0
        for drown in (0.0, 0.5):
            thr = host(codec.psy.global_masking_threshold(Xp, dev(t_ref.astype(np.float32)), drown))
            assert rel_elem(thr, g["thr_%s_d%02d_ref64" % (name, int(drown * 10))]) <= TOL
    # the codec's encode on the fixture's PCM: same spectrum, thresholds of its own spectrum within the bar of the model.
    # ONE launch: at 512 / 128 the masking model rides in the several-frames-per-wave kernels, at 960 in the LDS-FFT instance
    Xe, te, thre = codec.encode(dev(g["x"]))
    assert rel_peak(host(Xe), g["X_ref64"]) <= TOL
    if path == "auto":
        assert codec.encode_launches(2) == 1   # (960: the fused encode of the LDS-FFT tier, where it measured faster)
    # the fixture's interior frames are frames 1..3 of that spectrum (Xp_rand = X_ref32[:, 1:4]): reference-produced
    # tonality / thresholds of the reference's float32 spectrum against the fused kernel's own spectrum -- the spectra
    # agree to 1e-6 of the frame peak, which the thresholds feel at a few 1e-5
    assert tonality_err(host(te[:, 1:4]), g["t_rand_ref64"]) <= 1.0
    assert rel_elem(host(thre[:, 1:4]), g["thr_rand_d00_ref64"]) <= 3e-4
    o = PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    X64 = host(Xe).astype(np.float64)
    t64 = o.tonality(X64)
    assert tonality_err(host(te), t64) <= 1.0 and rel_elem(host(thre), o.global_masking_threshold(X64, t64)) <= TOL


@pytest.mark.parametrize("N", [108, 120, 240, 480, 500, 576, 768, 960, 1000, 1080, 1536, 1920, 2304, 3240, 4096])
@pytest.mark.parametrize("C", [2, 1])
def test_fused_encode_of_the_lds_fft_tier_equals_the_unfused_calls(N, C, monkeypatch):
    """k_enc_wave_v (ac_generic.hip): the LDS-FFT instances with the masking model in the same launch -- tonality and the band
    intensities frame by frame while the spectrum is in LDS, the rest four frames at a time after the strip -- against
    transform -> tonality -> global_masking_threshold (psychoacoustic.py:102-148 on mdctransformer.py:62-125): bit for bit (one
    definition of the arithmetic), every instance forced on (AC_LDS_WAVE_NOFUSE=2: the product fuses where it measured faster),
    one / several frames per wave and a frame on two / four waves, short and ragged strips, a batch that leaves lanes without
    a task; thresholds against the oracle at the bar."""
    monkeypatch.setenv("AC_LDS_WAVE_NOFUSE", "2")
    codec = audiocodec_amd.AudioCodec(48000, N)
    assert codec.mdct.tier(C) == 2 and codec.psy.tier() == 1
    for (B, K) in ((3, 5), (1, 1), (2, 37)):
        x = (torch.rand((B, K * N, C), device="cuda") * 2 - 1) * torch.rand((B, 1, C), device="cuda")
        assert codec.encode_launches(C) == 1
        X, t, thr = codec.encode(x)
        monkeypatch.setenv("AC_LDS_WAVE_NOFUSE", "1")
        assert codec.encode_launches(C) == 2
        X2 = codec.mdct.transform(x)
        t2 = codec.psy.tonality(X2)
        thr2 = codec.psy.global_masking_threshold(X2, t2)
        X3, t3, thr3 = codec.encode(x)
        monkeypatch.setenv("AC_LDS_WAVE_NOFUSE", "2")
        for a, b in ((X, X2), (t, t2), (thr, thr2), (X, X3), (t, t3), (thr, thr3)):
            assert torch.equal(a, b)
        if (B, K) == (3, 5):
            o = PsychoOracle(48000, N, 64, compute_dtype=np.float64)
            X64 = host(X).astype(np.float64)
            t64 = o.tonality(X64)
            assert tonality_err(host(t), t64) <= 1.0 and rel_elem(host(thr), o.global_masking_threshold(X64, t64)) <= TOL


@pytest.mark.parametrize("N", [300, 480, 512, 600, 640, 960, 1024, 1280, 1536, 2048])
@pytest.mark.parametrize("C", [3, 4, 5, 6, 7, 8])
def test_masking_model_of_more_than_two_channels_through_whole_rows(N, C, monkeypatch):
    """k_psy_runs_c (ac_psy_mid.hip): the waves that take the channel pairs of one frame move the [filter_bands_n, C] row and
    the threshold row in whole 16-byte pieces and pick their pairs out of a row image in LDS (psychoacoustic.py:102-148 takes
    any channels_n) -- against the strided channel pairs (AC_PSY_NOTEAM=1): tonality and thresholds bit for bit, twice (no run
    may differ from another), with a given tonality and with the kernel's own, odd channel counts (a pair either side of a
    piece boundary, a half-empty last pair), ragged frame counts, a launch of thousands of workgroups; and against the oracle."""
    psy = audiocodec_amd.PsychoacousticModel(48000, N)
    for (B, F) in ((3, 5), (1, 1), (2, 37), (24, 96000 // N)):
        X = (torch.rand((B, F, N, C), device="cuda") * 2 - 1) * torch.rand((B, F, 1, C), device="cuda")
        monkeypatch.delenv("AC_PSY_NOTEAM", raising=False)
        monkeypatch.setenv("AC_PSY_TEAM_ALWAYS", "1")     # (the product takes the form where it measured faster)
        t = psy.tonality(X)
        thr = psy.global_masking_threshold(X, t, 0.2)
        thr_b = psy.global_masking_threshold(X, t, 0.2)
        monkeypatch.setenv("AC_PSY_NOTEAM", "1")
        t1 = psy.tonality(X)
        thr1 = psy.global_masking_threshold(X, t1, 0.2)
        assert torch.equal(t, t1) and torch.equal(thr, thr1) and torch.equal(thr, thr_b)
        if (B, F) == (3, 5):
            o = PsychoOracle(48000, N, 64, compute_dtype=np.float64)
            X64 = host(X).astype(np.float64)
            t64 = o.tonality(X64)
            assert tonality_err(host(t), t64) <= 1.0 and rel_elem(host(thr), o.global_masking_threshold(X64, t64, 0.2)) <= TOL
    monkeypatch.delenv("AC_PSY_NOTEAM", raising=False)
    monkeypatch.delenv("AC_PSY_TEAM_ALWAYS", raising=False)


@pytest.mark.parametrize("N,C", [(512, 2), (512, 1), (256, 2), (128, 2), (64, 2), (960, 2), (2160, 2), (4096, 1)])
def test_fused_encode_at_launches_that_fill_the_chip(N, C, monkeypatch):
    """Several workgroups per CU, thousands of them: the launch shape the small parity cases never reach.  The fused encode
    twice on the same input (bit-equal runs: no wave may see another's work) and against transform -> tonality ->
    global_masking_threshold (bit for bit), the stand-alone masking kernel twice.  Round 4's first run-structured masking model
    passed every small case and returned wrong values in 0.5 - 2 % of the frames of such a launch (v_mfma_f32_16x16x32_bf16
    issued by one wave disturbed the vector arithmetic of the others; tools/scale_check.py is the bench-sized form of this
    test)."""
    monkeypatch.setenv("AC_LDS_WAVE_NOFUSE", "2")
    B, K = 96, 144000 // N
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = torch.empty((B, K * N, C), device="cuda").uniform_(-1, 1, generator=g)
    codec = audiocodec_amd.AudioCodec(48000, N)
    assert codec.encode_launches(C) == 1
    X, t, thr = codec.encode(x)
    Xb, tb, thrb = codec.encode(x)
    assert torch.equal(X, Xb) and torch.equal(t, tb) and torch.equal(thr, thrb)
    monkeypatch.setenv("AC_LDS_WAVE_NOFUSE", "1")
    X2 = codec.mdct.transform(x)
    t2 = codec.psy.tonality(X2)
    thr2 = codec.psy.global_masking_threshold(X2, t2)
    assert torch.equal(codec.psy.global_masking_threshold(X2, t2), thr2)
    assert torch.equal(X, X2) and torch.equal(t, t2) and torch.equal(thr, thr2)


@pytest.mark.parametrize("N", [64, 120, 128, 480, 500, 960, 1024, 1536, 2048, 4096])
@pytest.mark.parametrize("C", [3, 4, 5, 6, 7])
def test_more_than_two_channels_through_whole_rows_equals_the_strided_pairs(N, C, monkeypatch):
    """k_fwd_wave_c / k_inv_wave_c (ac_generic.hip): the channel pairs of one signal as a team that moves whole [filters_n, C]
    rows between HBM and LDS in 16-byte pieces (mdctransformer.py:112, 289-297 takes any channels_n) -- forced on wherever the
    shape fits (AC_LDS_WAVE_NOTEAM=2; the product takes it where it measured faster) against the strided channel pairs
    (=1): bit for bit, one-shot and chunked (the stream state crosses the two forms), odd channel counts (a half-empty last
    pair), several teams per workgroup and groups left over, short and ragged strips; and against the oracle at the bar."""
    m = audiocodec_amd.MDCTransformer(N)
    assert m.tier(C) == 2
    for (B, K) in ((3, 7), (1, 1), (2, 41)):
        x = (torch.rand((B, K * N, C), device="cuda") * 2 - 1) * torch.rand((B, 1, C), device="cuda")
        monkeypatch.setenv("AC_LDS_WAVE_NOTEAM", "2")
        X = m.transform(x)
        y = m.inverse_transform(X)
        monkeypatch.setenv("AC_LDS_WAVE_NOTEAM", "1")
        X1 = m.transform(x)
        y1 = m.inverse_transform(X1)
        assert torch.equal(X, X1) and torch.equal(y, y1)
        if (B, K) == (3, 7):
            o = MDCTOracle(N, "vorbis", np.float64)
            Xo = o.transform(host(x).astype(np.float64))
            assert rel_peak(host(X), Xo) <= TOL and rel_l2(host(X), Xo) <= TOL
            assert np.max(np.abs(host(y) - o.inverse_transform(Xo))) <= LSB
            # chunked, the analysis in the team form and the synthesis in the strided one, then the other way round
            for fwd_mode, inv_mode in (("2", "1"), ("1", "2")):
                st = audiocodec_amd.StreamingMDCT(m, B, C)
                parts = []
                for a, b in ((0, 2), (2, 3), (3, 7)):
                    monkeypatch.setenv("AC_LDS_WAVE_NOTEAM", fwd_mode)
                    Xc = st.transform_chunk(x[:, a * N: b * N])
                    assert torch.equal(Xc, X[:, a: b])
                    monkeypatch.setenv("AC_LDS_WAVE_NOTEAM", inv_mode)
                    parts.append(st.inverse_chunk(Xc))
                assert torch.equal(torch.cat(parts, dim=1), y[:, : K * N])
    monkeypatch.delenv("AC_LDS_WAVE_NOTEAM")


@pytest.mark.parametrize("N,C,sr", [(512, 2, 48000), (512, 1, 48000), (256, 2, 48000), (256, 1, 44100), (128, 2, 48000), (128, 1, 48000),
                                    (64, 2, 48000), (64, 1, 32768), (64, 2, 64)])
def test_fused_encode_below_1024_equals_the_unfused_calls(N, C, sr):
    """filters_n 512 / 256 / 128 / 64 -- the reference's own test sizes (tests/test_mdctransformer.py:23,42, composition at
    N = 64 in tests/test_psychoacoustic.py:36-41): encode() is ONE launch, k_fwd_multi with the masking model for general
    band layouts on the frames it has just transformed, and returns bit for bit what transform -> tonality ->
    global_masking_threshold return (the same device function on the same values; X is not read back from HBM).  Ragged
    frame counts (F not a multiple of the frames per wave), an odd number of mono signals, drown; the streaming form
    (ac_stream_encode: state included) chunk by chunk; and the float64 oracle at the stated bar."""
    _lib.load().ac_set_force_generic(0)
    B, K = (5 if C == 1 else 3), 11
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    x[0, : 3 * N] *= 1e-3                                     # a quiet stretch
    x[-1, 2 * N: 5 * N, 0] = 0.0                              # and a silent one
    codec = audiocodec_amd.AudioCodec(sr, N)
    assert codec.encode_launches(C) == 1 and codec.psy.tier() == 1
    for drown in (0.0, 0.3):
        X, t, thr = codec.encode(x, drown)
        Xu = codec.mdct.transform(x)
        tu = codec.psy.tonality(Xu)
        thru = codec.psy.global_masking_threshold(Xu, tu, drown)
        assert torch.equal(X, Xu) and torch.equal(t, tu) and torch.equal(thr, thru)
    o, om = PsychoOracle(sr, N, 64, compute_dtype=np.float64), MDCTOracle(N, "vorbis", np.float64)
    Xo = om.transform(host(x).astype(np.float64))
    assert rel_peak(host(X), Xo) <= TOL
    X64 = host(X).astype(np.float64)
    t64 = o.tonality(X64)
    assert tonality_err(host(t), t64) <= 1.0
    assert rel_elem(host(thr), o.global_masking_threshold(X64, t64, 0.3)) <= TOL
    st = codec.stream(B, C)
    parts = [st.encode_chunk(x[:, a * N:b * N].contiguous(), drown=0.3) for a, b in ((0, 1), (1, 6), (6, 11))]
    for i, ref in enumerate((X, t, thr)):
        assert torch.equal(torch.cat([p_[i] for p_ in parts], dim=1), ref[:, :K])
    st.close()


def test_tonality_like_reference(path):
    """tests/test_psychoacoustic.py:32-65"""
    N = 64
    m = audiocodec_amd.MDCTransformer(N)
    p = audiocodec_amd.PsychoacousticModel(sample_rate=N, filter_bands_n=N)
    X = m.transform(dev(sine_wav(0.8, 4, sample_rate=64, duration_sec=5.0)))
    assert float(p.tonality(X)[0, 1]) == 1.0
    # (the reference draws unseeded noise and asserts a mean below 0.1, which white noise at 64 bins only just keeps:
    # the same seeded draw as the oracle's test, and the oracle's values beside the bound)
    xn = np.random.default_rng(3).uniform(-1, 1, (10, 10 * N, 2)).astype(np.float32)
    X = m.transform(dev(xn))
    t = p.tonality(X)
    assert tuple(t.shape) == (10, 11, 1, 2)
    assert float(t[0, 1:-1].mean()) < 0.1
    to = PsychoOracle(N, N, compute_dtype=np.float64).tonality(host(X).astype(np.float64))
    assert tonality_err(host(t), to) <= 1.0


@pytest.mark.parametrize("sr,N,M,B,F,C", [(48000, 1024, 64, 3, 4, 2), (48000, 1024, 64, 2, 3, 1), (48000, 1024, 64, 1, 2, 3),
                                          (44100, 256, 48, 2, 5, 2), (32768, 64, 64, 2, 3, 2), (48000, 2048, 64, 1, 2, 2),
                                          (44100, 1024, 64, 2, 3, 2), (16000, 1024, 64, 1, 3, 1), (8000, 1024, 64, 1, 2, 2),
                                          (96000, 2048, 64, 2, 2, 2), (22050, 2048, 64, 1, 3, 3), (48000, 2048, 64, 3, 2, 1)])
def test_psy_random_vs_oracle(path, sr, N, M, B, F, C):
    rng = np.random.default_rng(F * 100 + C)
    env = np.logspace(-5, 0, N).reshape(1, 1, N, 1)
    X = (rng.uniform(-1, 1, (B, F, N, C)) * env * rng.uniform(1e-3, 1, (B, F, 1, C))).astype(np.float32)
    X[0, 0, :, 0] = 0.0
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    o = PsychoOracle(sr, N, M, compute_dtype=np.float64)
    t = host(p.tonality(dev(X)))
    to = o.tonality(X.astype(np.float64))
    assert tonality_err(t, to) <= 1.0
    for drown in (0.0, 0.3, 1.0):
        thr = host(p.global_masking_threshold(dev(X), dev(to.astype(np.float32)), drown))
        assert rel_elem(thr, o.global_masking_threshold(X.astype(np.float64), to, drown)) <= TOL


@pytest.mark.parametrize("B,K,C,N", [(3, 5, 2, 1024), (2, 4, 1, 1024), (1, 3, 3, 1024), (2, 37, 2, 1024), (3, 4, 1, 1024),
                                     (2, 5, 2, 2048), (3, 3, 1, 2048), (1, 4, 3, 2048)])
def test_encode_fused_equals_unfused_and_oracle(path, B, K, C, N):
    rng = np.random.default_rng(11 + B)
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    x[0, : 2 * N, 0] *= 1e-3
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(dev(x), drown=0.2)
    Xu = codec.mdct.transform(dev(x))
    tu = codec.psy.tonality(Xu)
    thru = codec.psy.global_masking_threshold(Xu, tu, 0.2)
    assert float((X - Xu).abs().max()) <= 1e-6
    assert tonality_err(t, tu) <= 1.0
    assert float(((thr - thru).abs() / thru).max()) <= TOL
    om, op = MDCTOracle(N, "vorbis", np.float64), PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    Xo = om.transform(x.astype(np.float64))
    to = op.tonality(Xo)
    thro = op.global_masking_threshold(Xo, to, 0.2)
    assert rel_peak(host(X), Xo) <= TOL
    assert tonality_err(host(t), to) <= 1.0
    assert rel_elem(host(thr), thro) <= TOL       # threshold of the GPU's own X and t (two rounding sources)
    xh = host(codec.decode(X))
    assert np.max(np.abs(xh[:, N:-N] - x)) <= LSB


# BASELINE configs[3]: Bark spreading cast as a band x band MFMA contraction, bf16.  Tolerances, stated separately from the
# float32 path's: split-bfloat16 operands (hi + lo parts, four partial products) keep the 1e-4 bar; plain bfloat16
# operands (8 mantissa bits) are held to 5e-3 on the thresholds.
SPREAD_TOL = {"f32": TOL, "bf16x2_mfma": TOL, "bf16_mfma": 5e-3}


@pytest.mark.parametrize("spreading", ["bf16x2_mfma", "bf16_mfma"])
@pytest.mark.parametrize("sr,N,B,F", [(48000, 2048, 3, 5), (48000, 1024, 2, 7), (44100, 1024, 1, 3), (96000, 2048, 2, 2)])
def test_mfma_spreading_vs_oracle(spreading, sr, N, B, F):
    rng = np.random.default_rng(N + F)
    env = np.logspace(-5, 0, N).reshape(1, 1, N, 1)
    X = (rng.uniform(-1, 1, (B, F, N, 2)) * env * rng.uniform(1e-3, 1, (B, F, 1, 2))).astype(np.float32)
    X[0, 0, :, 0] = 0.0
    X[0, 1, :, 1] = 0.0
    X[0, 1, 100, 1] = 0.5                                   # a single tone: the spreading function alone shapes thr
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, spreading=spreading)
    assert p.is_fast() and p.plan_spreading() == spreading
    o = PsychoOracle(sr, N, 64, compute_dtype=np.float64)
    to = o.tonality(X.astype(np.float64))
    for drown in (0.0, 0.5, 1.0):
        thr = host(p.global_masking_threshold(dev(X), dev(to.astype(np.float32)), drown))
        assert rel_elem(thr, o.global_masking_threshold(X.astype(np.float64), to, drown)) <= SPREAD_TOL[spreading]


@pytest.mark.parametrize("spreading", ["bf16x2_mfma", "bf16_mfma"])
@pytest.mark.parametrize("N,B,K", [(2048, 3, 5), (1024, 2, 37)])
def test_mfma_spreading_fused_encode(spreading, N, B, K):
    """config 4 end to end: N = 2048 long-window MDCT + masking with the MFMA contraction, against the oracle and
    against the float32 form on the same input"""
    rng = np.random.default_rng(B + K)
    x = rng.uniform(-1, 1, (B, K * N, 2)).astype(np.float32)
    x[0, : 2 * N, 0] *= 1e-3
    codec = audiocodec_amd.AudioCodec(48000, N, spreading=spreading)
    plain = audiocodec_amd.AudioCodec(48000, N, spreading="f32")
    assert plain.psy.plan_spreading() == "f32"
    assert audiocodec_amd.AudioCodec(48000, N).psy.plan_spreading() == "bf16x2_mfma"      # the default where it applies
    assert audiocodec_amd.PsychoacousticModel(48000, 512).plan_spreading() == "f32"       # ... and elsewhere
    X, t, thr = codec.encode(dev(x), drown=0.2)
    X0, t0, thr0 = plain.encode(dev(x), drown=0.2)
    # only the spreading product differs -- and, at N = 1024, the pre-twiddles, which these kernels rebuild from the
    # post-twiddles (one more float32 rounding) to make room in LDS for the bf16 tiles
    assert float((X - X0).abs().max()) <= 2e-7 and float((t - t0).abs().max()) <= 2e-6
    tol = SPREAD_TOL[spreading]
    assert float(((thr - thr0).abs() / thr0).max()) <= tol
    thru = codec.psy.global_masking_threshold(X, t, 0.2)    # stand-alone kernel, same form of the product
    assert float(((thr - thru).abs() / thru).max()) <= TOL
    om, op = MDCTOracle(N, "vorbis", np.float64), PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    Xo = om.transform(x.astype(np.float64))
    to = op.tonality(Xo)
    assert rel_elem(host(thr), op.global_masking_threshold(Xo, to, 0.2)) <= tol
    if spreading == "bf16_mfma":
        assert float(((thr - thr0).abs() / thr0).max()) > 1e-4   # the bf16 rounding is really there


def test_mfma_spreading_scope():
    """other channel counts and 16-bit PCM keep the float32 product; plans outside the wave-level tier refuse"""
    N = 1024
    rng = np.random.default_rng(5)
    x = rng.uniform(-1, 1, (2, 3 * N, 3)).astype(np.float32)
    a = audiocodec_amd.AudioCodec(48000, N, spreading="bf16_mfma").encode(dev(x))
    b = audiocodec_amd.AudioCodec(48000, N, spreading="f32").encode(dev(x))
    assert all(torch.equal(u, v) for u, v in zip(a, b))
    with pytest.raises(RuntimeError, match="wave-level"):
        audiocodec_amd.PsychoacousticModel(48000, filter_bands_n=512, spreading="bf16x2_mfma").tonality(dev(np.zeros((1, 1, 512, 1), np.float32)))
    with pytest.raises(ValueError):
        audiocodec_amd.PsychoacousticModel(48000, spreading="fp8")


def test_db_and_noise(golden, path):
    g = golden("db_utils")
    p = audiocodec_amd.PsychoacousticModel(48000)
    a = dev(g["a"])
    np.testing.assert_allclose(host(p.amplitude_to_dB(a)), g["dB_ref64"], rtol=0, atol=2e-4)
    np.testing.assert_allclose(host(p.amplitude_to_dB_norm(a)), g["dBn_ref64"], rtol=0, atol=2e-6)
    X = torch.zeros(4, 8, 1024, 2, device="cuda")
    thr = torch.full_like(X, 0.3)
    y = p.add_noise(X, thr, seed=5)
    assert abs(float(y.mean())) < 5e-4 and abs(float(y.std()) - 0.05) < 5e-4
    assert torch.equal(y, p.add_noise(X, thr, seed=5)) and not torch.equal(y, p.add_noise(X, thr, seed=6))
    assert float((y.abs() > 0.3).float().mean()) < 0.01


# ---- streaming ------------------------------------------------------------------------------------------

@pytest.mark.parametrize("N,C,chunks", [(1024, 2, (3, 1, 4, 2)), (256, 1, (2, 2, 5)), (1024, 1, (5, 3)), (960, 2, (2, 3, 1)),
                                        (2048, 2, (2, 5, 1)), (128, 2, (3, 9, 1, 4)), (128, 1, (11, 2)), (512, 2, (1, 3, 2)),
                                        (64, 2, (5, 17, 1, 16)), (64, 1, (33, 2)),
                                        # strips of the 16-byte LDS-FFT wave kernels: longer / shorter than a strip, state in and out
                                        (480, 2, (40, 3, 1)), (120, 2, (70, 1)), (16, 2, (100, 3)), (48, 2, (5,)), (960, 2, (33,)), (1920, 2, (3, 9)),
                                        (4096, 2, (2, 1, 3)), (1536, 1, (2, 3)), (8192, 2, (2, 1)), (960, 3, (5, 2)), (480, 6, (40, 3)), (1920, 5, (2, 2)), (480, 1, (40, 3, 1)), (1920, 1, (3, 2)), (24, 1, (7, 60))])
def test_streaming_equals_one_shot(path, N, C, chunks):
    B, K = 2, sum(chunks)
    x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
    m = audiocodec_amd.MDCTransformer(N)
    X_full = m.transform(x)                       # [B, K+1, N, C]
    st = audiocodec_amd.StreamingMDCT(m, B, C)
    outs, pos = [], 0
    for k in chunks:
        outs.append(st.transform_chunk(x[:, pos * N:(pos + k) * N]))
        pos += k
    outs.append(st.transform_chunk(torch.zeros(B, N, C, device="cuda")))   # flush: tail frame
    X_stream = torch.cat(outs, dim=1)
    assert torch.equal(X_stream, X_full)
    x_full = m.inverse_transform(X_full)          # [B, (K+2) N, C]
    st.reset()
    outs, pos = [], 0
    for k in chunks + (1,):
        outs.append(st.inverse_chunk(X_full[:, pos:pos + k]))
        pos += k
    outs.append(st.inverse_chunk(torch.zeros(B, 1, N, C, device="cuda")))
    x_stream = torch.cat(outs, dim=1)
    assert float((x_stream - x_full).abs().max()) <= 1e-6
    assert float((x_stream[:, N:-N] - x).abs().max()) <= LSB
    st.close()


@pytest.mark.parametrize("N,C,chunks", [(1024, 2, (3, 1, 5)), (1024, 1, (2, 4)), (1024, 3, (4, 2)), (2048, 2, (2, 3)),
                                        (2048, 1, (3, 1)), (256, 2, (5, 2)), (1024, 2, (256, 17))])
def test_streaming_encode_equals_one_shot(path, N, C, chunks):
    """ac_stream_encode (MDCT + tonality + masking threshold on a chunk, analysis state kept on the device and written
    by the kernel itself) returns, chunk by chunk, exactly the frames of the one-shot encode -- bit for bit."""
    B, K = 2, sum(chunks)
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    codec = audiocodec_amd.AudioCodec(48000, N)
    X_full, t_full, thr_full = codec.encode(x, drown=0.25)            # [B, K+1, ...]
    st = codec.stream(B, C)
    Xs, ts, thrs, pos = [], [], [], 0
    for k in chunks:
        X, t, thr = st.encode_chunk(x[:, pos * N:(pos + k) * N], drown=0.25)
        assert tuple(X.shape) == (B, k, N, C) and tuple(t.shape) == (B, k, 1, C) and tuple(thr.shape) == (B, k, N, C)
        Xs.append(X), ts.append(t), thrs.append(thr)
        pos += k
    X, t, thr = st.encode_chunk(torch.zeros(B, N, C, device="cuda"), drown=0.25)   # flush: the tail frame
    Xs.append(X), ts.append(t), thrs.append(thr)
    assert torch.equal(torch.cat(Xs, dim=1), X_full)
    assert torch.equal(torch.cat(ts, dim=1), t_full)
    assert torch.equal(torch.cat(thrs, dim=1), thr_full)
    # into caller-owned tensors, after a reset: the same again
    st.reset()
    out = (torch.empty_like(Xs[0]), torch.empty_like(ts[0]), torch.empty_like(thrs[0]))
    got = st.encode_chunk(x[:, :chunks[0] * N], drown=0.25, out=out)
    assert got[0] is out[0] and torch.equal(out[0], Xs[0]) and torch.equal(out[1], ts[0]) and torch.equal(out[2], thrs[0])
    with pytest.raises(ValueError):
        st.encode_chunk(x[:, :N + 1])
    with pytest.raises(ValueError):
        audiocodec_amd.StreamingMDCT(codec.mdct, B, C).encode_chunk(x[:, :N])     # no masking model attached
    st.close()


@pytest.mark.parametrize("N,k,K", [(1024, 16, 75), (2048, 8, 24), (256, 32, 64)])
def test_stream_run_equals_one_shot(path, N, k, K):
    """ac_stream_run (a resident signal through the stream in chunks with one library call) = the one-shot encode /
    decode, bit for bit on the analysis side; the last chunk may be shorter"""
    C = 2
    g = torch.Generator(device="cuda").manual_seed(K)
    x = torch.empty(1, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    codec = audiocodec_amd.AudioCodec(48000, N)
    X_full, t_full, thr_full = codec.encode(x, drown=0.1)
    xh_full = codec.decode(X_full)
    st = codec.stream(1, C)
    X, t, thr, xh = st.run(x, k, drown=0.1)
    torch.cuda.synchronize()
    assert torch.equal(X, X_full[:, :K]) and torch.equal(t, t_full[:, :K]) and torch.equal(thr, thr_full[:, :K])
    assert float((xh - xh_full[:, :K * N]).abs().max()) <= 1e-6      # block 0 is the leading half-aliased block
    assert float((xh[:, N:] - x[:, :-N]).abs().max()) <= LSB
    # continues where it stopped: the next call sees the state the run left behind
    X2, t2, thr2, xh2 = st.run(torch.zeros(1, N, C, device="cuda"), 1, synthesis=False, drown=0.1)
    assert xh2 is None and torch.equal(X2, X_full[:, K:]) and torch.equal(thr2, thr_full[:, K:])
    # list-of-chunks form for a batch of streams
    B = 3
    xb = torch.empty(B, 4 * k * N, C, device="cuda").uniform_(-1, 1, generator=g)
    stb = codec.stream(B, C)
    Xl, tl, thrl, xhl = stb.run([xb[:, i * k * N:(i + 1) * k * N].contiguous() for i in range(4)], k, drown=0.1)
    Xb, tb, thrb = codec.encode(xb, drown=0.1)
    assert torch.equal(torch.cat(Xl, dim=1), Xb[:, :-1]) and torch.equal(torch.cat(thrl, dim=1), thrb[:, :-1])
    assert torch.equal(torch.cat(tl, dim=1), tb[:, :-1])
    assert float((torch.cat(xhl, dim=1)[:, N:] - xb[:, :-N]).abs().max()) <= LSB
    st.close(), stb.close()


@pytest.mark.parametrize("N,C,k,K,masking", [(1024, 2, 16, 80, False), (1024, 1, 8, 40, False), (2048, 2, 8, 32, False),
                                              (2048, 1, 4, 20, False), (1024, 2, 256, 1024, True), (1024, 2, 7, 30, True)])
def test_stream_run_duplex_equals_the_chain(N, C, k, K, masking):
    """ac_stream_run with synthesis on small chunks: analysis of chunk i + 1 and synthesis of chunk i share one launch
    (k_duplex_fast).  Same results, bit for bit, as the chunk-by-chunk calls of the streaming API -- and a caller that
    hands over ONE X buffer for all chunks (so the two halves would collide) gets the dependent chain, same results."""
    g = torch.Generator(device="cuda").manual_seed(N + K + C)
    x = torch.empty(1, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    codec = audiocodec_amd.AudioCodec(48000, N)
    st = codec.stream(1, C)
    X, t, thr, xh = st.run(x, k, masking=masking, drown=0.2)
    ref = codec.stream(1, C)
    Xs, ts, thrs, xs = [], [], [], []
    for i in range(0, K, k):
        chunk = x[:, i * N:(i + k) * N]
        if masking:
            Xc, tc, thc = ref.encode_chunk(chunk, drown=0.2)
            ts.append(tc), thrs.append(thc)
        else:
            Xc = ref.transform_chunk(chunk)
        Xs.append(Xc), xs.append(ref.inverse_chunk(Xc))
    assert torch.equal(X, torch.cat(Xs, dim=1)) and torch.equal(xh, torch.cat(xs, dim=1))
    if masking:
        assert torch.equal(t, torch.cat(ts, dim=1)) and torch.equal(thr, torch.cat(thrs, dim=1))
    assert float((xh[:, N:] - x[:, :-N]).abs().max()) <= LSB
    # one X / t / thr buffer for every chunk, through the C ABI: the last chunk's spectrum, the whole signal's PCM
    import ctypes
    from audiocodec_amd import _host
    lib = _lib.load()
    n = K // k
    st2 = codec.stream(1, C)
    Xone = torch.empty(1, k, N, C, device="cuda")
    tone, throne = torch.empty(1, k, 1, C, device="cuda"), torch.empty(1, k, N, C, device="cuda")
    xh2 = torch.empty_like(x)
    arr = lambda ps: (ctypes.c_void_p * n)(*ps)   # noqa: E731
    _lib.check(lib.ac_stream_run(st2._handle, codec.psy._plan(x.device) if masking else None, n, k,
                                 arr([x[:, i * k * N:].data_ptr() for i in range(n)]), arr([Xone.data_ptr()] * n),
                                 arr([tone.data_ptr()] * n) if masking else None, arr([throne.data_ptr()] * n) if masking else None,
                                 arr([xh2[:, i * k * N:].data_ptr() for i in range(n)]), 0.2, _host.stream_ptr(x.device)))
    torch.cuda.synchronize()
    assert torch.equal(xh2[:, :n * k * N], xh[:, :n * k * N]) and torch.equal(Xone, Xs[n - 1])
    st.close(), ref.close(), st2.close()


@pytest.mark.parametrize("N,C,k,K,masking", [(1024, 2, 16, 83, True), (1024, 1, 8, 40, False), (2048, 2, 8, 24, False),
                                              (256, 2, 32, 96, True)])
def test_stream_run_as_a_replayed_graph(N, C, k, K, masking):
    """StreamingMDCT.run(graph=True): the call is captured once (odd chunk counts included: ac_stream_run leaves the
    state buffers where it found them) and replayed on new contents of the same input buffer; results equal the plain
    calls bit for bit, the stream state carries over from call to call, reset() still works"""
    g = torch.Generator(device="cuda").manual_seed(N + K)
    codec = audiocodec_amd.AudioCodec(48000, N)
    xbuf = torch.empty(1, K * N, C, device="cuda")
    sg, sp = codec.stream(1, C), codec.stream(1, C)
    for call in range(3):
        xbuf.uniform_(-1, 1, generator=g)
        got = sg.run(xbuf, k, masking=masking, drown=0.1, graph=True)
        ref = sp.run(xbuf.clone(), k, masking=masking, drown=0.1)       # continues its own stream the same way
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert (a is None and b is None) or torch.equal(a, b), call
    assert len(sg._graphs) == 1
    sg.reset(), sp.reset()
    xbuf.uniform_(-1, 1, generator=g)
    got, ref = sg.run(xbuf, k, masking=masking, drown=0.1, graph=True), sp.run(xbuf.clone(), k, masking=masking, drown=0.1)
    assert torch.equal(got[0], ref[0]) and torch.equal(got[3], ref[3])
    X_full = codec.encode(xbuf, drown=0.1)[0] if masking else codec.mdct.transform(xbuf)   # (the fused kernel's own rounding)
    assert torch.equal(got[0], X_full[:, :K])
    sg.close(), sp.close()


def test_stream_run_with_unaligned_chunk_boundaries():
    """run() on a tensor whose chunk boundaries are not 16-byte aligned (filters_n * channels_n * 4 bytes per block not a
    multiple of 16: N = 6 mono, N = 10 with three channels) goes through per-chunk tensors instead of failing with
    AC_EINVAL; result = the one-shot transform / a delayed copy of the input."""
    for N, C, k, K in ((6, 1, 1, 5), (10, 3, 3, 7), (6, 1, 2, 5)):
        mdct = audiocodec_amd.MDCTransformer(N)
        x = torch.rand(1, K * N, C, device="cuda") * 2 - 1
        st = audiocodec_amd.StreamingMDCT(mdct, 1, C)
        X, t, thr, xh = st.run(x, k, masking=False)
        ref = mdct.transform(x)
        assert t is None and thr is None
        assert float((X - ref[:, :K]).abs().max()) <= 1e-6
        assert float((xh[:, N:] - x[:, :-N]).abs().max()) <= LSB
        st.close()


def test_graph_replay_after_chunk_calls_finds_the_state():
    """A captured run(graph=True) addresses the stream's home state buffers; chunk calls swap the double-buffered state an
    odd number of times in between and reset() zeroes it: the next replay must still continue the signal where the chunk
    calls left it (it used to read the stale buffer silently).  Reference = a second stream fed the same
    sequence through the plain (un-captured) calls."""
    N, C, k, K = 1024, 2, 4, 12
    g = torch.Generator(device="cuda").manual_seed(77)
    codec = audiocodec_amd.AudioCodec(48000, N)
    xbuf = torch.empty(1, K * N, C, device="cuda")
    sg, sp = codec.stream(1, C), codec.stream(1, C)

    def both_run():
        xbuf.uniform_(-1, 1, generator=g)
        got = sg.run(xbuf, k, masking=True, drown=0.0, graph=True)
        ref = sp.run(xbuf.clone(), k, masking=True, drown=0.0)
        torch.cuda.synchronize()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)

    both_run()                                           # capture
    for odd in (1, 3):                                   # an odd number of chunk calls: the state sits in the other buffer
        for _ in range(odd):
            xc = torch.empty(1, 2 * N, C, device="cuda").uniform_(-1, 1, generator=g)
            Xa, Xb = sg.transform_chunk(xc), sp.transform_chunk(xc)
            assert torch.equal(Xa, Xb)
            assert torch.equal(sg.inverse_chunk(Xa), sp.inverse_chunk(Xb))
        both_run()                                       # replay
    sg.transform_chunk(xbuf[:, :N].contiguous()), sp.transform_chunk(xbuf[:, :N].contiguous())
    sg.reset(), sp.reset()                               # reset issued while the state sat in the other buffer
    both_run()
    assert len(sg._graphs) == 1
    # the same through the C ABI alone: ac_stream_settle is idempotent and leaves results unchanged
    lib = audiocodec_amd._lib.load()
    for _ in range(2):
        audiocodec_amd._lib.check(lib.ac_stream_settle(sg._handle, None))
    both_run()
    sg.close(), sp.close()


def test_db_backward_on_a_permuted_input():
    """amplitude_to_dB(_norm) of a dense, permuted view that requires a gradient: the gradient comes back in the view's
    logical order (empty_like kept the permuted strides while the kernel wrote linearly)."""
    psy = audiocodec_amd.PsychoacousticModel(48000, 1024)
    g = torch.Generator(device="cuda").manual_seed(5)
    base = (torch.rand(3, 1024, 5, 2, device="cuda", generator=g) * 2 - 1)
    w = torch.rand(3, 5, 1024, 2, device="cuda", generator=g)
    for norm in (False, True):
        a = base.clone().requires_grad_(True)
        view = a.permute(0, 2, 1, 3)                      # [3, 5, 1024, 2], not contiguous
        out = (psy.amplitude_to_dB_norm(view) if norm else psy.amplitude_to_dB(view))
        (out * w).sum().backward()
        b = base.clone().requires_grad_(True)
        vb = b.permute(0, 2, 1, 3).contiguous()
        outb = (psy.amplitude_to_dB_norm(vb) if norm else psy.amplitude_to_dB(vb))
        (outb * w).sum().backward()
        assert torch.equal(out, outb) and torch.equal(a.grad, b.grad)


def test_codec_encode_decode_are_differentiable():
    """AudioCodec.encode / encode_ex / decode on an input that requires a gradient: the differentiable composition (the
    reference's op chain is differentiable), same values as the fused launch within the tolerance of the un-fused path,
    gradients equal those of the explicit composition; the *_into forms refuse such tensors instead of dropping the graph"""
    N, B, K, C = 1024, 2, 3, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    w = torch.rand(B, K + 1, N, C, device="cuda", generator=g)
    xa = x.clone().requires_grad_(True)
    X, t, thr = codec.encode(xa, drown=0.2)
    assert X.requires_grad and t.requires_grad and thr.requires_grad
    Xf, tf, thrf = codec.encode(x, drown=0.2)
    assert float((X.detach() - Xf).abs().max()) <= 1e-6 and float(((thr.detach() - thrf).abs() / thrf).max()) <= 1e-5
    xh = codec.decode(X)
    assert xh.requires_grad
    ((thr * w).sum() + (xh ** 2).sum()).backward()
    xb = x.clone().requires_grad_(True)
    Xb = codec.mdct.transform(xb)
    tb = codec.psy.tonality(Xb)
    thrb = codec.psy.global_masking_threshold(Xb, tb, 0.2)
    ((thrb * w).sum() + (codec.mdct.inverse_transform(Xb) ** 2).sum()).backward()
    assert torch.equal(xa.grad, xb.grad) and float(xa.grad.abs().max()) > 0
    ex = codec.encode_ex(x.clone().requires_grad_(True), drown=0.2, noise_seed=5, db_norm=True)
    assert all(v.requires_grad for v in ex)
    assert torch.equal(ex[3].detach(), codec.psy.add_noise(ex[0].detach(), ex[2].detach(), seed=5))
    with pytest.raises(ValueError, match="gradient"):
        codec.encode_into(xa, Xf, tf, thrf)
    with torch.no_grad():
        codec.encode_into(xa, Xf, tf, thrf)          # explicit no_grad: fine
    st = codec.stream(B, C)
    with pytest.raises(ValueError, match="gradient"):
        st.transform_chunk(xa[:, :N])
    st.close()


def test_misaligned_tensors_are_copied_or_refused():
    """The kernels move rows with 16-byte accesses.  An input view that starts inside an allocation is copied by the Python
    layer (same results); a caller-owned OUTPUT at such an address is refused (ValueError), and the C ABI itself refuses
    any misaligned tensor (AC_EINVAL) instead of launching on it."""
    N, B, K, C = 1024, 2, 3, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    buf = torch.rand(B * K * N * C + 8, device="cuda") * 2 - 1
    x_off = buf[1:1 + B * K * N * C].view(B, K * N, C)
    assert x_off.is_contiguous() and x_off.data_ptr() % 16 == 4
    X, t, thr = codec.encode(x_off)
    Xa, ta, thra = codec.encode(x_off.clone())
    assert torch.equal(X, Xa) and torch.equal(t, ta) and torch.equal(thr, thra)
    assert torch.equal(codec.mdct.transform(x_off), codec.mdct.transform(x_off.clone()))
    Xbuf = torch.empty(B * (K + 1) * N * C + 8, device="cuda")
    X_off = Xbuf[1:1 + B * (K + 1) * N * C].view(B, K + 1, N, C)
    with pytest.raises(ValueError):
        codec.encode_into(x_off.clone(), X_off, ta, thra)
    lib = _lib.load()
    from audiocodec_amd import _host
    st = lib.ac_mdct_forward(codec.mdct._plan(buf.device), _host.ptr(x_off), _host.ptr(Xa), B, K, C, _host.stream_ptr(buf.device))
    assert st == -1 and b"aligned" in lib.ac_last_error()
    torch.cuda.synchronize()


def test_encode_decode_under_graph_capture():
    """The entry points only enqueue kernels on the caller's stream (no allocation, no synchronisation once the plans
    exist), so an encode + decode pair is capturable into a HIP graph; a replay on new input gives the eager result."""
    N, B, K, C = 1024, 4, 12, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    X, t = torch.empty(B, K + 1, N, C, device="cuda"), torch.empty(B, K + 1, 1, C, device="cuda")
    thr, xh = torch.empty_like(X), torch.empty(B, (K + 2) * N, C, device="cuda")
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):                      # plans are built here, outside the capture
        codec.encode_into(x, X, t, thr)
        codec.decode_into(X, xh)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        codec.encode_into(x, X, t, thr, drown=0.3)
        codec.decode_into(X, xh)
    x.uniform_(-1, 1, generator=g)                     # new input in the captured buffers
    graph.replay()
    torch.cuda.synchronize()
    Xe, te, thre = codec.encode(x, drown=0.3)
    assert torch.equal(X, Xe) and torch.equal(t, te) and torch.equal(thr, thre) and torch.equal(xh, codec.decode(Xe))
    assert float((xh[:, N:-N] - x).abs().max()) <= LSB


def test_full_size_properties(path):
    """BASELINE config 2 shape (B=256 stereo, K=46, N=1024): size-independent properties only."""
    if path == "generic":
        pytest.skip("O(N^2) kernels: full size covered on the fast path")
    N, B, K, C = 1024, 256, 46, 2
    g = torch.Generator(device="cuda").manual_seed(1234)
    x = torch.rand(B, K * N, C, device="cuda", generator=g) * 2 - 1
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(x)
    xh = codec.decode(X)
    assert float((xh[:, N:-N] - x).abs().max()) <= LSB
    assert torch.equal(torch.round(xh[:, N:-N] * 32767), torch.round(x * 32767)) or \
        float((torch.round(xh[:, N:-N] * 32767) - torch.round(x * 32767)).abs().max()) <= 1
    assert float(t.min()) >= -1e-6 and float(t.max()) <= 1.0
    assert float(thr.min()) >= 1e-7 * (1 - 1e-6) and bool(torch.isfinite(thr).all())
    # every clip is independent: a clip processed alone gives the same bits
    X1, t1, thr1 = codec.encode(x[17:18])
    assert torch.equal(X1, X[17:18]) and torch.equal(thr1, thr[17:18]) and torch.equal(t1, t[17:18])
    # parity on a sample of clips against the oracle
    om, op = MDCTOracle(N, "vorbis", np.float64), PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    xs = host(x[:2]).astype(np.float64)
    Xo = om.transform(xs)
    assert rel_peak(host(X[:2]), Xo) <= TOL
    assert rel_elem(host(thr[:2]), op.global_masking_threshold(Xo, op.tonality(Xo))) <= TOL


@pytest.mark.parametrize("N,C,spreading", [(1024, 2, None), (1024, 2, "f32"), (1024, 2, "bf16_mfma"), (2048, 2, None),
                                           (1024, 1, None), (256, 2, None)])
def test_encode_ex_equals_encode_plus_elementwise_tail(path, golden, N, C, spreading):
    """ac_encode_fused_ex (AC_EMIT_NOISY | AC_EMIT_DB_NORM): one launch for stereo float32 at N = 1024, the encode plus
    the two element-wise kernels elsewhere -- either way X, t, thr equal encode()'s and the tail equals add_noise /
    amplitude_to_dB_norm on them, bit for bit (same generator, same formula; psychoacoustic.py:150-167, :87-100)"""
    B, K = 3, 7
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    x[0, : 2 * N] *= 1e-4
    codec = audiocodec_amd.AudioCodec(48000, N, spreading=spreading) if N in (1024, 2048) else audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(x, drown=0.3)
    X2, t2, thr2, noisy, dbn = codec.encode_ex(x, drown=0.3, noise_seed=1234567, db_norm=True)
    assert torch.equal(X, X2) and torch.equal(t, t2) and torch.equal(thr, thr2)
    assert torch.equal(noisy, codec.psy.add_noise(X2, thr2, seed=1234567))
    assert torch.equal(dbn, codec.psy.amplitude_to_dB_norm(X2))
    # one flag at a time, and none
    a = codec.encode_ex(x, drown=0.3, noise_seed=99)
    assert a[4] is None and torch.equal(a[3], codec.psy.add_noise(a[0], a[2], seed=99)) and torch.equal(a[2], thr2)
    b = codec.encode_ex(x, drown=0.3, db_norm=True)
    assert b[3] is None and torch.equal(b[4], dbn) and torch.equal(b[0], X) and torch.equal(b[2], thr2)
    c = codec.encode_ex(x, drown=0.3)
    assert c[3] is None and c[4] is None and torch.equal(c[2], thr)
    # statistics of the fused noise: (noisy - X) / thr ~ Normal(0, 1/6)
    z = ((noisy - X2) / thr2).double()
    assert abs(float(z.mean())) < 3e-3 and abs(float(z.std()) - 1.0 / 6.0) < 3e-3
    assert float((z.abs() > 0.5).double().mean()) < 0.006          # 3 sigma: 0.27 %
    # exact-formula check of the dB normalisation against the reference-generated values
    gd = golden("db_utils")
    p = codec.psy
    np.testing.assert_allclose(host(p.amplitude_to_dB_norm(dev(gd["a"]))), gd["dBn_ref64"], rtol=0, atol=2e-6)
    assert float(dbn.min()) >= 0.0 and float(dbn.max()) <= 1.0 + 1e-6


def test_autograd_of_add_noise_and_db(path):
    """add_noise and amplitude_to_dB(_norm) are differentiable like the reference's plain TF op chains
    (psychoacoustic.py:150-167, 71-100): gradients against torch.autograd on a torch restatement"""
    p = audiocodec_amd.PsychoacousticModel(48000, 1024)
    g = torch.Generator(device="cuda").manual_seed(2)
    X = (torch.rand(2, 3, 1024, 2, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    thr = (torch.rand(2, 3, 1024, 2, device="cuda", generator=g) * 0.1 + 1e-3).requires_grad_(True)
    w = torch.rand(2, 3, 1024, 2, device="cuda", generator=g)
    y = p.add_noise(X, thr, seed=77)
    assert y.requires_grad
    (y * w).sum().backward()
    n = (p.add_noise(torch.zeros_like(X), torch.ones_like(thr), seed=77)).detach()      # the normals of seed 77 (x 1/6)
    assert torch.equal(X.grad, w)
    assert float((thr.grad - w * n).abs().max()) <= 1e-7
    # only one input needs a gradient
    y2 = p.add_noise(X.detach(), thr.detach().requires_grad_(True), seed=77)
    assert y2.requires_grad and torch.equal(y2, y.detach())
    for norm in (False, True):
        a = (torch.rand(4, 5, 64, 2, device="cuda", generator=g) * 2 - 1)
        a[0, 0, :4] = 0.0                                   # inside the clamp: gradient 0
        a[0, 1, :4] = 1e-8
        a = a.requires_grad_(True)
        f = p.amplitude_to_dB_norm if norm else p.amplitude_to_dB
        d = f(a)
        wa = torch.rand_like(a)
        (d * wa).sum().backward()
        ad = a.detach().double().requires_grad_(True)
        dd = 10.0 * torch.log10(torch.clamp(ad * ad, min=1e-14)) + 120.0
        if norm:
            dd = (dd + 20.0) / 140.0
        (dd * wa.double()).sum().backward()
        assert float((d.detach().double() - dd.detach()).abs().max()) <= (2e-6 if norm else 2e-4)
        ref = ad.grad
        err = (a.grad.double() - ref).abs() / (ref.abs() + 1e-3)
        assert float(err.max()) <= 1e-5
        assert float(a.grad[0, 0, :4].abs().max()) == 0.0 and float(a.grad[0, 1, :4].abs().max()) == 0.0
    with pytest.raises(NotImplementedError):
        p64 = audiocodec_amd.PsychoacousticModel(48000, 64, compute_dtype=torch.float64)
        p64.amplitude_to_dB(torch.rand(1, 1, 64, 1, device="cuda", dtype=torch.float64).requires_grad_(True))


@pytest.mark.parametrize("sr,N,M,B,F,C", [(48000, 512, 64, 3, 5, 2), (48000, 256, 64, 2, 7, 2), (16000, 512, 64, 3, 4, 1),
                                          (44100, 256, 48, 2, 5, 2), (48000, 1024, 32, 2, 3, 2), (96000, 512, 64, 1, 3, 1),
                                          (48000, 256, 17, 5, 2, 1), (8000, 512, 64, 2, 3, 2),
                                          # filter_bands_n that are not multiples of 128: partly filled granule registers
                                          (48000, 960, 64, 2, 4, 2), (48000, 480, 64, 3, 3, 1), (44100, 576, 48, 2, 3, 2),
                                          (16000, 240, 32, 2, 5, 1), (48000, 120, 20, 3, 4, 2),
                                          (48000, 30, 8, 2, 2, 1), (48000, 1000, 64, 1, 3, 2),
                                          # above 1024 bins: 16 / 32 granule registers per lane, W_inv entries read from global memory
                                          (48000, 1920, 64, 2, 3, 2), (48000, 4096, 64, 1, 3, 2), (44100, 2048, 48, 2, 2, 1),
                                          (48000, 1536, 48, 2, 2, 1), (96000, 4096, 64, 1, 2, 1), (48000, 2880, 64, 3, 2, 2),
                                          # more than two channels: the pairs (c, c + 1), rows strided by the channel count
                                          (48000, 960, 64, 2, 3, 3), (48000, 480, 64, 2, 4, 6), (44100, 512, 48, 3, 3, 5),
                                          (48000, 1920, 64, 1, 2, 4), (48000, 120, 20, 2, 5, 7)])
def test_masking_model_general_band_layouts(sr, N, M, B, F, C):
    """filter_bands_n other than 1024 / 2048 with 64 bands (256, 512, 960, 480 ...; 1024 with other band counts): the
    wave-level masking kernels for general band layouts
    (ac_psy_mid.hip; a bin may overlap three or four Bark bands here) against the fp64 oracle, against the generic kernels,
    and the one-launch tonality + threshold of the un-fused encode against the two separate calls"""
    rng = np.random.default_rng(N + M + C)
    env = np.logspace(-5, 0, N).reshape(1, 1, N, 1)
    X = (rng.uniform(-1, 1, (B, F, N, C)) * env * rng.uniform(1e-3, 1, (B, F, 1, C))).astype(np.float32)
    X[0, 0, :, 0] = 0.0
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    assert p.tier() == 1 and not p.is_fast()
    o = PsychoOracle(sr, N, M, compute_dtype=np.float64)
    Xd = dev(X)
    t = p.tonality(Xd)
    to = o.tonality(X.astype(np.float64))
    assert tonality_err(host(t), to) <= 1.0
    lib = _lib.load()
    for drown in (0.0, 0.4, 1.0):
        thr = p.global_masking_threshold(Xd, dev(to.astype(np.float32)), drown)
        assert rel_elem(host(thr), o.global_masking_threshold(X.astype(np.float64), to, drown)) <= TOL
        assert lib.ac_set_force_generic(1) == 0
        try:
            thrg = p.global_masking_threshold(Xd, dev(to.astype(np.float32)), drown)
            tg = p.tonality(Xd)
        finally:
            lib.ac_set_force_generic(0)
        assert float(((thr - thrg).abs() / thrg).max()) <= TOL and tonality_err(t, tg) <= 1.0
    # the un-fused encode at these sizes: transform, then tonality + threshold in one launch
    codec = audiocodec_amd.AudioCodec(sr, N, bark_bands_n=M)
    x = dev(rng.uniform(-1, 1, (B, 6 * N, C)).astype(np.float32))
    Xe, te, thre = codec.encode(x, drown=0.2)
    assert torch.equal(Xe, codec.mdct.transform(x))
    assert tonality_err(te, codec.psy.tonality(Xe)) <= 1e-3          # same arithmetic; another instantiation may round one ulp apart
    assert float(((thre - codec.psy.global_masking_threshold(Xe, te, 0.2)).abs() / thre).max()) <= 1e-6


def test_mdct_tiers():
    """ac_mdct_plan_tier: which kernels serve a size and channel count (so that a silent fall to a slower form shows)."""
    tier = lambda N, C=2, w="vorbis": audiocodec_amd.MDCTransformer(N, w).tier(C)
    assert [tier(N) for N in (64, 512, 1024, 2048)] == [3, 3, 3, 3] and tier(1024, 5) == 2 and tier(2048, 3) == 2
    assert [tier(N, C) for N in (16, 120, 960, 1920, 4096, 8192) for C in (1, 2, 3, 6)] == [2] * 24
    assert tier(1024, 2, "rect") == 2 and tier(256, 3) == 2          # what the wave-level kernels leave: instances too
    assert tier(30) == 1 and tier(90, 1) == 1                         # filters_n % 4 == 2: the run-time forms
    assert tier(7500) == 0 and tier(34) == 0 and tier(8190) == 0      # no plan / half not 5-smooth: O(N^2)


def test_fast_path_selection():
    """The wave-level kernels serve N = 64, 128, 256, 512 (several frames per wave), 1024 and 2048 with a Princen-Bradley window;
    everything else runs the LDS-FFT / generic kernels (the rectangular window's fold blocks are not rotations)."""
    _lib.load().ac_set_force_generic(0)
    assert audiocodec_amd.MDCTransformer(1024, "vorbis").is_fast()
    assert audiocodec_amd.MDCTransformer(1024, "sine").is_fast()
    assert not audiocodec_amd.MDCTransformer(1024, "rect").is_fast()
    assert audiocodec_amd.MDCTransformer(2048, "vorbis").is_fast()
    assert not audiocodec_amd.MDCTransformer(2048, "rect").is_fast()
    assert audiocodec_amd.MDCTransformer(512).is_fast() and audiocodec_amd.MDCTransformer(256, "sine").is_fast()
    # (the several-frames-per-wave kernels have a form with four coefficients per fold block: rectangular and
    # float32-precomputed windows run at wave level there too)
    assert audiocodec_amd.MDCTransformer(512, "rect").is_fast()
    assert audiocodec_amd.MDCTransformer(128).is_fast() and audiocodec_amd.MDCTransformer(128, "rect").is_fast()
    assert audiocodec_amd.MDCTransformer(64).is_fast() and audiocodec_amd.MDCTransformer(64, "rect").is_fast()
    assert audiocodec_amd.MDCTransformer(256, precompute_dtype=torch.float32).is_fast()
    assert not audiocodec_amd.MDCTransformer(1024, precompute_dtype=torch.float32).is_fast()
    assert not audiocodec_amd.MDCTransformer(32).is_fast() and not audiocodec_amd.MDCTransformer(4096).is_fast()
    assert audiocodec_amd.PsychoacousticModel(48000, 1024, 64).is_fast()
    assert audiocodec_amd.PsychoacousticModel(48000, 2048, 64).is_fast()
    assert not audiocodec_amd.PsychoacousticModel(48000, 512, 64).is_fast()
    assert not audiocodec_amd.PsychoacousticModel(48000, 1024, 32).is_fast()


@pytest.mark.parametrize("N,wt,B,K,C", [(512, "vorbis", 3, 9, 2), (512, "sine", 2, 1, 2), (512, "vorbis", 5, 40, 1),
                                        (512, "vorbis", 1, 2, 1), (256, "vorbis", 3, 11, 2), (256, "sine", 2, 1, 2),
                                        (256, "vorbis", 3, 70, 1), (256, "vorbis", 4, 4, 2), (256, "vorbis", 1, 3, 1),
                                        (512, "vorbis", 2, 130, 2), (256, "vorbis", 2, 6, 3), (512, "vorbis", 2, 5, 4),
                                        (128, "vorbis", 3, 13, 2), (128, "sine", 2, 1, 2), (128, "vorbis", 3, 150, 1),
                                        (128, "vorbis", 1, 7, 1), (128, "vorbis", 2, 300, 2), (128, "vorbis", 2, 9, 3),
                                        (64, "vorbis", 3, 21, 2), (64, "sine", 2, 1, 2), (64, "vorbis", 3, 301, 1),
                                        (64, "vorbis", 1, 15, 1), (64, "vorbis", 2, 16, 2), (64, "vorbis", 2, 600, 2),
                                        (64, "vorbis", 2, 9, 3)])
def test_short_frames_on_the_wave_level_kernels(N, wt, B, K, C):
    """filters_n = 512 / 256 / 64 (the reference's own test sizes, tests/test_mdctransformer.py:23,42) and 128: two / four /
    sixteen / eight frames per wave.  Frame counts that leave lane groups idle (K + 1 not a multiple of 2 / 4 / 8 / 16), odd mono clip counts, signals longer than a
    synthesis strip, single blocks; against the fp64 oracle and against the LDS-FFT tier of this library.  Other channel
    counts at these sizes take the LDS-FFT tier and must agree too."""
    rng = np.random.default_rng(N + 10 * K + C)
    x = rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    m = audiocodec_amd.MDCTransformer(N, window_type=wt)
    assert m.is_fast()
    o = MDCTOracle(N, wt, np.float64)
    xd = dev(x)
    X = m.transform(xd)
    Xo = o.transform(x.astype(np.float64))
    assert tuple(X.shape) == (B, K + 1, N, C)
    assert rel_peak(host(X), Xo) <= TOL and rel_l2(host(X), Xo) <= TOL
    xh = m.inverse_transform(X)
    assert tuple(xh.shape) == (B, (K + 2) * N, C)
    assert float((xh[:, N:-N] - xd).abs().max()) <= LSB
    assert np.max(np.abs(host(xh) - o.inverse_transform(Xo))) <= LSB
    lib = _lib.load()
    assert lib.ac_set_force_generic(1) == 0
    try:
        Xg = m.transform(xd)
        xg = m.inverse_transform(X)
    finally:
        lib.ac_set_force_generic(0)
    peak = Xg.abs().amax(dim=2, keepdim=True).clamp_min(1e-20)
    assert float(((X - Xg).abs() / peak).max()) <= TOL and float((xh - xg).abs().max()) <= 2e-6
    # clips are independent and frames depend on two blocks only
    if B > 1:
        assert torch.equal(m.transform(xd[1:2].contiguous()), X[1:2])
    if K >= 4:
        assert torch.equal(m.transform(xd[:, N:3 * N].contiguous())[:, 1], X[:, 2])


@pytest.mark.parametrize("wt", ["rect", None])
def test_rect_window_n1024_vs_oracle(wt):
    N, B, K, C = 1024, 2, 3, 2
    x = np.random.default_rng(5).uniform(-1, 1, (B, K * N, C)).astype(np.float32)
    m = audiocodec_amd.MDCTransformer(N, window_type=wt)
    o = MDCTOracle(N, "rect", np.float64)
    X = host(m.transform(dev(x)))
    Xo = o.transform(x.astype(np.float64))
    assert rel_peak(X, Xo) <= TOL and rel_l2(X, Xo) <= TOL
    xh = host(m.inverse_transform(dev(X)))
    xo = o.inverse_transform(Xo)
    # the rectangular window's synthesis bank is not orthogonal (2x2 blocks [[0, 1], [1, -1]]): its output is not confined
    # to [-1, 1], so 1 LSB is taken on the signal's own peak (2.9 here)
    print("rect inverse: max |dx| = %.2e, peak %.2f" % (np.max(np.abs(xh - xo)), np.max(np.abs(xo))))
    assert np.max(np.abs(xh - xo)) <= LSB * max(1.0, np.max(np.abs(xo)))


def test_streaming_config5_ten_minutes():
    """BASELINE config 5: 10 min of 48 kHz stereo fed in chunks of 256 blocks through the device-resident overlap state;
    every frame equals the one-shot transform and the round trip stays within 1 LSB."""
    N, C, K, k = 1024, 2, 28125, 256
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand(1, K * N, C, device="cuda", generator=g) * 2 - 1
    m = audiocodec_amd.MDCTransformer(N)
    X_full = m.transform(x)
    st = audiocodec_amd.StreamingMDCT(m, 1, C)
    outs = [st.transform_chunk(x[:, p * N:min(K, p + k) * N]) for p in range(0, K, k)]
    outs.append(st.transform_chunk(torch.zeros(1, N, C, device="cuda")))
    X_stream = torch.cat(outs, dim=1)
    assert torch.equal(X_stream, X_full)
    st.reset()
    outs = [st.inverse_chunk(X_stream[:, p:min(K + 1, p + k)]) for p in range(0, K + 1, k)]
    x_stream = torch.cat(outs, dim=1)                   # [1, (K+1) N, C]: block 0 is the leading half-aliased block
    assert float((x_stream[:, N:] - x).abs().max()) <= LSB
    st.close()


@pytest.mark.parametrize("N,wt", [(1024, "vorbis"), (2048, "sine"), (64, "vorbis")])
def test_autograd_of_the_filter_bank(path, N, wt):
    """transform / inverse_transform are differentiable (the reference is usable inside a training graph,
    psychoacoustic.py:311): the backward pass is the transposed bank, checked against <T x, g> = <x, T^T g>."""
    B, K, C = 2, 4, 2
    m = audiocodec_amd.MDCTransformer(N, window_type=wt)
    x = (torch.rand(B, K * N, C, device="cuda") * 2 - 1).requires_grad_(True)
    g = torch.randn(B, K + 1, N, C, device="cuda")
    X = m.transform(x)
    (X * g).sum().backward()
    lhs = float((X.detach().double() * g.double()).sum())
    rhs = float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-5 * max(1.0, abs(lhs))
    # independent check of the gradient itself: T^T g from the oracle's linear map applied to basis-free identity
    o = MDCTOracle(N, wt, np.float64)
    ref = o.inverse_transform(host(g).astype(np.float64))[:, N:-N] / (4.0 * N)
    assert np.max(np.abs(host(x.grad) - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
    Xv = torch.randn(B, K, N, C, device="cuda").requires_grad_(True)
    gy = torch.randn(B, (K + 1) * N, C, device="cuda")
    y = m.inverse_transform(Xv)
    (y * gy).sum().backward()
    lhs = float((y.detach().double() * gy.double()).sum())
    rhs = float((Xv.detach().double() * Xv.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))


@pytest.mark.parametrize("N,wt,pre,C", [(64, "rect", "float64", 2), (16, "rect", "float64", 3), (256, None, "float64", 1),
                                         (64, "vorbis", "float32", 2), (128, "sine", "float32", 1), (1024, "rect", "float64", 2),
                                         (960, "vorbis", "float32", 2)])
def test_autograd_where_the_synthesis_bank_is_not_the_transpose(path, N, wt, pre, C):
    """Backward of transform / inverse_transform for the rectangular window (mdctransformer.py:209-229: 2x2 fold blocks
    [[1, 1], [1, 0]], whose inverse is not their transpose) and for float32-precomputed windows: the transposed bank runs as
    the synthesis / analysis kernels on transposed fold coefficients (ac_mdct_plan_adjoint).  Reference = the ORACLE's
    linear maps: T and S assembled column by column from oracle.transform / inverse_transform on unit impulses (float64),
    then T^T g and S^T g -- no self-comparison.  N = 1024 / 960: inner products only (the dense maps would be GBs)."""
    K, B = 3, 2
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, precompute_dtype=pre)
    o = MDCTOracle(N, wt, np.float64, precompute_dtype=np.float32 if pre == "float32" else np.float64)
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = (torch.rand(B, K * N, C, device="cuda", generator=g) * 2 - 1).requires_grad_(True)
    gX = torch.randn(B, K + 1, N, C, device="cuda", generator=g)
    X = m.transform(x)
    (X * gX).sum().backward()
    lhs, rhs = float((X.detach().double() * gX.double()).sum()), float((x.detach().double() * x.grad.double()).sum())
    assert abs(lhs - rhs) <= 2e-5 * max(1.0, abs(lhs))
    Xv = torch.randn(B, K, N, C, device="cuda", generator=g).requires_grad_(True)
    gy = torch.randn(B, (K + 1) * N, C, device="cuda", generator=g)
    y = m.inverse_transform(Xv)
    (y * gy).sum().backward()
    lhs, rhs = float((y.detach().double() * gy.double()).sum()), float((Xv.detach().double() * Xv.grad.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * max(1.0, abs(lhs))
    if N > 256:
        return
    eye = np.eye(K * N).reshape(K * N, K * N, 1)                                   # unit impulses as a batch of mono signals
    T = o.transform(eye).reshape(K * N, (K + 1) * N)                               # row i = T e_i  ->  T^T g = T_rows @ g
    gx_ref = np.einsum("if,bfc->bic", T, host(gX).astype(np.float64).reshape(B, (K + 1) * N, C))
    assert np.max(np.abs(host(x.grad) - gx_ref)) <= 2e-5 * max(1.0, np.max(np.abs(gx_ref)))
    eyeX = np.eye(K * N).reshape(K * N, K, N, 1)
    S = o.inverse_transform(eyeX).reshape(K * N, (K + 1) * N)                      # row i = S e_i
    gX_ref = np.einsum("is,bsc->bic", S, host(gy).astype(np.float64)).reshape(B, K, N, C)
    assert np.max(np.abs(host(Xv.grad) - gX_ref)) <= 1e-4 * max(1.0, np.max(np.abs(gX_ref)))


@pytest.mark.parametrize("dt", ["float64", "bfloat16", "float16"])
@pytest.mark.parametrize("N,wt,C", [(64, "vorbis", 2), (96, "rect", 3), (1024, "vorbis", 2), (960, "sine", 1)])
def test_autograd_of_the_filter_bank_in_every_dtype(N, wt, C, dt):
    """Backward of transform / inverse_transform for float64, bfloat16 and float16 tensors (the reference's op chain is
    differentiable in every dtype it accepts, mdctransformer.py:31-35): the transposed bank (ac_mdct_plan_adjoint) through the
    typed kernels.  Reference = the ORACLE's linear maps assembled from unit impulses in float64 (N <= 96), and the
    inner-product identity <T x, g> = <x, T^T g> with T x from the oracle at every size -- no self-comparison.  Tolerances: the
    storage type's rounding (float64 1e-10; bfloat16 2^-8, float16 2^-10 of the gradient's peak, plus its accumulation)."""
    tdt = getattr(torch, dt)
    tol = {"float64": 1e-10, "bfloat16": 1.2e-2, "float16": 2e-3}[dt]
    K, B = 3, 2
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, compute_dtype=tdt)
    o = MDCTOracle(N, wt, np.float64)
    g = torch.Generator(device="cuda").manual_seed(N + C)
    x = (torch.rand(B, K * N, C, device="cuda", generator=g, dtype=torch.float64) * 2 - 1).to(tdt).requires_grad_(True)
    gX = torch.randn(B, K + 1, N, C, device="cuda", generator=g, dtype=torch.float64).to(tdt)
    X = m.transform(x)
    (X * gX).sum().backward()
    assert x.grad.dtype == tdt
    x64, gX64 = host(x.detach().double()), host(gX.double())
    lhs = float(np.sum(o.transform(x64) * gX64))                       # <T x, g> with the oracle's T x
    rhs = float(np.sum(x64 * host(x.grad.double())))                   # <x, T^T g> with the kernels' T^T g
    assert abs(lhs - rhs) <= 40 * tol * max(1.0, float(np.sqrt(np.sum(gX64 ** 2) * np.sum(o.transform(x64) ** 2))) / 10)
    Xv = torch.randn(B, K, N, C, device="cuda", generator=g, dtype=torch.float64).to(tdt).requires_grad_(True)
    gy = torch.randn(B, (K + 1) * N, C, device="cuda", generator=g, dtype=torch.float64).to(tdt)
    y = m.inverse_transform(Xv)
    (y * gy).sum().backward()
    assert Xv.grad.dtype == tdt
    if N > 96:
        return
    eye = np.eye(K * N).reshape(K * N, K * N, 1)
    T = o.transform(eye).reshape(K * N, (K + 1) * N)
    gx_ref = np.einsum("if,bfc->bic", T, gX64.reshape(B, (K + 1) * N, C))
    assert np.max(np.abs(host(x.grad.double()) - gx_ref)) <= tol * max(1.0, np.max(np.abs(gx_ref)))
    eyeX = np.eye(K * N).reshape(K * N, K, N, 1)
    S = o.inverse_transform(eyeX).reshape(K * N, (K + 1) * N)
    gX_ref = np.einsum("is,bsc->bic", S, host(gy.double())).reshape(B, K, N, C)
    assert np.max(np.abs(host(Xv.grad.double()) - gX_ref)) <= 4 * tol * max(1.0, np.max(np.abs(gX_ref)))


@pytest.mark.parametrize("dt", ["float64", "bfloat16"])
@pytest.mark.parametrize("sr,N,M,C,drown", [(48000, 1024, 64, 2, 0.0), (44100, 256, 48, 3, 0.3), (48000, 960, 64, 1, 0.0)])
def test_autograd_of_the_masking_model_in_other_dtypes(sr, N, M, C, drown, dt):
    """Backward of tonality / global_masking_threshold for float64 and bfloat16 tensors (psychoacoustic.py:311: the reference's
    op chain is differentiable in the dtypes the model accepts) against torch.autograd on the float64 restatement of the
    reference's formulas -- evaluated at the very (rounded) inputs the kernels saw.  float64: 1e-8 of the gradient's norm;
    bfloat16: its output rounding (2^-8) on top of the float32 kernels' own bar."""
    tdt = getattr(torch, dt)
    tol = 1e-8 if dt == "float64" else 2e-2
    B, F = 2, 3
    g = torch.Generator(device="cuda").manual_seed(5)
    env = torch.logspace(-3, 0, N, device="cuda", dtype=torch.float64).reshape(1, 1, N, 1)
    X = ((torch.rand(B, F, N, C, device="cuda", generator=g, dtype=torch.float64) * 2 - 1) * env).to(tdt).requires_grad_(True)
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M, compute_dtype=tdt)
    w = (torch.rand(B, F, N, C, device="cuda", generator=g, dtype=torch.float64) + 0.5).to(tdt)
    t = p.tonality(X)
    thr = p.global_masking_threshold(X, t, drown)
    (thr * w).sum().backward()
    assert X.grad.dtype == tdt
    Xd = X.detach().double().requires_grad_(True)
    td = _torch_tonality_reference(Xd)
    if dt == "bfloat16":
        td = td + (t.detach().double() - td).detach()      # the threshold kernel saw the ROUNDED tonality; its gradient path is td's
    thrd = _torch_psy_reference(p, Xd, td, drown)
    (thrd * w.double()).sum().backward()
    gref = Xd.grad
    assert float(torch.linalg.vector_norm(X.grad.double() - gref) / torch.linalg.vector_norm(gref)) <= tol
    # threshold alone, tonality as an independent differentiable input
    X2 = X.detach().clone().requires_grad_(True)
    t2 = t.detach().clone().requires_grad_(True)
    (p.global_masking_threshold(X2, t2, drown) * w).sum().backward()
    Xd2 = X.detach().double().requires_grad_(True)
    td2 = t.detach().double().requires_grad_(True)
    (_torch_psy_reference(p, Xd2, td2, drown) * w.double()).sum().backward()
    assert float((t2.grad.double() - td2.grad).abs().max()) <= 2 * tol * float(td2.grad.abs().max())
    assert float(torch.linalg.vector_norm(X2.grad.double() - Xd2.grad) / torch.linalg.vector_norm(Xd2.grad)) <= tol


@pytest.mark.parametrize("N,C", [(64, 2), (960, 1), (1024, 3)])
def test_float64_streaming(N, C):
    """Streaming overlap-add on float64 tensors (ac_stream_*_typed with AC_F64: a state of its own in double, the float64
    kernels): chunk by chunk -- ragged chunk lengths -- spectra, tonality, thresholds and the synthesised PCM equal the
    one-shot float64 calls bit for bit, which the float64 ORACLE confirms at 1e-12 / 1e-10 (mdctransformer.py:62-153)."""
    B, K = 2, 7
    rng = np.random.default_rng(N + C)
    x = rng.uniform(-1, 1, (B, K * N, C))
    xd = dev(x)
    codec = audiocodec_amd.AudioCodec(48000, N, compute_dtype=torch.float64)
    X, t, thr = codec.encode(xd, drown=0.1)
    xh = codec.decode(X)
    st = codec.stream(B, C)
    cuts = ((0, 1), (1, 4), (4, 7))
    parts = [st.encode_chunk(xd[:, a * N:b * N].contiguous(), drown=0.1) for a, b in cuts]
    for i, ref in enumerate((X, t, thr)):
        assert torch.equal(torch.cat([p_[i] for p_ in parts], dim=1), ref[:, :K])
    back = torch.cat([st.inverse_chunk(p_[0]) for p_ in parts], dim=1)
    assert back.dtype == torch.float64 and torch.equal(back, xh[:, :K * N])
    st.reset()
    again = torch.cat([st.transform_chunk(xd[:, a * N:b * N].contiguous()) for a, b in cuts], dim=1)
    assert torch.equal(again, X[:, :K])
    o = MDCTOracle(N, "vorbis", np.float64)
    assert rel_peak(host(X), o.transform(x)) <= 1e-12
    assert np.max(np.abs(host(back)[:, N:] - x[:, :-N])) <= 1e-12
    st.close()


def test_tensors_beyond_4_gib():
    """Offsets are 64-bit: a batch whose tensors exceed 4 GiB gives, clip for clip, the bits of a small batch."""
    N, B, K, C = 1024, 1200, 468, 2                       # x: 4.6 GB, X / thr: 4.6 GB each
    free, _ = torch.cuda.mem_get_info()
    if free < 24 * 2 ** 30:
        pytest.skip("needs 24 GiB of free HBM")
    g = torch.Generator(device="cuda").manual_seed(99)
    x = torch.rand(B, K * N, C, device="cuda", generator=g) * 2 - 1
    assert x.numel() * 4 > 2 ** 32
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(x)
    for b in (0, 599, 1199):
        Xb, tb, thrb = codec.encode(x[b:b + 1].contiguous())
        assert torch.equal(Xb, X[b:b + 1]) and torch.equal(tb, t[b:b + 1]) and torch.equal(thrb, thr[b:b + 1])
    del t, thr
    xh = codec.decode(X)
    assert float((xh[-1:, N:-N] - x[-1:]).abs().max()) <= LSB
    assert torch.equal(codec.decode(X[1199:].contiguous()), xh[1199:])


def _torch_psy_reference(p, X, t, drown):
    """The masking model in float64 torch ops (test infrastructure: lets torch.autograd produce reference gradients)."""
    W = p.W.double().cuda()
    Wi = p.W_inv.double().cuda()
    S = p.spreading_matrix.double().cuda()
    quiet = p.quiet_threshold_intensity.double().cuda()
    alpha, M = float(p.alpha), p.bark_bands_n
    eps = 1e-14
    I = X ** 2
    P = torch.einsum("nbic,ij->nbjc", I, W)
    Q = torch.clamp(P, min=eps) ** alpha
    A = torch.einsum("nbic,ij->nbjc", Q, S)
    # (the reference evaluates linspace in compute_dtype, psychoacoustic.py:187-189: float32 unless the model is float64)
    bdt = torch.float64 if p.compute_dtype == torch.float64 else torch.float32
    beta = torch.linspace(0.0, float(p.max_bark), M, dtype=bdt).double().cuda().reshape(1, 1, M, 1)
    O = (1.0 - drown) * (t * beta + 9.0 * t + 5.5)
    fac = 10.0 ** (-alpha * O / 10.0)
    T = torch.clamp(fac * A, min=eps) ** (1.0 / alpha)
    G = torch.maximum(T, quiet)
    E = torch.einsum("nbjc,jf->nbfc", G, Wi)
    return torch.sqrt(torch.clamp(E, min=eps))


def _torch_tonality_reference(X):
    eps = 1e-14
    I = X ** 2
    N = X.shape[2]
    sfm = 10.0 * (torch.log(torch.clamp(I, min=eps)).mean(dim=2, keepdim=True)
                  - torch.log(I.mean(dim=2, keepdim=True) + eps)) / np.log(10.0)
    return torch.clamp(sfm / -60.0, max=1.0)


@pytest.mark.parametrize("sr,N,M,C,drown", [(48000, 1024, 64, 2, 0.0), (44100, 256, 48, 1, 0.3), (48000, 2048, 64, 3, 0.0)])
def test_autograd_of_the_masking_model(path, sr, N, M, C, drown):
    """tonality and global_masking_threshold are differentiable: explicit adjoint kernels against torch.autograd on
    a float64 torch restatement of the same formulas."""
    B, F = 2, 3
    g = torch.Generator(device="cuda").manual_seed(3)
    env = torch.logspace(-3, 0, N, device="cuda").reshape(1, 1, N, 1)
    X = ((torch.rand(B, F, N, C, device="cuda", generator=g) * 2 - 1) * env).requires_grad_(True)
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    # forward parity of the reference restatement (guards the test itself)
    t = p.tonality(X)
    thr = p.global_masking_threshold(X, t, drown)
    Xd = X.detach().double().requires_grad_(True)
    td = _torch_tonality_reference(Xd)
    thrd = _torch_psy_reference(p, Xd, td, drown)
    assert tonality_err(t, td) <= 1.0
    assert float(((thr.detach().double() - thrd.detach()).abs() / thrd.detach()).max()) <= TOL
    w = torch.rand(B, F, N, C, device="cuda", generator=g) + 0.5
    (thr * w).sum().backward()
    (thrd * w.double()).sum().backward()
    gref = Xd.grad
    scale = float(gref.abs().max())
    assert float((X.grad.double() - gref).abs().max()) <= 2e-3 * scale
    assert float(torch.linalg.vector_norm(X.grad.double() - gref) / torch.linalg.vector_norm(gref)) <= 1e-3
    # threshold alone, tonality as an independent differentiable input
    X2 = X.detach().clone().requires_grad_(True)
    t2 = t.detach().clone().requires_grad_(True)
    (p.global_masking_threshold(X2, t2, drown) * w).sum().backward()
    Xd2 = X.detach().double().requires_grad_(True)
    td2 = t.detach().double().requires_grad_(True)
    (_torch_psy_reference(p, Xd2, td2, drown) * w.double()).sum().backward()
    assert float((t2.grad.double() - td2.grad).abs().max()) <= 2e-3 * float(td2.grad.abs().max())
    assert float(torch.linalg.vector_norm(X2.grad.double() - Xd2.grad) / torch.linalg.vector_norm(Xd2.grad)) <= 1e-3


@pytest.mark.parametrize("N,C", [(1024, 2), (1024, 1), (2048, 2), (1024, 3), (2048, 1), (2048, 3),
                                 (512, 2), (512, 1), (256, 2), (256, 1), (128, 2), (128, 1), (64, 2), (64, 1)])
def test_pcm16_at_the_boundary(N, C):
    """int16 PCM in / out: bit-identical to the float path fed pcm / 32768, and the round trip returns the PCM exactly
    (filters_n below 1024: the several-frames-per-wave kernels, block counts that leave lane groups idle)."""
    B, K = 3, (5 if N >= 1024 else 21)
    pcm = torch.randint(-32768, 32768, (B, K * N, C), device="cuda", dtype=torch.int16)
    pcm[0, :7, 0] = torch.tensor([-32768, 32767, 0, 1, -1, 12345, -12345], dtype=torch.int16)
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(pcm)
    Xf, tf, thrf = codec.encode(pcm.float() / 32768.0)
    assert X.dtype == torch.float32
    out = codec.decode(X, pcm16=True)
    assert out.dtype == torch.int16 and tuple(out.shape) == (B, (K + 2) * N, C)
    assert torch.equal(out[:, N:-N], pcm)
    ref = torch.clamp(torch.round(codec.decode(X) * 32768.0), -32768, 32767).to(torch.int16)
    if C > 2:
        # three channels: 16-bit PCM runs the wave-level kernels' strided form, float32 the channel-pair instances of the
        # LDS-FFT tier and the general-layout masking kernels -- two routes, equal to float32 rounding
        assert float((X - Xf).abs().max()) <= 2e-6 * float(Xf.abs().max())
        assert float((t - tf).abs().max()) <= 1e-5 and float(((thr - thrf).abs() / thrf).max()) <= 1e-4
        assert int((out.int() - ref.int()).abs().max()) <= 1
        return
    assert torch.equal(X, Xf)
    if N == 2048 and C != 2:
        # here the two inputs take different routes (PCM: transform + masking-model launch; float: one fused
        # launch), whose compilers need not contract the same multiply-adds
        assert float((t - tf).abs().max()) <= 1e-6 and float(((thr - thrf).abs() / thrf).max()) <= 1e-5
    else:
        assert torch.equal(t, tf) and torch.equal(thr, thrf)
    assert torch.equal(out, ref)
    with pytest.raises(_lib.AudioCodecError):      # the LDS-FFT tier takes 16-bit PCM at seven sizes, mono / stereo
        audiocodec_amd.AudioCodec(48000, 800).encode(torch.zeros(1, 4 * 800, 2, device="cuda", dtype=torch.int16))
    with pytest.raises(_lib.AudioCodecError):
        audiocodec_amd.AudioCodec(48000, 960).encode(torch.zeros(1, 4 * 960, 3, device="cuda", dtype=torch.int16))
    with pytest.raises(_lib.AudioCodecError):      # ... nor do the short-frame kernels for other channel counts
        audiocodec_amd.AudioCodec(48000, 256).encode(torch.zeros(1, 4 * 256, 3, device="cuda", dtype=torch.int16))


@pytest.mark.parametrize("N", [120, 240, 480, 960, 1920, 576, 1152])
@pytest.mark.parametrize("C", [2, 1])
def test_pcm16_at_the_boundary_of_the_lds_fft_tier(N, C):
    """int16 PCM in / out at the frame lengths of the speech and music codecs (Opus 2.5 ... 20 ms, 1920, MP3 576 / 1152):
    bit-identical spectra to the float path fed pcm / 32768, the round trip returns the PCM exactly, an odd mono batch."""
    B, K = 3, 37
    pcm = torch.randint(-32768, 32768, (B, K * N, C), device="cuda", dtype=torch.int16)
    pcm[0, :7, 0] = torch.tensor([-32768, 32767, 0, 1, -1, 12345, -12345], dtype=torch.int16)
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(pcm)
    Xf, tf, thrf = codec.encode(pcm.float() / 32768.0)
    assert X.dtype == torch.float32 and torch.equal(X, Xf) and torch.equal(t, tf) and torch.equal(thr, thrf)
    out = codec.decode(X, pcm16=True)
    assert out.dtype == torch.int16 and tuple(out.shape) == (B, (K + 2) * N, C)
    assert torch.equal(out[:, N:-N], pcm)
    ref = torch.clamp(torch.round(codec.decode(X) * 32768.0), -32768, 32767).to(torch.int16)
    assert torch.equal(out, ref)


def test_plan_and_stream_lifecycle():
    """Plans and streams own device memory: creating and dropping them repeatedly returns it (no leak), and a model
    shared by two host threads on two HIP streams gives the single-threaded bits."""
    import gc
    import threading
    torch.cuda.synchronize()
    x = torch.rand(2, 4 * 1024, 2, device="cuda") * 2 - 1
    ref = audiocodec_amd.AudioCodec(48000, 1024).encode(x)
    gc.collect()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(50):
        c = audiocodec_amd.AudioCodec(48000, 1024)
        X, t, thr = c.encode(x)
        st = audiocodec_amd.StreamingMDCT(c.mdct, 2, 2)
        st.transform_chunk(x[:, :1024])
        st.close()
        del c, st
    gc.collect()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 8 * 2 ** 20, "device memory not returned: %d bytes" % (free0 - free1)
    assert torch.equal(X, ref[0]) and torch.equal(thr, ref[2])
    codec = audiocodec_amd.AudioCodec(48000, 1024)
    outs = [None, None]

    def work(i):
        s = torch.cuda.Stream()
        with torch.cuda.stream(s):
            for _ in range(5):
                outs[i] = codec.encode(x)
        s.synchronize()

    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    [t_.start() for t_ in th]
    [t_.join() for t_ in th]
    for o in outs:
        assert torch.equal(o[0], ref[0]) and torch.equal(o[1], ref[1]) and torch.equal(o[2], ref[2])


# ---- compute_dtype variants ------------------------------------------------------------

def _bf16_round(a):
    """float64 array -> the values a torch.bfloat16 tensor holds (round to nearest even)"""
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


@pytest.mark.parametrize("N,wt,C", [(1024, "vorbis", 2), (64, "sine", 1), (12, "vorbis", 3), (16, "rect", 2), (2048, "vorbis", 1)])
def test_float64_filter_bank_vs_oracle(N, wt, C):
    """compute_dtype = float64: float64 tensors, arithmetic and constants -- agrees with the oracle to rounding"""
    rng = np.random.default_rng(N + C)
    x = rng.uniform(-1, 1, (2, 4 * N, C))
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, compute_dtype=torch.float64)
    o = MDCTOracle(N, wt, np.float64)
    X = m.transform(dev(x))
    assert X.dtype == torch.float64
    Xo = o.transform(x)
    assert rel_peak(host(X), Xo) <= 1e-12
    xh = host(m.inverse_transform(X))
    assert np.max(np.abs(xh - o.inverse_transform(Xo))) <= 1e-12
    if wt != "rect":
        assert np.max(np.abs(xh[:, N:-N] - x)) <= 1e-12
    with pytest.raises(ValueError):
        m.transform(dev(x.astype(np.float32)))              # no implicit cast, as in the reference
    assert m.transform(dev(x).requires_grad_()).requires_grad   # (differentiable in every dtype: test_autograd_of_the_filter_bank_in_every_dtype)


@pytest.mark.parametrize("sr,N,M,C", [(48000, 1024, 64, 2), (44100, 256, 48, 1), (48000, 2048, 64, 3)])
def test_float64_masking_model_vs_oracle(sr, N, M, C):
    rng = np.random.default_rng(N + M)
    env = np.logspace(-5, 0, N).reshape(1, 1, N, 1)
    X = rng.uniform(-1, 1, (2, 3, N, C)) * env * rng.uniform(1e-3, 1, (2, 3, 1, C))
    X[0, 0, :, 0] = 0.0
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M, compute_dtype=torch.float64)
    o = PsychoOracle(sr, N, M, compute_dtype=np.float64)
    np.testing.assert_allclose(p.W.numpy(), o.W, rtol=1e-11, atol=1e-14)   # libm vs numpy in the Bark edges
    t = p.tonality(dev(X))
    to = o.tonality(X)
    assert t.dtype == torch.float64 and np.max(np.abs(host(t) - to)) <= 1e-12
    for drown in (0.0, 0.4):
        thr = host(p.global_masking_threshold(dev(X), t, drown))
        assert rel_elem(thr, o.global_masking_threshold(X, to, drown)) <= 1e-10
    a = dev(rng.uniform(-1, 1, 1000))
    np.testing.assert_allclose(host(p.amplitude_to_dB(a)), np.maximum(10 * np.log10(np.maximum(1e-14, host(a) ** 2)) + 120, -20), atol=1e-10)
    assert float(p.amplitude_to_dB_norm(a).min()) >= 0.0
    thr_t = dev(np.full(X.shape, 0.3))
    y64 = p.add_noise(dev(X), thr_t, seed=9)
    p32 = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    y32 = p32.add_noise(dev(X.astype(np.float32)), thr_t.float(), seed=9)
    assert float((y64 - y32.double()).abs().max()) <= 1e-6     # the same normals for the same seed


@pytest.mark.parametrize("N,K", [(1024, 468), (2048, 234)])
def test_float32_kernels_vs_float64_kernels_at_full_size(N, K):
    """BASELINE configs[1] (N = 1024) and configs[3] (N = 2048, spreading product on the matrix cores) at their full size,
    element by element on the device: the wave-level float32 kernels against the float64 kernels on the same input (the
    float64 kernels are themselves held to the oracle at 1e-12 above)."""
    B, C = 256, 2
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.rand(B, K * N, C, device="cuda", generator=g) * 2 - 1
    c32 = audiocodec_amd.AudioCodec(48000, N)
    assert c32.psy.plan_spreading() == "bf16x2_mfma"
    X, t, thr = c32.encode(x)
    xh = c32.decode(X)
    assert float((xh[:, N:-N] - x).abs().max()) <= LSB
    del xh
    m64 = audiocodec_amd.MDCTransformer(N, compute_dtype=torch.float64)
    p64 = audiocodec_amd.PsychoacousticModel(48000, N, compute_dtype=torch.float64)
    step = 32                                                # clips per float64 pass (memory: 2 x 8 bytes per sample)
    worst_X = worst_l2 = worst_t = worst_thr = 0.0
    for b in range(0, B, step):
        X64 = m64.transform(x[b:b + step].double())
        d = (X[b:b + step].double() - X64)
        peak = X64.abs().amax(dim=2, keepdim=True).clamp_min(1e-30)
        worst_X = max(worst_X, float((d.abs().amax(dim=2, keepdim=True) / peak).max()))
        worst_l2 = max(worst_l2, float(d.norm() / X64.norm()))
        t64 = p64.tonality(X64)
        worst_t = max(worst_t, tonality_err(t[b:b + step], t64))
        thr64 = p64.global_masking_threshold(X64, t64)
        worst_thr = max(worst_thr, float(((thr[b:b + step].double() - thr64).abs() / thr64).max()))
        del X64, d, t64, thr64
    assert worst_X <= TOL and worst_l2 <= TOL, (worst_X, worst_l2)
    assert worst_t <= 1.0, worst_t                            # in units of the tonality bar (1e-4 |t| + 1e-6)
    assert worst_thr <= TOL, worst_thr                        # threshold of the float32 X and t (two rounding sources)


@pytest.mark.parametrize("N,wt,C", [(1024, "vorbis", 2), (256, "sine", 1), (12, "vorbis", 3), (2048, "vorbis", 2), (64, "rect", 3)])
def test_bfloat16_filter_bank(path, N, wt, C):
    """compute_dtype = bfloat16: bfloat16 tensors, float32 arithmetic.  Tolerance: the output rounding of bfloat16
    (2^-9 of each value) on top of the float32 kernels' own error -- 4e-3 of the frame's peak."""
    rng = np.random.default_rng(N)
    x = _bf16_round(rng.uniform(-1, 1, (2, 5 * N, C)))
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, compute_dtype=torch.bfloat16)
    o = MDCTOracle(N, wt, np.float64)
    X = m.transform(dev(x).to(torch.bfloat16))
    assert X.dtype == torch.bfloat16
    Xo = o.transform(x)
    assert rel_peak(host(X.double()), Xo) <= 4e-3
    xh = host(m.inverse_transform(X).double())
    ref = o.inverse_transform(host(X.double()))              # the oracle on the very coefficients the kernel read
    assert np.max(np.abs(xh - ref)) <= 4e-3 * max(1.0, np.max(np.abs(ref)))
    if wt != "rect":
        assert np.max(np.abs(xh[:, N:-N] - x)) <= 2e-2       # round trip through bfloat16 coefficients


@pytest.mark.parametrize("N,wt,C", [(1024, "vorbis", 2), (960, "vorbis", 2), (256, "sine", 1), (12, "vorbis", 3), (4096, "vorbis", 2), (64, "rect", 3)])
def test_float16_filter_bank(path, N, wt, C):
    """compute_dtype = float16 (the reference's filter bank accepts it and up-casts inside its DCT-IV, mdctransformer.py:
    327-344): float16 tensors, float32 arithmetic.  Tolerance: the output rounding of float16 (2^-11 of each value) on top of
    the float32 kernels' own error -- 1e-3 of the frame's peak; the masking model refuses the type (psychoacoustic.py:42-43)."""
    rng = np.random.default_rng(N)
    x = torch.from_numpy(rng.uniform(-1, 1, (2, 5 * N, C))).to(torch.float16)
    x64 = x.double().numpy()
    m = audiocodec_amd.MDCTransformer(N, window_type=wt, compute_dtype=torch.float16)
    o = MDCTOracle(N, wt, np.float64)
    X = m.transform(x.cuda())
    assert X.dtype == torch.float16
    assert rel_peak(host(X.double()), o.transform(x64)) <= 1e-3
    xh = host(m.inverse_transform(X).double())
    ref = o.inverse_transform(host(X.double()))              # the oracle on the very coefficients the kernel read
    assert np.max(np.abs(xh - ref)) <= 1e-3 * max(1.0, np.max(np.abs(ref)))
    if wt != "rect":
        assert np.max(np.abs(xh[:, N:-N] - x64)) <= 5e-3     # round trip through float16 coefficients
    with pytest.raises(TypeError):
        audiocodec_amd.PsychoacousticModel(48000, N, compute_dtype=torch.float16)
    with pytest.raises(ValueError):
        m.transform(x.cuda().float())                        # inputs must carry compute_dtype (mdctransformer.py:22-23)


@pytest.mark.parametrize("N", [1024, 2048, 512, 64, 960, 1920, 100])
def test_non_finite_and_denormal_inputs(path, N):
    """NaN / infinite / denormal samples against the oracle, whose np.maximum / np.minimum propagate NaN as tf.maximum /
    tf.minimum do (psychoacoustic.py:113-116, 205-208, 331): the two frames that contain a NaN or infinite sample have NaN
    coefficients, a NaN tonality and an all-NaN threshold row FOR THAT SIGNAL; every other frame, and the other signal, meet
    the oracle at the usual bar -- through the fused encode of every tier and through the three separate calls (which, given
    a finite tonality for a poisoned spectrum, still return the NaN row).  Denormal samples are ordinary numbers."""
    import warnings
    C, K = 2, 8
    rng = np.random.default_rng(N)
    x = rng.uniform(-1, 1, (3, K * N, C)).astype(np.float32)
    x[0, 2 * N + 5, 0] = np.nan                  # clip 0, signal 0: frames 2 and 3
    x[1, 5 * N + N // 2, 1] = np.inf             # clip 1, signal 1: frames 5 and 6
    x[2, 3 * N:4 * N, :] = 1e-40                 # clip 2: a block of denormals
    codec = audiocodec_amd.AudioCodec(48000, N)
    X, t, thr = codec.encode(dev(x))
    o_m, o_p = MDCTOracle(N, "vorbis", np.float64), PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        Xo = o_m.transform(x.astype(np.float64))
        to = o_p.tonality(Xo)
        thro = o_p.global_masking_threshold(Xo, to)
    Xh, th, thrh = host(X), host(t), host(thr)
    # the same coefficients are not finite (an infinite sample: infinities and NaNs, which of the two an FFT leaves where is the
    # algorithm's business -- the reference's dense products make them all NaN); the same tonalities and thresholds are NaN
    assert np.array_equal(np.isfinite(Xh), np.isfinite(Xo)) and np.array_equal(np.isnan(th), np.isnan(to))
    assert np.array_equal(np.isnan(thrh), np.isnan(thro))
    assert np.isnan(Xh[0, 2:4, :, 0]).all() and np.isfinite(Xh[0, 2:4, :, 1]).all() and np.isnan(th[0, 2:4, 0, 0]).all()
    assert np.isnan(thrh[0, 2:4, :, 0]).all() and np.isnan(thrh[1, 5:7, :, 1]).all() and np.isfinite(thrh[2]).all()
    ok = ~np.isnan(thro)
    okX = np.isfinite(Xo)
    # ... and the rest meets the oracle (coefficients: per-frame peak metric over the finite frames)
    Xc, Xoc = np.where(okX, Xh, 0.0), np.where(okX, Xo, 0.0)
    assert rel_peak(Xc, Xoc) <= TOL
    X64 = np.where(okX, Xh.astype(np.float64), np.nan)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t64 = o_p.tonality(X64)
        thr64 = o_p.global_masking_threshold(X64, t64)
    okt = ~np.isnan(t64)
    assert tonality_err(th[okt], t64[okt]) <= 1.0
    assert float(np.max(np.abs(thrh[ok] - thr64[ok]) / thr64[ok])) <= TOL
    # the separate calls: same values; a finite tonality handed in for a poisoned spectrum does not un-poison the row (a NaN
    # sample: NaN, as the reference; an infinite one: not finite -- infinities where the coefficients are all infinite and the
    # reference has them, NaN where the matrix-core product met them)
    t2 = codec.psy.tonality(X)
    assert torch.equal(torch.nan_to_num(t2, nan=-7.0), torch.nan_to_num(t, nan=-7.0))
    thr2 = host(codec.psy.global_masking_threshold(X, torch.nan_to_num(t, nan=0.5)))
    assert np.array_equal(np.isnan(thr2[0]), np.isnan(thro[0])) and np.array_equal(np.isfinite(thr2[1]), np.isfinite(thro[1]))
    fin = ~np.isnan(thro[0])
    assert rel_elem(thr2[2], thrh[2]) <= 1e-5 and rel_elem(thr2[0][fin], thrh[0][fin]) <= 1e-5   # (bit-equal below 1024; one rounding apart there)


@pytest.mark.parametrize("N,C", [(1024, 2), (1024, 1), (2048, 2), (2048, 1)])
def test_bfloat16_streaming(N, C):
    """Streaming overlap-add on bfloat16 tensors (ac_stream_*_typed: the wave-level kernels, state kept in float32): chunk
    by chunk -- ragged chunk lengths -- the spectra, tonality, thresholds and the synthesised PCM equal the one-shot
    bfloat16 calls bit for bit; against the float64 ORACLE on the bfloat16-rounded input they stay within bfloat16's
    rounding (coefficients 4e-3 of the frame peak, round trip 2e-2)."""
    _lib.load().ac_set_force_generic(0)
    B, K = 3, 9
    rng = np.random.default_rng(N + C)
    x = _bf16_round(rng.uniform(-1, 1, (B, K * N, C)))
    xd = dev(x).to(torch.bfloat16)
    codec = audiocodec_amd.AudioCodec(48000, N, compute_dtype=torch.bfloat16)
    X, t, thr = codec.encode(xd, drown=0.2)
    xh = codec.decode(X)
    st = codec.stream(B, C)
    cuts = ((0, 2), (2, 3), (3, 9))
    parts = [st.encode_chunk(xd[:, a * N:b * N].contiguous(), drown=0.2) for a, b in cuts]
    for i, ref in enumerate((X, t, thr)):
        assert torch.equal(torch.cat([p_[i] for p_ in parts], dim=1), ref[:, :K])
    back = torch.cat([st.inverse_chunk(p_[0]) for p_ in parts], dim=1)
    assert back.dtype == torch.bfloat16 and torch.equal(back, xh[:, :K * N])
    st.reset()
    again = torch.cat([st.transform_chunk(xd[:, a * N:b * N].contiguous()) for a, b in cuts], dim=1)
    assert torch.equal(again, codec.mdct.transform(xd)[:, :K])
    o = MDCTOracle(N, "vorbis", np.float64)
    assert rel_peak(host(X.double()), o.transform(x)) <= 4e-3
    assert np.max(np.abs(host(back.double())[:, N:] - x[:, :-N])) <= 2e-2
    st.close()
    with pytest.raises(_lib.AudioCodecError):
        audiocodec_amd.StreamingMDCT(audiocodec_amd.MDCTransformer(512, compute_dtype=torch.bfloat16), 1, 2).transform_chunk(
            torch.zeros(1, 512, 2, device="cuda", dtype=torch.bfloat16))


@pytest.mark.parametrize("sr,N,M,C", [(48000, 1024, 64, 2), (44100, 256, 48, 3), (48000, 2048, 64, 1)])
def test_bfloat16_masking_model(path, sr, N, M, C):
    rng = np.random.default_rng(M + C)
    env = np.logspace(-4, 0, N).reshape(1, 1, N, 1)
    X = _bf16_round(rng.uniform(-1, 1, (2, 3, N, C)) * env * rng.uniform(1e-2, 1, (2, 3, 1, C)))
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M, compute_dtype=torch.bfloat16)
    o = PsychoOracle(sr, N, M, compute_dtype=np.float64)
    Xd = dev(X).to(torch.bfloat16)
    t = p.tonality(Xd)
    assert t.dtype == torch.bfloat16
    assert np.max(np.abs(host(t.double()) - o.tonality(X))) <= 4e-3     # 2^-9 of a value in [0, 1]
    tb = host(t.double())
    thr = host(p.global_masking_threshold(Xd, t, 0.1).double())
    assert rel_elem(thr, o.global_masking_threshold(X, tb, 0.1)) <= 6e-3
    dB = p.amplitude_to_dB(Xd)
    ref = np.maximum(10 * np.log10(np.maximum(1e-14, X ** 2)) + 120, -20)
    assert dB.dtype == torch.bfloat16 and np.max(np.abs(host(dB.double()) - ref)) <= 0.5   # 2^-9 of values up to 120
    y = p.add_noise(Xd, torch.full_like(Xd, 0.25), seed=3)
    assert y.dtype == torch.bfloat16 and abs(float((y.double() - Xd.double()).std()) - 0.25 / 6) < 2e-3


@pytest.mark.parametrize("N,C", [(1024, 2), (1024, 1), (2048, 2), (2048, 1), (1024, 3)])
def test_bfloat16_fused_encode(path, N, C):
    """bfloat16 codec on the wave-level kernels (stereo / mono; 3 channels fall to the LDS-FFT tier): the fused encode
    equals the three calls bit for bit (X and tonality are rounded to bfloat16 before the masking model uses them) and
    stays within bfloat16's rounding of the oracle"""
    rng = np.random.default_rng(N + C)
    B, K = 3, 6
    x = _bf16_round(rng.uniform(-1, 1, (B, K * N, C)))
    xd = dev(x).to(torch.bfloat16)
    codec = audiocodec_amd.AudioCodec(48000, N, compute_dtype=torch.bfloat16)
    X, t, thr = codec.encode(xd, drown=0.1)
    Xu = codec.mdct.transform(xd)
    tu = codec.psy.tonality(Xu)
    thru = codec.psy.global_masking_threshold(Xu, tu, 0.1)
    assert torch.equal(X, Xu) and torch.equal(t, tu) and torch.equal(thr, thru)
    om, op = MDCTOracle(N, "vorbis", np.float64), PsychoOracle(48000, N, 64, compute_dtype=np.float64)
    assert rel_peak(host(X.double()), om.transform(x)) <= 4e-3
    Xb, tb = host(X.double()), host(t.double())
    assert np.max(np.abs(tb - op.tonality(Xb))) <= 4e-3
    assert rel_elem(host(thr.double()), op.global_masking_threshold(Xb, tb, 0.1)) <= 6e-3
    xh = codec.decode(X)
    assert xh.dtype == torch.bfloat16 and float((xh[:, N:-N].double() - dev(x)).abs().max()) <= 2e-2


@pytest.mark.parametrize("dtype", [torch.float64, torch.bfloat16])
def test_codec_in_other_compute_dtypes(dtype):
    N = 256
    x = (torch.rand(2, 6 * N, 2, device="cuda") * 2 - 1).to(dtype)
    codec = audiocodec_amd.AudioCodec(48000, N, compute_dtype=dtype)
    X, t, thr = codec.encode(x, drown=0.3)
    assert X.dtype == t.dtype == thr.dtype == dtype
    Xu = codec.mdct.transform(x)
    assert torch.equal(X, Xu) and torch.equal(t, codec.psy.tonality(Xu))
    assert torch.equal(thr, codec.psy.global_masking_threshold(Xu, t, 0.3))
    xh = codec.decode(X)
    assert xh.dtype == dtype
    assert float((xh[:, N:-N] - x).double().abs().max()) <= (1e-12 if dtype == torch.float64 else 2e-2)
    if dtype == torch.float64:                      # (float64 streams: every size, test_float64_streaming)
        assert audiocodec_amd.StreamingMDCT(codec.mdct, 2, 2).transform_chunk(x[:, :N].contiguous()).dtype == torch.float64
    else:                                           # (bfloat16 streams exist where the wave-level kernels serve them: N = 1024 / 2048)
        with pytest.raises(_lib.AudioCodecError):
            audiocodec_amd.StreamingMDCT(codec.mdct, 2, 2).transform_chunk(x[:, :N].contiguous())
    assert codec.psy.tonality(X.clone().requires_grad_()).requires_grad   # (test_autograd_of_the_masking_model_in_other_dtypes)


def test_fuzz_wave_kernels_against_the_generic_kernels():
    """seeded random shapes (ragged batch / block / channel counts, both wave-level sizes, both Princen-Bradley windows):
    every entry point on the wave-level kernels against the O(N^2) kernels on the same device tensors"""
    rng = np.random.default_rng(2024)
    lib = _lib.load()
    for case in range(36):
        N = int(rng.choice([1024, 2048]))
        wt = str(rng.choice(["vorbis", "sine"]))
        B, K, C = int(rng.integers(1, 6)), int(rng.integers(0, 8)), int(rng.integers(1, 6))
        drown = float(rng.choice([0.0, 0.25, 1.0]))
        x = torch.from_numpy(rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)).cuda()
        if K > 1:
            x[0, N:2 * N] *= 1e-4                               # a quiet block
        codec = audiocodec_amd.AudioCodec(48000, N, window_type=wt)
        assert codec.mdct.is_fast() and codec.psy.is_fast()
        lib.ac_set_force_generic(0)
        X, t, thr = codec.encode(x, drown=drown)
        Xs = codec.mdct.transform(x)
        ts = codec.psy.tonality(X)
        thrs = codec.psy.global_masking_threshold(X, t, drown)
        xh = codec.decode(X)
        lib.ac_set_force_generic(1)
        try:
            Xg, tg, thrg = codec.encode(x, drown=drown)
            xg = codec.decode(X)
        finally:
            lib.ac_set_force_generic(0)
        tag = "case %d: N=%d %s B=%d K=%d C=%d" % (case, N, wt, B, K, C)
        assert tuple(X.shape) == (B, K + 1, N, C) and tuple(xh.shape) == (B, (K + 2) * N, C), tag
        peak = Xg.abs().amax(dim=2, keepdim=True).clamp_min(1e-20)
        assert float(((X - Xg).abs() / peak).max()) <= TOL, tag
        assert float(((Xs - Xg).abs() / peak).max()) <= TOL, tag
        assert tonality_err(t, tg) <= 1.0 and tonality_err(ts, tg) <= 1.0, tag
        assert float(((thr - thrg).abs() / thrg).max()) <= TOL and float(((thrs - thrg).abs() / thrg).max()) <= TOL, tag
        assert float((xh - xg).abs().max()) <= 2e-6, tag
        if K > 0:
            assert float((xh[:, N:-N] - x).abs().max()) <= LSB, tag


def test_fuzz_other_sizes_against_the_generic_kernels():
    """the same for the other tiers: several frames per wave (filters_n 64 ... 512), the mixed-radix LDS-FFT tier (960,
    480, 240, 120, 576, 192, 96), the masking model for general band layouts beside them; ragged shapes, block counts
    that leave lane groups idle, K = 0"""
    rng = np.random.default_rng(4096)
    lib = _lib.load()
    for case in range(48):
        N = int(rng.choice([64, 128, 256, 512, 960, 480, 240, 120, 576, 192, 96]))
        wt = str(rng.choice(["vorbis", "sine"]))
        B, K, C = int(rng.integers(1, 6)), int(rng.integers(0, 40)), int(rng.integers(1, 4))
        M = int(rng.choice([64, 32, 20])) if N >= 128 else int(rng.choice([16, 8]))
        drown = float(rng.choice([0.0, 0.25, 1.0]))
        x = torch.from_numpy(rng.uniform(-1, 1, (B, K * N, C)).astype(np.float32)).cuda()
        if K > 1:
            x[0, N:2 * N] *= 1e-4
        codec = audiocodec_amd.AudioCodec(48000, N, bark_bands_n=M, window_type=wt)
        lib.ac_set_force_generic(0)
        X, t, thr = codec.encode(x, drown=drown)
        Xs = codec.mdct.transform(x)
        thrs = codec.psy.global_masking_threshold(X, t, drown)
        xh = codec.decode(X)
        lib.ac_set_force_generic(1)
        try:
            Xg = codec.mdct.transform(x)
            tg = codec.psy.tonality(X)                         # the masking model on the same spectrum: at these frame
            thrg = codec.psy.global_masking_threshold(X, t, drown)   # sizes a rounding of X moves the tonality measurably
            xg = codec.decode(X)
        finally:
            lib.ac_set_force_generic(0)
        tag = "case %d: N=%d %s B=%d K=%d C=%d M=%d tier %d" % (case, N, wt, B, K, C, M, codec.psy.tier())
        assert tuple(X.shape) == (B, K + 1, N, C) and tuple(xh.shape) == (B, (K + 2) * N, C), tag
        peak = Xg.abs().amax(dim=2, keepdim=True).clamp_min(1e-20)
        assert float(((X - Xg).abs() / peak).max()) <= TOL and torch.equal(X, Xs), tag
        assert tonality_err(t, tg) <= 1.0, tag
        assert float(((thr - thrg).abs() / thrg).max()) <= TOL and float(((thrs - thrg).abs() / thrg).max()) <= TOL, tag
        assert float((xh - xg).abs().max()) <= 2e-6, tag
        if K > 0:
            assert float((xh[:, N:-N] - x).abs().max()) <= LSB, tag


@pytest.mark.parametrize("dtype", [torch.float32, torch.float64, torch.bfloat16])
@pytest.mark.parametrize("N", [1024, 2048, 64])
def test_empty_and_ragged_shapes_in_every_dtype(path, dtype, N):
    """no blocks, no clips, a single block, odd signal counts -- in each compute dtype and on each tier"""
    codec = audiocodec_amd.AudioCodec(48000, N, compute_dtype=dtype)
    z = lambda *s: torch.zeros(*s, device="cuda", dtype=dtype)   # noqa: E731
    X, t, thr = codec.encode(z(2, 0, 2))                          # K = 0: one frame of silence
    assert tuple(X.shape) == (2, 1, N, 2) and float(X.abs().max()) == 0.0 and float(thr.min()) > 0.0
    assert tuple(codec.decode(X).shape) == (2, 2 * N, 2)
    X, t, thr = codec.encode(z(0, 2 * N, 1))                      # B = 0
    assert tuple(X.shape) == (0, 3, N, 1) and tuple(t.shape) == (0, 3, 1, 1)
    assert tuple(codec.decode(z(1, 0, N, 1)).shape) == (1, N, 1)  # no frames at all
    x = (torch.rand(3, N, 1, device="cuda") * 2 - 1).to(dtype)    # three mono clips of one block: an odd pair count
    X, t, thr = codec.encode(x)
    xh = codec.decode(X)
    tol = {torch.float32: LSB, torch.float64: 1e-12, torch.bfloat16: 2e-2}[dtype]
    assert float((xh[:, N:-N] - x).double().abs().max()) <= tol
    single = codec.encode(x[1:2])                                 # the middle clip alone gives the same numbers
    assert torch.equal(single[0], X[1:2]) and torch.equal(single[1], t[1:2]) and torch.equal(single[2], thr[1:2])


def test_integration_md_stub_runs_against_the_library():
    """the ctypes stub INTEGRATION.md shows to a maintainer of the reference, executed as written (only the library path is
    substituted), gives the class API's results through raw device pointers"""
    import os
    import re
    from conftest import ROOT
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(# audiocodec/_amd.py.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libaudiocodec_amd.so")', "ctypes.CDLL(%r)" % _lib.LIB_PATH)
    stub = {}
    exec(compile(code, "INTEGRATION.md", "exec"), stub)
    N, B, K, C = 1024, 2, 3, 2
    x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
    X = torch.empty(B, K + 1, N, C, device="cuda")
    t = torch.empty(B, K + 1, 1, C, device="cuda")
    thr = torch.empty_like(X)
    xh = torch.empty(B, (K + 2) * N, C, device="cuda")
    mp, pp = stub["mdct_plan"](N, "vorbis"), stub["psy_plan"](N, 64, 48000.0, 0.6)
    stream = torch.cuda.current_stream().cuda_stream
    stub["transform"](mp, x.data_ptr(), X.data_ptr(), B, K, C, stream)
    stub["tonality"](pp, X.data_ptr(), t.data_ptr(), B, K + 1, C, stream)
    stub["global_masking_threshold"](pp, X.data_ptr(), t.data_ptr(), 0.25, thr.data_ptr(), B, K + 1, C, stream)
    stub["inverse_transform"](mp, X.data_ptr(), xh.data_ptr(), B, K + 1, C, stream)
    torch.cuda.synchronize()
    codec = audiocodec_amd.AudioCodec(48000, N)
    assert torch.equal(X, codec.mdct.transform(x)) and torch.equal(t, codec.psy.tonality(X))
    assert torch.equal(thr, codec.psy.global_masking_threshold(X, t, 0.25))
    assert float((xh[:, N:-N] - x).abs().max()) <= LSB
    with pytest.raises(ValueError):
        stub["mdct_plan"](7, "vorbis")
    lib = stub["_lib"]
    lib.ac_mdct_plan_destroy(mp)
    lib.ac_psy_plan_destroy(pp)


def test_workspace_places_buffers_without_changing_results():
    """audiocodec_amd.Workspace only decides where the caller's tensors live: shapes, results and the round trip are those of
    plain allocations, with and without the placement probing"""
    N, B, K, C = 1024, 4, 6, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
    ref = codec.encode(x)
    for tune in (True, False):
        ws = audiocodec_amd.Workspace(codec, B, K, C, span_gib=2.0, max_tries=3, tune=tune)
        assert tuple(ws.x.shape) == (B, K * N, C) and tuple(ws.X.shape) == (B, K + 1, N, C)
        assert tuple(ws.t.shape) == (B, K + 1, 1, C) and tuple(ws.thr.shape) == tuple(ws.X.shape)
        assert tuple(ws.xh.shape) == (B, (K + 2) * N, C)
        assert all(v.is_contiguous() and v.data_ptr() % (1 << 21) == 0 for v in (ws.x, ws.X, ws.thr, ws.xh))
        assert ws.report["tuned"] == tune and 1 <= ws.report["tries"] <= (3 if tune else 1)
        ws.x.copy_(x)
        codec.encode_into(ws.x, ws.X, ws.t, ws.thr)
        codec.decode_into(ws.X, ws.xh)
        assert torch.equal(ws.X, ref[0]) and torch.equal(ws.t, ref[1]) and torch.equal(ws.thr, ref[2])
        assert float((ws.xh[:, N:-N] - x).abs().max()) <= LSB
    with pytest.raises(NotImplementedError):
        audiocodec_amd.Workspace(audiocodec_amd.AudioCodec(48000, N, compute_dtype=torch.float64), B, K, C)


def test_library_allocated_results_are_placed_without_changing_them():
    """AudioCodec.encode / decode allocate what they return (the reference's API shape), so the LIBRARY decides where it
    lives: the first encode too large for the Infinity Cache builds the device's pool (ac_workspace_create: <= 8 timed
    candidates, spacers and losers back with the driver), later results are carved out of its two regions
    (ac_workspace_alloc_dlpack) and give their extent back when they die.  Same values as caller-owned plain tensors, bit
    for bit; spectra and thresholds in different regions; extents recycled; fallback to torch.empty when the regions are
    full; AC_NO_PLACEMENT semantics via placement.release."""
    from audiocodec_amd import placement
    placement.release()
    N, B, K, C = 1024, 80, 234, 2                       # X = 154 MB: X and thr together exceed the 256 MiB Infinity Cache
    codec = audiocodec_amd.AudioCodec(48000, N)
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    Xr, tr, thrr = (torch.empty(B, K + 1, N, C, device="cuda"), torch.empty(B, K + 1, 1, C, device="cuda"),
                    torch.empty(B, K + 1, N, C, device="cuda"))
    codec.encode_into(x, Xr, tr, thrr)
    Xt_ref = codec.mdct.transform(x[:8])                 # (no pool yet: plain allocations)
    thrt_ref = codec.psy.global_masking_threshold(Xt_ref, codec.psy.tonality(Xt_ref))
    assert placement.report() is None
    X, t, thr = codec.encode(x)
    rep = codec.placement_report()
    assert rep is not None and 1 <= rep["tries"] <= 8 and rep["sized_for"]["batches_n"] == B and rep["live_tensors"] == 2
    assert rep["bytes_held"] == rep["region_spectra_bytes"] + rep["region_other_bytes"] <= 16 * 2 ** 30
    assert torch.equal(X, Xr) and torch.equal(t, tr) and torch.equal(thr, thrr)
    import ctypes
    a, b, na, nb = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_size_t(), ctypes.c_size_t()
    _lib.load().ac_workspace_regions(placement.pool(x.device).handle, ctypes.byref(a), ctypes.byref(na), ctypes.byref(b), ctypes.byref(nb))
    assert a.value <= X.data_ptr() < a.value + na.value and b.value <= thr.data_ptr() < b.value + nb.value
    xh = codec.decode(X)
    assert b.value <= xh.data_ptr() < b.value + nb.value and float((xh[:, N:-N] - x).abs().max()) <= LSB
    # views keep their extent alive; the extent comes back when the last one dies
    view = thr[:, 1:3]
    p_thr = thr.data_ptr()
    del thr
    assert codec.placement_report()["live_tensors"] == 3
    keep = view.clone()
    del view
    assert codec.placement_report()["live_tensors"] == 2
    X2, t2, thr2 = codec.encode(x)                       # second generation while the first is alive
    assert torch.equal(X2, Xr) and torch.equal(thr2, thrr) and torch.equal(thr2[:, 1:3], keep)
    assert X2.data_ptr() != X.data_ptr() and thr2.data_ptr() == p_thr           # (first fit: the freed extent)
    more, outside = [], False                            # further generations while all are alive: the regions run dry
    for _ in range(4):                                   # (the third spectrum may still take the slot of the probe's input)
        X3, t3, thr3 = codec.encode(x)
        assert torch.equal(X3, Xr) and torch.equal(thr3, thrr)
        more.append((X3, thr3))
        outside = outside or not (b.value <= thr3.data_ptr() < b.value + nb.value)
    assert outside                                       # ... and plain allocations take over, same values
    del more, t3
    # the other entry points of the reference's API draw from the same pool
    Xt = codec.mdct.transform(x[:8])
    thrt = codec.psy.global_masking_threshold(Xt, codec.psy.tonality(Xt))
    assert torch.equal(Xt, Xt_ref) and torch.equal(thrt, thrt_ref)
    assert a.value <= Xt.data_ptr() < a.value + na.value and b.value <= thrt.data_ptr() < b.value + nb.value
    del X, X2, X3, thr2, thr3, xh, Xt, thrt, t, t2
    assert codec.placement_report()["live_tensors"] == 0
    placement.release()
    assert placement.report() is None


def test_pool_streams_and_graph_capture():
    """What torch's allocator does for its own blocks and the pool has to do itself: (1) an extent released
    after work on the DEFAULT stream -- whose handle is a null pointer -- is not "fresh" to a request on a side stream;
    (2) placement.record_stream: a consumer on a side stream that drops its reference early does not get the memory
    rewritten under it by the extent's next tenant; (3) under torch.cuda.graph nothing is carved out of the pool and no
    pool is built (captured addresses must stay valid for the replays); (4) no pool outside the configuration its effect
    was measured on."""
    from audiocodec_amd import placement
    placement.release()
    N, B, K, C = 1024, 80, 234, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    x = torch.rand(B, K * N, C, device="cuda") * 2 - 1
    # (4) a configuration without the fused wave-level encode: plain allocations, no probe inside the user's call
    c960 = audiocodec_amd.AudioCodec(48000, 960)
    c960.encode(torch.rand(160, 117 * 960, 2, device="cuda"))
    assert placement.report() is None
    # (3) capture before any pool exists: none is built
    xs = x[:4].contiguous()
    codec.encode(xs)
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        Xg, tg, thrg = codec.encode(x)
    assert placement.report() is None
    gr.replay()
    torch.cuda.synchronize()
    X, t, thr = codec.encode(x)                            # builds the pool
    assert torch.equal(X, Xg) and torch.equal(thr, thrg)
    pl = placement.pool(x.device)
    assert pl is not None and placement.report()["live_tensors"] == 2
    gr2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr2):                            # ... and with a pool: the capture's tensors are torch's own
        Xh, th, thrh = codec.encode(x)
    assert placement.report()["live_tensors"] == 2
    gr2.replay()
    torch.cuda.synchronize()
    assert torch.equal(Xh, X) and torch.equal(thrh, thr)
    del Xg, tg, thrg, Xh, th, thrh, gr, gr2
    # (1) default stream, then a side stream
    shape = tuple(thr.shape)
    p0 = thr.data_ptr()
    del thr                                                # released after default-stream work
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        a = placement.empty(placement.REGION_OTHER, shape, torch.float32, x.device)
    assert a.data_ptr() != p0                              # the freed extent waits for a request on ITS stream
    b = placement.empty(placement.REGION_OTHER, shape, torch.float32, x.device)
    assert b.data_ptr() == p0
    del a
    # (2) record_stream
    b.fill_(1.0)
    ev = torch.cuda.Event()
    ev.record()
    with torch.cuda.stream(side):
        side.wait_event(ev)
        torch.cuda._sleep(200_000_000)                     # ~0.1 s: the consumer's read is still queued when the producer moves on
        got = b[:2].sum()
    placement.record_stream(b, side)
    del b
    c = placement.empty(placement.REGION_OTHER, shape, torch.float32, x.device)   # the same extent again (default stream)
    assert c.data_ptr() == p0
    c.fill_(2.0)
    torch.cuda.synchronize()
    assert float(got) == float(2 * shape[1] * shape[2] * shape[3])
    del c, X, t
    placement.release()


def test_workspace_c_abi():
    """ac_workspace_* through ctypes alone (what a C caller has): placed buffers, fixed tensors of two copies, report."""
    import ctypes
    lib = _lib.load()
    N, B, K, C = 1024, 8, 16, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    dev_ = torch.device("cuda", torch.cuda.current_device())
    ws = ctypes.c_void_p()
    _lib.check(lib.ac_workspace_create(codec.mdct._plan(dev_), codec.psy._plan(dev_), B, K, C, 2, 3, 24.0, None, ctypes.byref(ws)))
    ptrs = [[ctypes.c_void_p() for _ in range(5)] for _ in range(2)]
    for cpy in range(2):
        _lib.check(lib.ac_workspace_buffers(ws, cpy, *[ctypes.byref(p_) for p_ in ptrs[cpy]]))
    assert lib.ac_workspace_buffers(ws, 2, None, None, None, None, None) == _lib.AC_EINVAL
    assert ptrs[0][0].value == ptrs[1][0].value and ptrs[0][1].value != ptrs[1][1].value      # one x, two X
    assert all(p_.value % (1 << 21) == 0 for row in ptrs for p_ in row)
    tries, chosen, spacer = ctypes.c_int(), ctypes.c_int(), ctypes.c_double()
    ms = (ctypes.c_float * 16)()
    _lib.check(lib.ac_workspace_report(ws, ctypes.byref(tries), ctypes.byref(chosen), ms, ctypes.byref(spacer)))
    assert 1 <= tries.value <= 3 and 0 <= chosen.value < tries.value and ms[chosen.value] > 0 and spacer.value <= 24.0
    # run the codec on raw pointers of copy 1 and compare with plain tensors (the probe left uniform noise in x)
    x_, X_, t_, thr_, xh_ = ptrs[1]
    _lib.check(lib.ac_encode_fused(codec.mdct._plan(dev_), codec.psy._plan(dev_), x_, X_, t_, thr_, 0.0, B, K, C, None))
    _lib.check(lib.ac_mdct_inverse(codec.mdct._plan(dev_), X_, xh_, B, K + 1, C, None))
    torch.cuda.synchronize()
    xs = torch.empty(B, K * N, C, device="cuda")
    import ctypes as ct
    hip = ct.CDLL("libamdhip64.so")
    assert hip.hipMemcpy(ct.c_void_p(xs.data_ptr()), x_, ct.c_size_t(xs.numel() * 4), 3) == 0
    assert float(xs.abs().max()) <= 1.0 and float(xs.std()) > 0.5
    Xp, tp, thrp = codec.encode(xs)
    Xw = torch.empty_like(Xp)
    assert hip.hipMemcpy(ct.c_void_p(Xw.data_ptr()), X_, ct.c_size_t(Xw.numel() * 4), 3) == 0
    assert torch.equal(Xw, Xp)
    assert lib.ac_workspace_destroy(ws) == 0


def _run_bench(*args):
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + [str(a) for a in args],
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert 1 <= len(lines) <= 2, out.stdout + out.stderr       # the side measurements, then the contract line (LAST)
    assert len(lines[-1]) < 2048, len(lines[-1])                  # the driver keeps the tail of stdout: the line must fit
    d = json.loads(lines[-1])
    if len(lines) == 2:
        d["_side"] = json.loads(lines[0])["bench_side"]
    return d


def test_bench_contract_line_on_a_small_workload():
    """bench.py end to end on a small workload: one JSON line with the contract's keys, the roofline object, the reduced
    checksums and a sane value (the full-size run is the driver's; this guards the script itself)"""
    d = _run_bench("--clips", 16, "--blocks", 32, "--steps", 5, "--warmup", 2, "--settle-ms", 5, "--no-cpu-baseline",
                   "--no-other-configs", "--no-workspace")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "reduced_over_ranks", "settle_ms",
              "timed_region_s", "cold_start_value", "encode_ms", "decode_ms", "caller_owned_value", "caller_owned_encode_ms"):
        assert k in d, k
    assert "cold_start" in d["_side"] and "kernels" in d["_side"] and "caller_owned" in d["_side"]
    assert "codec.encode()" in d["config"]["workload"]   # the headline is the reference's call shape: the library allocates its results
    assert abs(d["timed_region_s"] / (d["ms_per_step"] * d["steps"] * 1e-3) - 1.0) < 1e-5 and d["config"]["backend"] is None
    assert d["metric"].startswith("MDCT frames/s") and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 5
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["clips_per_gpu"] == 16 and "workload" in d["config"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert r["traffic"] is None and r["traffic_source"] is None   # measured traffic is on file for the full-size workload only
    assert d["value"] > 1e6 and d["reduced_over_ranks"]["round_trip_max_abs_err"] <= LSB
    assert d["reduced_over_ranks"]["frames_per_step"] == 16 * 2 * 32


def test_bench_two_ranks_on_one_gpu_equal_one_rank():
    """`python bench.py --gpus 2` starts its own two ranks (gloo when the box has fewer devices than ranks; both run the
    HIP kernels on the one GPU), prints n_gpus = 2, and the frame count / checksums of X, thr, PCM, tonality reduced over
    the ranks equal those of ONE rank over the same 16 clips: clips are independent (mdctransformer.py:292-295), sharding
    the batch axis changes no value."""
    two = _run_bench("--gpus", 2, "--clips", 8, "--blocks", 32, "--steps", 3, "--warmup", 1, "--settle-ms", 5)
    one = _run_bench("--gpus", 1, "--clips", 16, "--blocks", 32, "--steps", 3, "--warmup", 1, "--settle-ms", 5,
                     "--no-cpu-baseline", "--no-other-configs", "--no-workspace")
    assert two["n_gpus"] == 2 and one["n_gpus"] == 1
    assert two["config"]["clips_per_gpu"] == 8 and two["config"]["clips_total"] == 16
    a, b = two["reduced_over_ranks"], one["reduced_over_ranks"]
    assert a["frames_per_step"] == b["frames_per_step"] == 16 * 2 * 32
    for k in ("checksum_X", "checksum_thr", "checksum_pcm", "checksum_tonality"):
        assert abs(a[k] - b[k]) <= 1e-9 * max(1.0, abs(b[k])), (k, a[k], b[k])       # float64 sums of identical float32 values
    assert a["round_trip_max_abs_err"] <= LSB and a["round_trip_max_abs_err"] == b["round_trip_max_abs_err"]


def test_bench_one_rank_over_rccl():
    """The RCCL branch on the hardware there is: `python bench.py --gpus 1 --dist nccl` starts ONE rank under
    torch.distributed.run; its barrier, max-over-ranks time, checksum reductions and the closing barrier(device_ids=...) /
    destroy_process_group go through RCCL on the device -- the code the 1 / 2 / 4 / 8 curve depends on
    (no data-path collective: clips are independent, mdctransformer.py:292-295).  Same results bit for bit, and the same
    rate as the plain run on BASELINE configs[1] itself: the headline (library-placed results of codec.encode / decode)
    within 5 % -- two processes differ by up to 4 % on their own; the caller-owned figure is printed, not asserted (where
    the allocator puts X and thr moves that step by up to 11 % between two processes, and RCCL's own buffers shift every
    later allocation)."""
    common = ("--steps", 60, "--warmup", 5, "--no-cpu-baseline", "--no-other-configs", "--no-workspace", "--no-smi")
    nccl = _run_bench("--gpus", 1, "--dist", "nccl", *common)
    plain = _run_bench("--gpus", 1, *common)
    assert nccl["config"]["backend"] == "nccl" and plain["config"]["backend"] is None
    assert nccl["n_gpus"] == 1 and nccl["config"]["devices"] == 1
    a, b = nccl["reduced_over_ranks"], plain["reduced_over_ranks"]
    assert a["frames_per_step"] == b["frames_per_step"] == 256 * 2 * 468
    for k in ("checksum_X", "checksum_thr", "checksum_pcm", "checksum_tonality", "round_trip_max_abs_err"):
        assert a[k] == b[k], (k, a[k], b[k])
    print("one rank over RCCL %.1f M frames/s (caller-owned tensors %.1f M), plain %.1f M (%.1f M)"
          % (nccl["value"] / 1e6, nccl["caller_owned_value"] / 1e6, plain["value"] / 1e6, plain["caller_owned_value"] / 1e6))
    ratio = nccl["value"] / plain["value"]
    if abs(ratio - 1.0) >= 0.05:
        # one more pair before calling it a difference: a single process now and then lands 5 - 6 % off the others (placement of
        # the first allocations); the better of the two runs of each kind is compared
        nccl2, plain2 = _run_bench("--gpus", 1, "--dist", "nccl", *common), _run_bench("--gpus", 1, *common)
        ratio = max(nccl["value"], nccl2["value"]) / max(plain["value"], plain2["value"])
        print("second pair: over RCCL %.1f M, plain %.1f M" % (nccl2["value"] / 1e6, plain2["value"] / 1e6))
    assert abs(ratio - 1.0) < 0.05, (nccl["value"], plain["value"], ratio)
    # the driver's own form of the same thing: torch.distributed.run --nproc-per-node 1 bench.py --gpus 1
    import json
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
                          "127.0.0.1", "--master-port", "29631", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--clips", "16",
                          "--blocks", "32", "--no-encode-api"] + [str(v) for v in common], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["config"]["backend"] == "nccl" and d["n_gpus"] == 1


def test_bench_refuses_a_world_size_mismatch():
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1", "--warmup", "0"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert out.returncode != 0 and "WORLD_SIZE" in (out.stdout + out.stderr)


def test_config2_rank_share_512_clips():
    """BASELINE configs[2] is B = 4096 clips over 8 GPUs: one rank's share is 512 stereo clips of 10 s.  At that size:
    the round trip holds to 1 LSB, and the result equals, bit for bit, the two halves of the batch processed on their own
    (what another sharding of the same clips would compute)."""
    N, B, K, C = 1024, 512, 468, 2
    codec = audiocodec_amd.AudioCodec(48000, N)
    g = torch.Generator(device="cuda").manual_seed(4096)
    x = torch.empty(B, K * N, C, device="cuda").uniform_(-1, 1, generator=g)
    X, t, thr = codec.encode(x)
    xh = codec.decode(X)
    assert float((xh[:, N:-N] - x).abs().max()) <= LSB
    assert float(thr.min()) >= 1e-7 * (1 - 1e-6) and bool(torch.isfinite(thr).all())
    del xh
    for lo, hi in ((0, 256), (256, 512)):
        Xp, tp, thrp = codec.encode(x[lo:hi])
        assert torch.equal(Xp, X[lo:hi]) and torch.equal(tp, t[lo:hi]) and torch.equal(thrp, thr[lo:hi])
        del Xp, tp, thrp
