"""CPU: the run-structured masking-model image (csrc/ac_psy_runs_dev.h) built by the library's host code, replayed lane by
lane in numpy (tests/emulate_runs.py) and held to the oracle -- band sums through the partial-sum lists, threshold entries
and per-bin look-ups -- at the sizes of every tier that runs the model (psychoacoustic.py:102-148, 169-210, 301-331)."""

import numpy as np
import pytest

from audiocodec_amd import _lib
from emulate_runs import runs_image, thresholds
from oracle.audiocodec_oracle import PsychoOracle


def _spectra(N, rng):
    f = np.arange(N)
    yield "white", rng.standard_normal(N).astype(np.float32) * 0.1
    yield "pink", (rng.standard_normal(N) / (1.0 + f / 8.0)).astype(np.float32)
    yield "tone", (1e-4 * rng.standard_normal(N) + (f == N // 3) * 0.7).astype(np.float32)
    yield "silence", np.zeros(N, dtype=np.float32)
    yield "top bin", ((f == N - 1) * 0.5).astype(np.float32)


@pytest.mark.parametrize("N,M,sr", [(64, 64, 48000), (128, 64, 48000), (256, 64, 48000), (512, 64, 48000), (960, 64, 48000),
                                    (480, 64, 44100), (1920, 64, 48000), (4096, 64, 48000), (100, 64, 48000), (512, 32, 48000),
                                    (36, 64, 16000), (1024, 48, 96000)])
def test_replayed_image_meets_the_oracle(N, M, sr):
    lib = _lib.load()
    got = runs_image(lib, N, M, sr, 0.6)
    assert got is not None, "the reference's Bark mapping has the run structure at float64 precompute"
    L, img = got
    assert L["slot"] % 16 == 0 and L["o4"] % 16 == 0 and L["o16"] % 16 == 0 and L["oz"] % 8 == 0
    assert L["kb"] >= 1 and 2 * L["lw"] <= 40
    o = PsychoOracle(sr, N, M, alpha=0.6, compute_dtype=np.float64)
    g = np.asarray(o._spreading_prototype() if hasattr(o, "_spreading_prototype") else _prototype(o, M), dtype=np.float64)
    rng = np.random.default_rng(N + M)
    for name, X in _spectra(N, rng):
        for drown in (0.0, 0.5):
            Xo = X.astype(np.float64).reshape(1, 1, N, 1)
            t = o.tonality(Xo)
            ref = o.global_masking_threshold(Xo, t, drown=drown)[0, 0, :, 0]
            thr = thresholds(L, img, N, M, 0.6, X, float(t[0, 0, 0, 0]), g, drown=drown)
            err = float(np.max(np.abs(thr - ref) / ref))
            assert err <= 2e-6, (name, drown, err)


def _prototype(o, M):
    # S[i][j] = g[M - i + j] (psychoacoustic.py:223-228): first column and last row of S give the 2M - 1 defined entries
    S = np.asarray(o.spreading_matrix, dtype=np.float64)
    g = np.zeros(2 * M)
    for d in range(-(M - 1), M):
        i, j = (0, d) if d >= 0 else (-d, 0)
        g[M + d] = S[i, j]
    return g


def test_tables_without_the_structure_are_refused():
    """float32 precompute at 8 kHz: interior weights of W are not exactly 1 / W_inv varies inside a band (measured when the
    form was designed); the builder must refuse, and the plan keeps the band walk."""
    lib = _lib.load()
    assert runs_image(lib, 1024, 64, 8000, 0.6, precompute=0) is None
    assert runs_image(lib, 1024, 64, 44100, 0.6, precompute=0) is not None


def test_longest_lists_stay_short():
    """what the form is for: the widest band's interior as a short list (the band walk took up to N / 15 steps)"""
    lib = _lib.load()
    for N, bound in [(256, 8), (512, 12), (960, 14), (1024, 12), (2048, 18), (4096, 20)]:
        L, _ = runs_image(lib, N, 64, 48000, 0.6)
        assert 2 * L["lw"] <= bound + 1, (N, L["lw"])
