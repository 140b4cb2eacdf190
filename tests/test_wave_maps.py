"""CPU: the index maps of the wave-level kernels (audiocodec_amd/csrc/ac_fast.hip), emulated lane by lane in numpy
(tests/emulate_wave_fft.py): LDS exchanges are bijections and bank-conflict-free under the gfx950 lane-group rules, the
in-wave FFT equals numpy's, and the fold / unfold register maps reproduce the oracle's transform for 8 and 16 points per
lane (filters_n 1024 and 2048)."""

import numpy as np
import pytest

import emulate_wave_fft as emu


def test_exchanges_are_bijective_and_conflict_free():
    lane = emu.lane
    for r in range(8):
        assert emu.cycles("write_b128", 16 * emu.ex1_write(r, lane)) == 8
        assert emu.cycles("read_b128", 16 * emu.ex1_read(lane, r)) == 4
        assert emu.cycles("write_b128", 16 * emu.ex2_write(lane, r)) == 8
        assert emu.cycles("read_b128", 16 * emu.ex2_read(lane, r)) == 4
        assert emu.cycles("write_b64", 8 * emu.rev_write(lane, r)) == 4
        assert emu.cycles("read_b64", 8 * emu.rev_read(lane, r)) == 2
    emu.check_banks()   # asserts the bijections


@pytest.mark.parametrize("R", [8, 16])
def test_wave_fft_equals_numpy(R):
    rng = np.random.default_rng(R)
    t = rng.standard_normal((64, R)) + 1j * rng.standard_normal((64, R))
    flat = np.zeros(64 * R, complex)
    flat[emu.lane[:, None] + 64 * np.arange(R)[None, :]] = t
    got, worst = emu.fft_on_wave_R(t, R)
    ref = np.fft.fft(flat)[emu.lane[:, None] + 64 * np.arange(R)[None, :]]
    assert np.max(np.abs(got - ref)) < 1e-10
    assert worst["ex2 write_b128"] == 8


@pytest.mark.parametrize("R,window", [(8, "vorbis"), (8, "sine"), (16, "vorbis"), (16, "sine")])
def test_fold_and_unfold_maps(R, window, capsys):
    emu.walks_R(R, window)
    out = capsys.readouterr().out
    err_a = float(out.split("analysis err")[1].split()[0])
    err_s = float(out.split("synthesis err")[1].split()[0])
    assert err_a < 1e-10 and err_s < 1e-8   # the synthesis side carries the reference's fp64 cancellation noise
