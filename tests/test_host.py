"""CPU: host logic of the product -- the C-ABI library loads and exports every symbol the header declares,
host-side constant builders match the golden tables, argument validation, sharding helpers (gloo, world 2)."""

import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from conftest import ROOT

import audiocodec_amd
from audiocodec_amd import _lib
from audiocodec_amd.dist import clip_range


def _header_symbols(names=("audiocodec_amd.h", "audiocodec_amd_testing.h")):
    syms = set()
    for name in names:
        text = open(os.path.join(ROOT, "include", name)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        syms |= set(re.findall(r"\b(ac_[a-z0-9_]+)\s*\(", text))
    return sorted(syms)


def test_library_exports_every_header_symbol():
    lib = _lib.load()
    syms = _header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), "libaudiocodec_amd.so does not export %s" % s
    assert sorted(_lib.PROTOTYPES) == syms, "ctypes prototypes and header disagree"
    assert lib.ac_version() == 171
    assert _header_symbols(("audiocodec_amd_testing.h",)) == ["ac_set_force_generic", "ac_testing_runs_image"]
    assert not set(_header_symbols(("audiocodec_amd_testing.h",))) & set(_header_symbols(("audiocodec_amd.h",)))


def test_library_exports_nothing_but_the_c_abi():
    """-fvisibility=hidden: the dynamic symbol table of the library defines the ac_* entry points of include/*.h and
    nothing else -- no mangled C++ internals for another library to collide with or a caller to depend on."""
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    defined = sorted(ln.split()[-1] for ln in out.stdout.splitlines() if ln.strip())
    assert defined == _header_symbols(), sorted(set(defined) ^ set(_header_symbols()))


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "-S", _lib.LIB_PATH], capture_output=True, text=True)
    assert ".hip_fatbin" in out.stdout
    data = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in data


@pytest.mark.parametrize("N,wt", [(8, "vorbis"), (16, "sine"), (16, "rect"), (12, "vorbis")])
def test_dense_matrices_match_oracle(N, wt):
    from oracle.audiocodec_oracle import MDCTOracle
    m = audiocodec_amd.MDCTransformer(N, window_type=wt)
    o = MDCTOracle(N, wt, np.float64)
    np.testing.assert_allclose(m.H.numpy(), o.dense_H().astype(np.float32), atol=1e-7)
    np.testing.assert_allclose(m.H_inv.numpy(), o.dense_H_inv().astype(np.float32), atol=1e-6)
    assert m.H.shape == (2, N, N) and int((m.H != 0).sum()) <= 2 * N


def test_dense_H_matches_reference_golden(golden):
    g = golden("mdct_n8_H")
    m = audiocodec_amd.MDCTransformer(8)
    np.testing.assert_allclose(m.H.numpy(), g["H"].astype(np.float32), atol=1e-7)
    np.testing.assert_allclose(m.H_inv.numpy(), g["H_inv"].astype(np.float32), atol=1e-6)


@pytest.mark.parametrize("N,wt", [(1024, "vorbis"), (1024, "sine"), (64, "rect"), (2048, "Vorbis")])
def test_fold_coefficients_match_oracle(N, wt):
    from oracle.audiocodec_oracle import fold_coefficients
    c = audiocodec_amd.MDCTransformer(N, window_type=wt).fold_coefficients()
    o = fold_coefficients(N, wt)
    for i, k in enumerate(["a1", "a2", "a3", "a4", "s1", "s2", "s3", "s4"]):
        np.testing.assert_allclose(c[i], o[k], rtol=1e-12, atol=1e-15)


def _dense(idx, val, shape):
    m = np.zeros(shape)
    m[idx[:, 0], idx[:, 1]] = val
    return m


@pytest.mark.parametrize("sr,N,M", [(48000, 1024, 64), (48000, 2048, 64), (32768, 64, 64), (44100, 256, 48)])
def test_psy_tables_match_reference_golden(golden, sr, N, M):
    g = golden("psy_%d_%d_%d_tables" % (sr, N, M))
    p = audiocodec_amd.PsychoacousticModel(sr, filter_bands_n=N, bark_bands_n=M)
    np.testing.assert_array_equal(p.W.numpy(), _dense(g["W_idx"], g["W_val"], (N, M)).astype(np.float32))
    np.testing.assert_array_equal(p.W_inv.numpy(), _dense(g["W_inv_idx"], g["W_inv_val"], (M, N)).astype(np.float32))
    np.testing.assert_allclose(p.spreading_matrix.numpy(), g["S"].astype(np.float32), rtol=2e-7)
    np.testing.assert_allclose(p.quiet_threshold_intensity.numpy().reshape(-1), g["quiet"].astype(np.float32), rtol=2e-7)
    assert abs(float(p.max_bark) - float(g["max_bark"])) < 1e-13
    assert abs(float(p.bark_band_width) - float(g["bark_band_width"])) < 1e-14
    assert float(p._dB_MIN) == -20.0
    assert tuple(p.quiet_threshold_intensity.shape) == (1, 1, M, 1)


def test_energy_conservation_like_reference():
    """tests/test_psychoacoustic.py:14-30 on the product's tables"""
    p = audiocodec_amd.PsychoacousticModel(sample_rate=32768, filter_bands_n=64)
    assert float(torch.sum(torch.abs(torch.sum(p.W, dim=1) - 1.0))) < 1e-6
    assert float(torch.sum(torch.abs(torch.sum(p.W_inv, dim=1) - 1.0))) < 1e-6


def test_constructor_validation():
    with pytest.raises(AssertionError):
        audiocodec_amd.MDCTransformer(7)
    with pytest.raises(TypeError):
        audiocodec_amd.PsychoacousticModel(48000, compute_dtype=torch.float16)
    # float16: the filter bank takes it (mdctransformer.py:327-344 up-casts inside the DCT-IV), the masking model refuses it by
    # name (psychoacoustic.py:42-43), and a codec needs both
    assert audiocodec_amd.MDCTransformer(8, compute_dtype=torch.float16)._dtype_id == 3
    with pytest.raises(TypeError):
        audiocodec_amd.AudioCodec(48000, 64, compute_dtype=torch.float16)
    p64 = audiocodec_amd.PsychoacousticModel(48000, compute_dtype=torch.float64)
    assert p64.W.dtype == torch.float64 and p64._dB_MAX.dtype == torch.float64
    assert float((p64.W.float() - audiocodec_amd.PsychoacousticModel(48000).W).abs().max()) < 1e-7
    assert audiocodec_amd.PsychoacousticModel(48000, compute_dtype="bfloat16").W.dtype == torch.float32
    assert audiocodec_amd.MDCTransformer(8, window_type=None)._window == 2
    assert audiocodec_amd.MDCTransformer(8, window_type="SINE")._window == 1
    assert audiocodec_amd.MDCTransformer(8, window_type="hann")._window == 2
    lib = _lib.load()
    out = ctypes.c_void_p()
    assert lib.ac_mdct_plan_create(7, 0, 0, ctypes.byref(out)) == _lib.AC_EINVAL
    assert b"even" in lib.ac_last_error()
    with pytest.raises(ValueError):
        _lib.check(lib.ac_mdct_fold_coefficients_host(8, 9, (ctypes.c_double * 32)()))
    assert lib.ac_mdct_forward_typed(None, None, None, 9, 1, 1, 1, None) == _lib.AC_EINVAL
    assert b"AC_F32" in lib.ac_last_error()
    best = ctypes.c_int(-1)
    assert lib.ac_probe_placement(None, None, None, None, None, None, 0, 1, 1, 2, None, ctypes.byref(best), None) == _lib.AC_EINVAL
    assert lib.ac_psy_plan_create_ex(1024, 64, 48000.0, 0.6, 0, 7, ctypes.byref(out)) == _lib.AC_EINVAL
    assert b"AC_SPREAD" in lib.ac_last_error()
    with pytest.raises(ValueError):
        audiocodec_amd.PsychoacousticModel(48000, spreading="fp8")
    assert audiocodec_amd.PsychoacousticModel(48000, spreading="bf16x2_mfma").spreading == "bf16x2_mfma"


def test_no_cpu_fallback():
    m = audiocodec_amd.MDCTransformer(64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m.transform(torch.zeros(1, 128, 1))
    with pytest.raises(ValueError):
        m.transform(torch.zeros(1, 128, 1, dtype=torch.float64))
    p = audiocodec_amd.PsychoacousticModel(48000, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p.tonality(torch.zeros(1, 2, 64, 1))
    import audiocodec_amd.mdctransformer as prod
    src = open(prod.__file__).read() + open(audiocodec_amd.psychoacoustic.__file__).read()
    assert "oracle" not in src


def test_import_path_shim():
    from audiocodec.mdctransformer import MDCTransformer
    from audiocodec import psychoacoustic
    assert MDCTransformer is audiocodec_amd.MDCTransformer
    assert psychoacoustic.PsychoacousticModel is audiocodec_amd.PsychoacousticModel


def test_clip_range_partitions():
    for B in (0, 1, 7, 256, 4096):
        for W in (1, 2, 3, 8):
            spans = [clip_range(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(spans[i][1] == spans[i + 1][0] for i in range(W - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert clip_range(4096, 3, 8) == (1536, 2048)


_WORKER = r"""
import os, sys
sys.path.insert(0, %r)
import numpy as np, torch
from audiocodec_amd import dist as acd
from oracle.audiocodec_oracle import MDCTOracle
rank, world, _ = acd.init_process_group("gloo")
B = 6
x = np.random.default_rng(1234).uniform(-1, 1, (B, 4 * 64, 2)).astype(np.float32)   # same on every rank
lo, hi = acd.clip_range(B, rank, world)
X = MDCTOracle(64).transform(x[lo:hi])            # the shard this rank owns (CPU stand-in for the HIP call)
frames, checksum = acd.reduce_scalars([float((hi - lo) * 2 * 4), float(np.sum(X.astype(np.float64)))], "sum")
tmax, = acd.reduce_scalars([1.0 + rank], "max")
if rank == 0:
    full = float(np.sum(MDCTOracle(64).transform(x).astype(np.float64)))
    assert frames == B * 2 * 4, frames
    assert abs(checksum - full) < 1e-9, (checksum, full)
    assert tmax == float(world)
    print("OK", frames, tmax)
"""


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_sharding_world_size_2_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER % ROOT)
    out = None
    for attempt in range(3):   # a rendezvous port can be taken between the probe and torchrun's bind: try another one
        port = str(_free_port())
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
        out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                              "--master-addr", "127.0.0.1", "--master-port", port, str(script)],
                             capture_output=True, text=True, env=env, timeout=300)
        if out.returncode == 0:
            break
    assert out.returncode == 0, out.stdout + out.stderr
    assert "OK 48.0 2.0" in out.stdout


def test_host_table_builders_under_sanitizers(tmp_path):
    """The host-side constant builders (ac_tables.cpp: windows, fold coefficients, Bark tables, CSR forms) compiled with
    AddressSanitizer + UBSan and swept over sizes / band counts / sample rates (tests/native/sanitize_tables.cpp).  GPU
    sanitizers are not available on this pool; this is the part of the native code a CPU sanitizer can see."""
    exe = str(tmp_path / "sanitize_tables")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all",
           "-I" + os.path.join(ROOT, "audiocodec_amd", "csrc"), os.path.join(ROOT, "tests", "native", "sanitize_tables.cpp"),
           os.path.join(ROOT, "audiocodec_amd", "csrc", "ac_tables.cpp"), "-o", exe]
    build = subprocess.run(cmd, capture_output=True, text=True)
    if build.returncode != 0 and "sanitize" in build.stderr and "cannot find" in build.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert build.returncode == 0, build.stderr
    run = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert run.returncode == 0 and run.stdout.strip() == "ok", run.stdout + run.stderr


# ---- bench.py's rank launching (the data path has no collective: mdctransformer.py:292-295) ----------------
def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("ac_bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_bench_launch_plan_backend_by_device_count():
    """`python bench.py --gpus N` starts N ranks under torch.distributed.run: RCCL ("nccl") when the box has a device per
    rank, gloo (ranks share devices; rehearsal) otherwise; an explicit --dist / AC_BENCH_BACKEND is passed through."""
    b = _bench_module()
    for gpus, have, want in ((8, 8, "nccl"), (2, 8, "nccl"), (1, 1, "nccl"), (2, 1, "gloo"), (8, 1, "gloo")):
        p = b.plan_launch(gpus, have, "auto", {}, ["--gpus", str(gpus)], 12345)
        assert p["backend"] == want and p["env"]["AC_BENCH_BACKEND"] == want
        cmd = p["cmd"]
        assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nnodes=1" in cmd
        assert cmd[cmd.index("--nproc-per-node") + 1] == str(gpus)
        assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "12345"
        assert cmd[-2:] == ["--gpus", str(gpus)] and cmd[-3].endswith("bench.py")
    assert b.plan_launch(2, 1, "nccl", {}, [], 1)["backend"] == "nccl"            # asked for: the ranks refuse, not the planner
    assert b.plan_launch(2, 8, "auto", {"AC_BENCH_BACKEND": "gloo"}, [], 1)["backend"] == "gloo"


def test_bench_dry_run_launch_prints_the_plan():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-launch"],
                         capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stdout + out.stderr
    import json
    plan = json.loads(out.stdout.strip().splitlines()[-1])
    assert plan["ranks"] == 2 and plan["backend"] == ("nccl" if plan["devices"] >= 2 else "gloo")
    assert "--dry-run-launch" not in plan["cmd"]


def test_bench_rank_without_a_device_refuses_nccl():
    """A rank of an RCCL run whose box has fewer devices than ranks exits non-zero before any process group exists."""
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0", AC_BENCH_BACKEND="nccl", MASTER_ADDR="127.0.0.1",
               MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                          "--no-cpu-baseline", "--no-smi"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box has a device per rank")
    assert out.returncode != 0
    assert "MI355X" in (out.stdout + out.stderr) or "devices" in (out.stdout + out.stderr)


def test_one_rank_process_group_goes_through_the_backend(tmp_path):
    """init_process_group(force=True) gives ONE rank a real group: the same barrier / all-reduce calls an N-rank run issues
    (gloo here; `-m gpu` runs the RCCL form on the device)."""
    script = tmp_path / "one.py"
    script.write_text("import sys\nsys.path.insert(0, %r)\nimport torch.distributed as d\nfrom audiocodec_amd import dist as acd\n"
                      "r, w, _ = acd.init_process_group('gloo', force=True)\nassert d.is_initialized() and d.get_world_size() == 1\n"
                      "assert acd.reduce_scalars([2.5, 4.0], 'sum') == [2.5, 4.0]\nassert acd.reduce_scalars([7.0], 'max') == [7.0]\n"
                      "acd.barrier()\nd.destroy_process_group()\nprint('OK', r, w)\n" % ROOT)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    out = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, env=env, timeout=300)
    assert out.returncode == 0 and "OK 0 1" in out.stdout, out.stdout + out.stderr


# ---- precompute_dtype (mdctransformer.py:13-14,31-35; psychoacoustic.py:14-15) ---------------------------------------
def test_precompute_dtype_float32_host_tables(golden):
    """The constructors take the reference's precompute_dtype keyword: float64 (default) or float32.  In float32 the
    host builders round every constant and operation to float32 in the reference's order: the cancellation of
    mdctransformer.py:218-221 yields exactly 0 for j = 0 at N = 64, the dense H / H_inv meet the reference's own
    float32-precompute matrices to 2 ulp (glibc's sinf against numpy's), the Bark tables to what one ulp of max_bark
    moves them (the float32 Bark mapping is that coarse: its own distance to the float64 tables is 4e-4 in W)."""
    g = golden("precompute_float32_cases")
    m = audiocodec_amd.MDCTransformer(64, precompute_dtype=torch.float32)
    assert m.precompute_dtype == torch.float32
    c = m.fold_coefficients()
    assert c[1][0] == 0.0 and audiocodec_amd.MDCTransformer(64).fold_coefficients()[1][0] < -2e-4      # a2[0]
    np.testing.assert_allclose(m.H.numpy(), g["n64_H"], rtol=0, atol=3e-7)
    np.testing.assert_allclose(m.H_inv.numpy(), g["n64_H_inv"], rtol=0, atol=5e-7)
    from oracle.audiocodec_oracle import fold_coefficients
    o = fold_coefficients(64, "vorbis", np.float32)
    for i, k in enumerate(["a1", "a2", "a3", "a4", "s1", "s2", "s3", "s4"]):
        np.testing.assert_allclose(c[i], o[k].astype(np.float64), rtol=0, atol=5e-7)
    for wt in ("sine", "rect"):
        mw = audiocodec_amd.MDCTransformer(16, window_type=wt, precompute_dtype="float32")
        np.testing.assert_allclose(mw.H.numpy(), g["n16_%s_H" % wt], rtol=0, atol=3e-7)
        np.testing.assert_allclose(mw.H_inv.numpy(), g["n16_%s_H_inv" % wt], rtol=0, atol=5e-7)
    for sr, N, M in ((48000, 1024, 64), (32768, 64, 64)):
        tag = "psy_%d_%d_%d_" % (sr, N, M)
        p = audiocodec_amd.PsychoacousticModel(sr, N, M, precompute_dtype=torch.float32)
        assert p.max_bark.dtype == torch.float32 and abs(float(p.max_bark) - float(g[tag + "max_bark"])) <= 4e-6
        np.testing.assert_allclose(p.W.numpy(), _dense(g[tag + "W_idx"], g[tag + "W_val"], (N, M)), rtol=0, atol=2e-3)
        np.testing.assert_allclose(p.W_inv.numpy(), _dense(g[tag + "W_inv_idx"], g[tag + "W_inv_val"], (M, N)), rtol=0, atol=1e-4)
        np.testing.assert_allclose(p.spreading_matrix.numpy(), g[tag + "S"], rtol=1e-4)
        np.testing.assert_allclose(p.quiet_threshold_intensity.numpy().reshape(-1), g[tag + "quiet"], rtol=2e-4)
        assert float(torch.sum(torch.abs(torch.sum(p.W, dim=1) - 1.0))) < 1e-3          # rows still sum to one
    with pytest.raises(NotImplementedError):
        audiocodec_amd.MDCTransformer(64, precompute_dtype=torch.bfloat16)
    with pytest.raises(NotImplementedError):
        audiocodec_amd.PsychoacousticModel(48000, precompute_dtype=torch.float16)
    lib = _lib.load()
    out = ctypes.c_void_p()
    assert lib.ac_mdct_plan_create_pre(64, 0, 7, 0, ctypes.byref(out)) == _lib.AC_EINVAL and b"precompute" in lib.ac_last_error()
    assert lib.ac_psy_plan_create_pre(64, 64, 48000.0, 0.6, 0, -2, 1, ctypes.byref(out)) == _lib.AC_EINVAL
