#!/usr/bin/env python
"""Benchmark of the MDCT + psychoacoustic-masking hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path over one batch of synthetic PCM resident in HBM:
fused encode (x -> X, tonality, threshold; one HIP kernel) followed by decode (X -> x^; one HIP kernel).
Workload (BASELINE.json configs[1] at its roofline length): 256 stereo 48 kHz clips of
K = 468 blocks (10 s) per GPU, N = 1024 -- 3.9 GB touched per step, far beyond the 256 MiB Infinity Cache.
Clips are independent, so with N GPUs every rank processes its own 256 clips (weak scaling, no data-path
collective); RCCL is used only for the barrier and the max-over-ranks reduction of the elapsed time.

Prints ONE JSON line on rank 0.  value = frames/s over all GPUs, a frame being one 1024-sample hop of one
channel, encode + decode both done (20 484 algorithmic bytes per frame).
"""

import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import audiocodec_amd  # noqa: E402
from audiocodec_amd import dist as acd  # noqa: E402

N = 1024
ENC_BYTES = 4 * N + (4 * N + 4 * N + 4)      # PCM in; X, thr, tonality out           = 12 292 B / frame
DEC_BYTES = 4 * N + 4 * N                    # X in; PCM out                          =  8 192 B / frame
HBM_PEAK_GBS = 8000.0                        # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(seconds=12.0):
    """The oracle (closed-form numpy/scipy restatement, oracle/audiocodec_oracle.py) timed on the host cores on a
    bounded sample of the same workload: one single-threaded process per core (oracle/cpu_bench.py), each looping
    encode + decode over stereo clips of 46 blocks for `seconds`; frames/s summed over the processes.  Runs before
    this process touches the GPU (child processes are started from a GPU-free parent)."""
    cores = max(1, min(os.cpu_count() or 1, 16))
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    cmd = [sys.executable, "-m", "oracle.cpu_bench", "--seconds", str(seconds)]
    procs = [subprocess.Popen(cmd + ["--seed", str(i)], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True)
             for i in range(cores)]
    rate, frames = 0.0, 0
    for p in procs:
        out, _ = p.communicate(timeout=seconds * 6 + 120)
        r = json.loads(out.strip().splitlines()[-1])
        rate += r["frames"] / r["seconds"]
        frames += r["frames"]
    # the reference-shaped flavour (dense [2,N,N] polyphase products, 2N-point DCT-III, dense Bark einsums,
    # materialised 5-D masking tensor) on one core, for scale
    out = subprocess.run(cmd[:-1] + ["3", "--dense"], cwd=ROOT, env=env, stdout=subprocess.PIPE, text=True,
                         timeout=600).stdout
    d = json.loads(out.strip().splitlines()[-1])
    return {
        "value": rate, "unit": "frames/s", "cores": cores, "kind": "port",
        "sample": "%d single-threaded processes x %.0f s of (B=2 stereo clips, K=46 blocks, N=%d) encode+decode, "
                  "closed-form numpy/scipy oracle; %d frames in total" % (cores, seconds, N, frames),
        "reference_shaped_value_per_core": d["frames"] / d["seconds"],
        "reference_shaped_sample": "1 process, dense polyphase / DCT-III / einsum restatement of the reference's op "
                                   "sequence, %.1f s" % d["seconds"],
    }


def measured_traffic():
    """HBM bytes per launch of the fused encode kernel from the committed PMC passes (profiles/<round>/traffic.json:
    FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE), or None."""
    best = None
    prof = os.path.join(ROOT, "profiles")
    if os.path.isdir(prof):
        for r in sorted(os.listdir(prof)):
            f = os.path.join(prof, r, "traffic.json")
            if os.path.exists(f):
                with open(f) as fh:
                    best = json.load(fh)
    return best


def other_configs(dev):
    """BASELINE configs[3] and configs[4] timed beside the headline (rank 0, N = 1 only, outside the timed region; they
    are parity-test cases first -- tests/test_gpu_parity.py -- and these are their measured rates)."""
    out = {}
    # configs[3]: N = 2048 long-window MDCT + masking, Bark spreading as a band x band bf16 MFMA contraction
    n, B, K, C = 2048, 256, 234, 2
    x = torch.rand((B, K * n, C), device=dev) * 2 - 1
    X = torch.empty((B, K + 1, n, C), device=dev)
    t = torch.empty((B, K + 1, 1, C), device=dev)
    thr = torch.empty_like(X)
    ref = torch.empty_like(X)
    xh = torch.empty((B, (K + 2) * n, C), device=dev)

    def med(fn, reps=7):
        fn()
        ts = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        return float(np.median(ts))

    rows, codecs = {}, {}
    for mode in ("f32", "bf16x2_mfma", "bf16_mfma"):
        codecs[mode] = audiocodec_amd.AudioCodec(48000, n, spreading=mode)
        codecs[mode].encode_into(x, X, t, thr)
        if mode == "f32":
            ref.copy_(thr)
        rows[mode] = {"thr_max_rel_dev_vs_f32": float(((thr - ref).abs() / ref).max()), "encode_ms_rounds": []}
    for _ in range(3):   # interleaved rounds, median per form
        for mode, codec in codecs.items():
            rows[mode]["encode_ms_rounds"].append(med(lambda: codec.encode_into(x, X, t, thr), reps=5))
    for mode in rows:
        rows[mode]["encode_ms"] = float(np.median(rows[mode].pop("encode_ms_rounds")))
    codec = codecs["bf16x2_mfma"]
    dec = med(lambda: codec.decode_into(X, xh))
    fr = B * C * K
    out["configs[3]"] = {"workload": "batch=256 stereo 48 kHz clips, N=2048, K=234 blocks (10 s)", "spreading": rows,
                         "decode_ms": dec, "bytes_per_frame_encode": 12 * n + 4, "bytes_per_frame_decode": 8 * n,
                         "frames_per_s_bf16x2_mfma": fr / ((rows["bf16x2_mfma"]["encode_ms"] + dec) * 1e-3),
                         "encode_GBs_bf16x2_mfma": (12 * n + 4) * fr / (rows["bf16x2_mfma"]["encode_ms"] * 1e-3) / 1e9}
    del x, X, t, thr, ref, xh
    # configs[4]: streaming overlap-add, 10 min of stereo in chunks of 256 blocks through the device-resident state
    m = audiocodec_amd.MDCTransformer(N)
    Kt, k = 28125, 256
    xs = torch.rand((1, Kt * N, 2), device=dev) * 2 - 1
    st = audiocodec_amd.StreamingMDCT(m, 1, 2)

    def stream_pass():
        st.reset()
        for p in range(0, Kt, k):
            Xc = st.transform_chunk(xs[:, p * N:min(Kt, p + k) * N])
            st.inverse_chunk(Xc)

    ms = med(stream_pass, reps=3)
    st.close()
    out["configs[4]"] = {"workload": "1 stereo clip of 10 min (28125 blocks), chunks of 256 blocks, analysis + synthesis per chunk",
                         "ms_per_10_min": ms, "frames_per_s": 2 * Kt / (ms * 1e-3), "x_real_time": 600.0 / (ms * 1e-3)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)    # warm-up 10 iterations, then >= 200 timed
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-ms", type=float, default=0.0,
                    help="keep the device busy with the same step for this long before the warmup steps (an idle GPU "
                         "takes some tens of ms of load to reach its sustained clocks); reported as settle_ms")
    ap.add_argument("--clips", type=int, default=256, help="stereo clips per GPU")
    ap.add_argument("--blocks", type=int, default=468, help="blocks (hops) per clip; 468 = 10 s at 48 kHz")
    ap.add_argument("--placement-span-gib", type=float, default=112.0,
                    help="memory audiocodec_amd.Workspace may allocate for a moment while it looks for a good place for "
                         "the step's tensors; 0 = plain torch allocations, no placement probing")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the configs[3] / configs[4] side measurements")
    args = ap.parse_args()

    rank, world, _ = acd.env_rank_world()
    cpu = None
    if world == 1 and args.gpus == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()          # before this process initialises the GPU
    # RCCL ("nccl") carries the barrier and the scalar reductions; AC_BENCH_BACKEND=gloo rehearses the multi-rank path
    # on a box with fewer GPUs than ranks (ranks then share devices)
    backend = os.environ.get("AC_BENCH_BACKEND", "nccl")
    rank, world, local_rank = acd.init_process_group(backend if args.gpus > 1 else None)
    assert world == args.gpus or world == 1, "launch with torch.distributed.run --nproc-per-node %d" % args.gpus
    assert torch.cuda.is_available(), "bench.py needs the MI355X"
    dev = torch.device("cuda", (local_rank % torch.cuda.device_count()) if world > 1 else 0)
    torch.cuda.set_device(dev)

    B, K, C = args.clips, args.blocks, 2
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    codec = audiocodec_amd.AudioCodec(48000, N)
    assert codec.mdct.is_fast(dev) and codec.psy.is_fast(dev), "wave-level kernels not selected"
    spreading = codec.psy.plan_spreading(dev)
    placement = None
    if args.placement_span_gib > 0:
        # X and (thr, xh) in stretches of VRAM of different classes (audiocodec_amd/workspace.py: found by timing the
        # encode kernel with thr in each of a row of candidate chunks; the chunks not chosen are freed again)
        ws = audiocodec_amd.Workspace(codec, B, K, C, span_gib=args.placement_span_gib, device=dev)
        x, X, t, thr, xh = ws.x, ws.X, ws.t, ws.thr, ws.xh
        x.copy_(torch.rand((B, K * N, C), device=dev, generator=gen, dtype=torch.float32) * 2 - 1)
        placement = ws.report
    else:
        x = torch.rand((B, K * N, C), device=dev, generator=gen, dtype=torch.float32) * 2 - 1
        X = torch.empty((B, K + 1, N, C), device=dev, dtype=torch.float32)
        t = torch.empty((B, K + 1, 1, C), device=dev, dtype=torch.float32)
        thr = torch.empty((B, K + 1, N, C), device=dev, dtype=torch.float32)
        xh = torch.empty((B, (K + 2) * N, C), device=dev, dtype=torch.float32)

    def step():
        codec.encode_into(x, X, t, thr)
        codec.decode_into(X, xh)

    if args.settle_ms > 0:
        t_end = time.perf_counter() + args.settle_ms * 1e-3
        while time.perf_counter() < t_end:
            step()
            torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    # parity guard on the benchmark data itself: round trip within 1 LSB of int16
    err = float((xh[:, N:-N] - x).abs().max())
    assert err <= 1.0 / 32768.0, "round trip error %g" % err

    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(args.steps)]
    acd.reduce_scalars([0.0], "sum", device=dev)      # barrier (RCCL all-reduce of one scalar)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        ev[i][0].record()
        codec.encode_into(x, X, t, thr)
        ev[i][1].record()
        codec.decode_into(X, xh)
        ev[i][2].record()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    elapsed_max, = acd.reduce_scalars([elapsed], "max", device=dev)   # also the closing barrier
    enc_all = [e[0].elapsed_time(e[1]) for e in ev]
    dec_all = [e[1].elapsed_time(e[2]) for e in ev]
    enc_ms, dec_ms = float(np.mean(enc_all)), float(np.mean(dec_all))

    frames_rank = B * C * K
    frames_total = frames_rank * world * args.steps
    value = frames_total / elapsed_max
    if rank == 0:
        tr = measured_traffic() if (B, K) == (256, 468) else None
        traffic = tr["encode"]["hbm_bytes_per_launch"] if tr else None
        enc_gbs = ENC_BYTES * frames_rank / (enc_ms * 1e-3) / 1e9
        dec_gbs = DEC_BYTES * frames_rank / (dec_ms * 1e-3) / 1e9
        out = {
            "metric": "MDCT frames/s (48 kHz, N=1024) encode+decode",
            "value": value, "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle_ms": args.settle_ms,
            "ms_per_step": elapsed_max / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: batch=%d stereo 48 kHz clips per GPU, N=1024, K=%d blocks "
                                   "(%.1f s), fused MDCT+tonality+masking encode then IMDCT decode" % (B, K, K * N / 48000.0),
                       "clips_per_gpu": B, "channels": C, "blocks": K, "filters_n": N, "sample_rate": 48000,
                       "sharding": "clips split across ranks, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": "k_fwd_fast<8, 0, true, 4, 0, %d> (fused encode, spreading product: %s)"
                                   % (audiocodec_amd.PsychoacousticModel.SPREADING[spreading], spreading),
                         "achieved": enc_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": enc_gbs / HBM_PEAK_GBS,
                         "traffic": traffic, "bytes_per_frame": ENC_BYTES, "frames_per_launch": frames_rank,
                         "avg_launch_ms": enc_ms},
            "kernels": {"encode_ms": enc_ms, "encode_GBs": enc_gbs, "decode_ms": dec_ms, "decode_GBs": dec_gbs,
                        "encode_ms_median": float(np.median(enc_all)), "encode_ms_min": float(np.min(enc_all)),
                        "decode_ms_median": float(np.median(dec_all)), "decode_ms_min": float(np.min(dec_all)),
                        "step_frac_of_hbm_peak": (ENC_BYTES + DEC_BYTES) * (value / world) / (HBM_PEAK_GBS * 1e9),
                        # the same against the measured streaming-copy rate, and the read-only share
                        "step_frac_of_achievable_6290_GBs": (ENC_BYTES + DEC_BYTES) * (value / world) / 6.29e12,
                        "step_read_share_of_hbm_peak": 2 * 4 * N * (value / world) / (HBM_PEAK_GBS * 1e9)},
            "round_trip_max_abs_err": err,
            "placement": placement,
        }
        if placement is not None and world == 1:
            # the same step on plain torch allocations (what a caller gets without audiocodec_amd.Workspace), for scale
            try:
                px = x.clone()
                pX, pt, pthr, pxh = torch.empty_like(X), torch.empty_like(t), torch.empty_like(thr), torch.empty_like(xh)
                for _ in range(args.warmup):
                    codec.encode_into(px, pX, pt, pthr)
                    codec.decode_into(pX, pxh)
                torch.cuda.synchronize()
                p0 = time.perf_counter()
                for _ in range(args.steps):
                    codec.encode_into(px, pX, pt, pthr)
                    codec.decode_into(pX, pxh)
                torch.cuda.synchronize()
                pdt = time.perf_counter() - p0
                out["plain_allocations"] = {"value": frames_rank * args.steps / pdt, "ms_per_step": pdt / args.steps * 1e3}
                del px, pX, pt, pthr, pxh
            except Exception as e:
                out["plain_allocations"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if world == 1 and not args.no_other_configs:
            del x, X, t, thr, xh
            if placement is not None:
                del ws
            try:
                out["other_configs"] = other_configs(dev)
            except Exception as e:   # side measurements must never cost the headline line
                out["other_configs"] = {"error": "%s: %s" % (type(e).__name__, e)}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
