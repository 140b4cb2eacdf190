"""CPU timing leg of ``bench.py`` (``cpu_baseline``): the oracle's closed-form encode + decode on one core.

TEST / MEASUREMENT INFRASTRUCTURE ONLY -- never imported by the product.  ``bench.py`` starts one of these processes
per host core it wants to use (each pinned to one numpy / scipy thread) and adds the frame rates up.

    python -m oracle.cpu_bench --seconds 10 --seed 3      ->  one JSON line {"frames": ..., "seconds": ...}
"""
import os

for _v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
    os.environ[_v] = "1"

import argparse  # noqa: E402
import json  # noqa: E402
import time  # noqa: E402

import numpy as np  # noqa: E402
import scipy.fft  # noqa: E402

from oracle.audiocodec_oracle import MDCTOracle, PsychoOracle  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dense", action="store_true", help="reference-shaped flavour (dense polyphase / einsum forms)")
    args = ap.parse_args()
    N, B, K, C = 1024, 2, 46, 2
    om, op = MDCTOracle(N, "vorbis", np.float32), PsychoOracle(48000, N, 64, compute_dtype=np.float32)
    om.fft_workers = 1
    x = np.random.default_rng(1234 + args.seed).uniform(-1, 1, (B, K * N, C)).astype(np.float32)

    def one():
        X = om.transform(x, dense=args.dense)
        t = op.tonality(X)
        op.global_masking_threshold(X, t, dense=args.dense)
        om.inverse_transform(X, dense=args.dense)

    with scipy.fft.set_workers(1):
        one()
        reps, t0 = 0, time.perf_counter()
        while True:
            one()
            reps += 1
            el = time.perf_counter() - t0
            if el >= args.seconds:
                break
    print(json.dumps({"frames": reps * B * C * K, "seconds": el, "reps": reps, "B": B, "K": K, "C": C, "N": N}))


if __name__ == "__main__":
    main()
