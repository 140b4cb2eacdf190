#!/bin/bash
# Timings of every entry point on the bench-sized workloads -> gpurun_out/entry_points.txt (copied to profiles/)
out=gpurun_out/entry_points.txt
{
  echo "(encode_into on plain torch allocations; transform / inverse / threshold allocate their results: placed by audiocodec_amd/placement.py once a large encode() has run)"
  echo "== N = 1024, B = 256 stereo, K = 468 (tools/microbench.py)"; python tools/microbench.py 2>/dev/null
  echo; echo "== N = 2048, B = 256 stereo, K = 234"; N=2048 python tools/microbench.py 2>/dev/null
  echo; echo "== N = 1024, B = 256 mono, K = 468"; C=1 python tools/microbench.py 2>/dev/null
  echo; echo "== N = 512 (two frames per wave), B = 256 stereo, K = 936"; N=512 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 960 (LDS-FFT tier), B = 64 stereo, K = 499"; N=960 B=64 K=499 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 960, B = 256 stereo, K = 499"; N=960 python tools/microbench.py 2>/dev/null | sed -n 1,6p
  echo; echo "== N = 960, B = 256 mono, K = 499"; N=960 C=1 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 960, B = 84 clips of six channels, K = 499 (whole rows through LDS: team form of the LDS-FFT tier where it pays, k_psy_runs_c for the masking model)"; N=960 C=6 B=84 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 1024, B = 84 clips of six channels, K = 468 (the same)"; N=1024 C=6 B=84 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 960, B = 170 clips of three channels, K = 499"; N=960 C=3 B=170 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 960 six channels, the strided channel pairs everywhere (AC_LDS_WAVE_NOTEAM=1)"; AC_LDS_WAVE_NOTEAM=1 N=960 C=6 B=84 python tools/microbench.py 2>/dev/null | sed -n 2,3p
  echo; echo "== N = 1920 (LDS-FFT tier, two waves per frame; masking model with 16 granule registers), B = 256 stereo, K = 249"; N=1920 python tools/microbench.py 2>/dev/null | sed -n 1,6p
  echo; echo "== N = 4096 (four waves per frame; 32 granule registers), B = 256 stereo, K = 117"; N=4096 python tools/microbench.py 2>/dev/null | sed -n 1,6p
  echo; echo "== N = 512, B = 256 mono, K = 936"; N=512 C=1 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 256 (four frames per wave), B = 256 stereo, K = 1872"; N=256 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 128 (eight frames per wave), B = 256 stereo, K = 3744"; N=128 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 64 (sixteen frames per wave), B = 256 stereo, K = 7488"; N=64 python tools/microbench.py 2>/dev/null | sed -n 1,5p
  echo; echo "== N = 32 (LDS-FFT tier), B = 256 stereo, K = 14976"; N=32 python tools/microbench.py 2>/dev/null | sed -n 2,3p
  echo; echo "== backward passes, N = 1024, B = 256 stereo, K = 468 (tools/bwd_bench.py)"; python tools/bwd_bench.py 2>/dev/null
  echo; echo "== element-wise utilities on X [256, 469, 1024, 2] (tools/ew_bench.py)"; python tools/ew_bench.py 2>/dev/null
} > $out
cat $out
