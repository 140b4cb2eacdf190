#!/bin/bash
# usage (GPU box): tools/profile_round.sh rNN   -- regenerates everything profiles/rNN holds, under gpurun_out/profile_rNN/:
#   bench_driver_style.json     python bench.py --steps 20 --warmup 5   (how the driver runs it; cpu_baseline, other_configs, workspace)
#   bench.json                  python bench.py   (defaults: 200 steps, 10 warmup)
#   bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the driver-style command (without the side measurements)
#   bench_kernel_stats_api_timed_region.csv   the same trace restricted to the `steps` timed dispatches of the HEADLINE loop
#                               (codec.encode() / decode(), library-placed results) and of the caller-owned loop before it
#   placement_12_runs.txt       twelve fresh driver-style processes: encode ms of the headline loop in each
#   pmc_encode.txt, pmc_decode.txt   rocprofv3 --pmc, separate passes (tools/pmc.sh)
#   traffic.json                FETCH_SIZE (doubled: gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE per launch
#   pmc_short_frames_n256.txt   the same counters for transform / inverse / threshold at filters_n = 256
#   lds_fft_tier_sizes.txt      the LDS-FFT tier over its sizes (stereo, mono, three channels; the A/B forms of the kernels)
#   entry_points.txt            every entry point on bench-sized workloads (tools/entry_points.sh)
# Copy the directory's files into profiles/rNN afterwards.
r=${1:-r4}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C audiocodec_amd/csrc || exit 1   # never build under the profiler
out=gpurun_out/profile_$r
mkdir -p $out
python bench.py --steps 20 --warmup 5 > $out/bench_driver_style.json 2> $out/bench_driver_style.err || exit 1
echo "driver-style bench done"
python bench.py --no-cpu-baseline --no-other-configs > $out/bench.json 2> $out/bench.err || exit 1
echo "default bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-workspace --no-encode-api --no-smi > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
cp $out/trace/*/*kernel_stats.csv $out/bench_kernel_stats.csv
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
line = json.loads(open(out + "/bench_under_rocprof.json").read().strip().splitlines()[-1])
steps = line["steps"]
rows = collections.defaultdict(list)
for f in glob.glob(out + "/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "ac::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void ac::(anonymous namespace)::", "").split("(")[0]
            rows[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
api_len = line["settle_steps"] + line["warmup"] + steps   # dispatches of the headline loop: settle, warm-up, timed
with open(out + "/bench_kernel_stats_api_timed_region.csv", "w") as fh:
    fh.write("Name,Loop,Calls,AverageNs,MinNs,MaxNs,note\n")
    for name, v in rows.items():
        v.sort()
        d = [x[1] for x in v]
        timed = d[-steps:]      # nothing launches these kernels after the headline loop (side measurements switched off)
        fh.write("%s,codec.encode()/decode() (headline),%d,%.1f,%d,%d,the last %d of %d dispatches; events in the same run: encode %.4f ms decode %.4f ms\n"
                 % (name, len(timed), sum(timed) / len(timed), min(timed), max(timed), steps, len(d), line["encode_ms"], line["decode_ms"]))
        co = d[-api_len - steps:-api_len]   # the caller-owned loop's timed dispatches: the `steps` right before the headline loop's settle phase
        if len(co) == steps:
            fh.write("%s,encode_into()/decode_into() (caller-owned),%d,%.1f,%d,%d,events in the same run: encode %.4f ms decode %.4f ms\n"
                     % (name, len(co), sum(co) / len(co), min(co), max(co), line["caller_owned_encode_ms"], line["caller_owned_decode_ms"]))
print(open(out + "/bench_kernel_stats_api_timed_region.csv").read())
PY
{
  echo "# twelve fresh processes of: python bench.py --steps 20 --warmup 5 (side measurements off): the headline loop's encode / decode ms,"
  echo "# value in M frames/s, and what the placement search did (tries, encode ms by try)"
  for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
    python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --no-workspace --no-smi 2>/dev/null | python -c "
import json, sys
ls = [l for l in sys.stdin.read().splitlines() if l.startswith('{')]
d = json.loads(ls[-1]); s = json.loads(ls[0])['bench_side'] if len(ls) > 1 else {}
p = s.get('placement') or {}
print('run %2d  encode %.4f ms  decode %.4f ms  value %.1f M  caller-owned %.1f M (encode %.4f ms)  tries %s chosen %s by try %s' % ($i, d['encode_ms'], d['decode_ms'], d['value'] / 1e6, d['caller_owned_value'] / 1e6, d['caller_owned_encode_ms'], p.get('tries'), p.get('chosen_try'), p.get('encode_ms_by_try')))"
  done
} > $out/placement_12_runs.txt
cat $out/placement_12_runs.txt
tools/pmc.sh encode ${r}_enc > /dev/null 2>&1 && cp gpurun_out/pmc_${r}_enc.txt $out/pmc_encode.txt || exit 1
echo "pmc encode done"
tools/pmc.sh inverse ${r}_dec > /dev/null 2>&1 && cp gpurun_out/pmc_${r}_dec.txt $out/pmc_decode.txt || exit 1
echo "pmc decode done"
python - $out <<'PY'
import json, re, sys
out = sys.argv[1]
N, B, K, C = 1024, 256, 468, 2
def counters(path):
    vals, take = {}, False
    for line in open(path):
        if line.startswith("("):
            take = True   # tools/pmc.sh prints one block per kernel; the driver runs a single kernel kind
            continue
        m = re.match(r"\s+(\w+)\s+([0-9.e+]+)", line)
        if m and take: vals[m.group(1)] = float(m.group(2))
    return vals
enc, dec = counters(out + "/pmc_encode.txt"), counters(out + "/pmc_decode.txt")
frames = B * C * K
def entry(v, kernel, alg):
    f, w = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    return {"kernel": kernel, "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
            "algorithmic_bytes_per_launch": alg}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), bench workload B=256 stereo "
                   "K=468 N=1024; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section); counters "
                   "are in KiB; memory-side requests include Infinity-Cache hits",
           "encode": entry(enc, "k_fwd_fast<8, 0, true, 4, 0, 2, false>", (12 * N + 4) * frames),
           "decode": entry(dec, "k_inv_fast<8, 0, 4, 0>", 8 * N * frames)}, open(out + "/traffic.json", "w"), indent=1)
print(open(out + "/traffic.json").read())
PY
# the several-frames-per-wave kernels and the masking model for general band layouts at filters_n = 256: counters + traffic
{
  for what in encode transform inverse psy; do
    N=256 tools/pmc.sh $what ${r}_n256_$what > /dev/null 2>&1
    echo "== filters_n = 256, B = 256 stereo, K = 1872: $what (algorithmic bytes per launch: $(python3 -c "
f = 256 * 2 * 1872
print({'encode': 3076 * f, 'transform': 2048 * f, 'inverse': 2048 * f, 'psy': 2052 * f}['$what'])"); FETCH_SIZE / WRITE_SIZE in KiB, FETCH_SIZE to be doubled on gfx950)"
    cat gpurun_out/pmc_${r}_n256_$what.txt
  done
} > $out/pmc_short_frames_n256.txt
echo "pmc short frames done"
{
  for what in encode; do
    N=512 tools/pmc.sh $what ${r}_n512_$what > /dev/null 2>&1
    echo "== filters_n = 512, B = 256 stereo, K = 936: fused $what (algorithmic bytes per launch: $((6148 * 256 * 2 * 936)))"
    cat gpurun_out/pmc_${r}_n512_$what.txt
  done
  for what in encode; do
    N=960 tools/pmc.sh $what ${r}_n960_$what > /dev/null 2>&1
    echo "== filters_n = 960 (LDS-FFT tier, fused encode k_enc_wave_v), B = 256 stereo, K = 500: $what (algorithmic bytes per launch: $((11524 * 256 * 2 * 500)))"
    cat gpurun_out/pmc_${r}_n960_$what.txt
  done
  for what in transform inverse; do
    N=960 B=64 tools/pmc.sh $what ${r}_n960_$what > /dev/null 2>&1
    echo "== filters_n = 960 (LDS-FFT tier, 16-byte kernels, one wave per frame), B = 64 stereo, K = 499: $what (algorithmic bytes per launch: $((7680 * 64 * 2 * 499)))"
    cat gpurun_out/pmc_${r}_n960_$what.txt
  done
  for what in transform inverse; do
    N=4096 B=64 tools/pmc.sh $what ${r}_n4096_$what > /dev/null 2>&1
    echo "== filters_n = 4096 (LDS-FFT tier, 16-byte kernels, four waves per frame, in place), B = 64 stereo, K = 117: $what (algorithmic bytes per launch: $((32768 * 64 * 2 * 117)))"
    cat gpurun_out/pmc_${r}_n4096_$what.txt
  done
} > $out/pmc_fused_n512_and_lds_fft_n960_n4096.txt
echo "pmc n512 / n960 / n4096 done"
# the LDS-FFT tier over its sizes: B = 64 stereo / B = 128 mono clips of 10 s, analysis and synthesis through the Python API
{
  echo "== LDS-FFT tier, float32, 64 stereo clips of 10 s (algorithmic 8 N bytes per frame; times include the result's allocation)"
  SIZES=16,24,32,48,96,120,160,192,240,320,384,480,576,640,768,800,960,1000,1152,1280,1536,1920,2304,2880,3072,3840,4096,5120,6144,7680,8192 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  echo "== the same, 128 mono clips"
  C=1 SIZES=32,120,240,480,960,1920,4096,8192 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  echo "== the same, 42 clips of 3 channels (channel pairs (0, 1), (2, -): the team form -- whole rows through LDS -- where it pays)"
  C=3 B=42 SIZES=120,480,960,1920,4096 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  echo "== the same, the strided channel pairs everywhere (AC_LDS_WAVE_NOTEAM=1)"
  AC_LDS_WAVE_NOTEAM=1 C=3 B=42 SIZES=120,480,960,1920,4096 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  echo "== 21 clips of 6 channels, team form where it pays / strided everywhere"
  C=6 B=21 SIZES=120,480,960,1024,1920,2048 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  AC_LDS_WAVE_NOTEAM=1 C=6 B=21 SIZES=120,480,960,1024,1920,2048 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  echo "== 64 stereo clips, the run-time form of the 16-byte kernels (AC_LDS_WAVE_NOCT=1) and the 8-byte kernels (AC_LDS_WAVE_NOVEC=1)"
  AC_LDS_WAVE_NOCT=1 SIZES=120,480,960 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
  AC_LDS_WAVE_NOVEC=1 SIZES=120,480,960,1920,4096 python tools/smooth_sizes_bench.py 2>&1 | grep "^N"
} > $out/lds_fft_tier_sizes.txt
echo "lds-fft sizes done"
tools/entry_points.sh > /dev/null 2>&1; cp gpurun_out/entry_points.txt $out/entry_points.txt
echo "entry points done"
ls -la $out
