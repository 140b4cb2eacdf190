#!/bin/bash
# usage (GPU box): tools/profile_round.sh rNN   -- regenerates everything profiles/rNN holds, under gpurun_out/profile_rNN/:
#   bench.json                  python bench.py (the contract line, with cpu_baseline and other_configs)
#   bench_kernel_stats.csv      rocprofv3 --kernel-trace --stats of the same command (without the side measurements)
#   pmc_encode.txt, pmc_decode.txt   rocprofv3 --pmc, separate passes (tools/pmc.sh)
#   traffic.json                FETCH_SIZE (doubled: gfx950 correction of MI355X_MICROARCH.md) + WRITE_SIZE per launch
#   entry_points.txt            every entry point on bench-sized workloads (tools/entry_points.sh)
# Copy the directory's files into profiles/rNN afterwards.
r=${1:-r1}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profile_$r
mkdir -p $out
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python bench.py --no-cpu-baseline --no-other-configs > $out/bench_under_rocprof.json 2> $out/trace.log || exit 1
cp $out/trace/*/*kernel_stats.csv $out/bench_kernel_stats.csv
# bench.py also launches the encode kernel while audiocodec_amd.Workspace probes candidate placements, so the whole-run
# averages above mix placements; the timed region is the last `steps` dispatches of each kernel of the traced run
python - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
steps = json.loads(open(out + "/bench_under_rocprof.json").read().strip().splitlines()[-1])["steps"]
rows = collections.defaultdict(list)
for f in glob.glob(out + "/trace/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "ac::" in r["Kernel_Name"]:
            name = r["Kernel_Name"].replace("void ac::(anonymous namespace)::", "").split("(")[0]
            rows[name].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
with open(out + "/bench_kernel_stats_timed_region.csv", "w") as fh:
    fh.write("Name,Calls,AverageNs,MinNs,MaxNs,note\n")
    for name, v in rows.items():
        v.sort()
        d = [x[1] for x in v]
        # plain-allocation side measurement (warmup + steps launches) runs last; the timed region is the `steps` before it
        warm = json.loads(open(out + "/bench_under_rocprof.json").read().strip().splitlines()[-1])["warmup"]
        tail = steps + warm
        timed = d[-(tail + steps):-tail] if len(d) >= tail + steps else d[-steps:]
        fh.write("%s,%d,%.1f,%d,%d,timed region of bench.py (dispatches %d..%d of %d)\n"
                 % (name, len(timed), sum(timed) / len(timed), min(timed), max(timed), len(d) - tail - steps, len(d) - tail, len(d)))
print(open(out + "/bench_kernel_stats_timed_region.csv").read())
PY
tools/pmc.sh encode ${r}_enc > /dev/null 2>&1 && cp gpurun_out/pmc_${r}_enc.txt $out/pmc_encode.txt || exit 1
tools/pmc.sh inverse ${r}_dec > /dev/null 2>&1 && cp gpurun_out/pmc_${r}_dec.txt $out/pmc_decode.txt || exit 1
python - $out <<'PY'
import json, re, sys
out = sys.argv[1]
N, B, K, C = 1024, 256, 468, 2
def counters(path, kernel):
    vals, take = {}, False
    for line in open(path):
        if line.startswith("("):
            take = True   # tools/pmc.sh prints one block per kernel; the driver runs a single kernel kind
            continue
        m = re.match(r"\s+(\w+)\s+([0-9.e+]+)", line)
        if m and take: vals[m.group(1)] = float(m.group(2))
    return vals
enc, dec = counters(out + "/pmc_encode.txt", "k_fwd_fast"), counters(out + "/pmc_decode.txt", "k_inv_fast")
frames = B * C * K
def entry(v, kernel, alg):
    f, w = v["FETCH_SIZE"] * 1024 * 2, v["WRITE_SIZE"] * 1024
    return {"kernel": kernel, "fetch_bytes_corrected": f, "write_bytes": w, "hbm_bytes_per_launch": f + w,
            "algorithmic_bytes_per_launch": alg}
json.dump({"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc.sh), bench workload B=256 stereo "
                   "K=468 N=1024; FETCH_SIZE doubled per the gfx950 correction (MI355X_MICROARCH.md, HBM section); counters "
                   "are in KiB; memory-side requests include Infinity-Cache hits",
           "encode": entry(enc, "k_fwd_fast<8, 0, true, 4, 0, 2>", (12 * N + 4) * frames),
           "decode": entry(dec, "k_inv_fast<8, 0, 4, 0>", 8 * N * frames)}, open(out + "/traffic.json", "w"), indent=1)
print(open(out + "/traffic.json").read())
PY
tools/entry_points.sh > /dev/null 2>&1; cp gpurun_out/entry_points.txt $out/entry_points.txt
ls -la $out
