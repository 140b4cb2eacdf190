#!/bin/bash
# usage (GPU box): tools/spread_profile.sh   -- kernel stats + instruction-mix counters of the three spreading forms
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C audiocodec_amd/csrc || exit 1   # never build under the profiler
out=gpurun_out/spread
mkdir -p $out
for n in 1024 2048; do
  N=$n ROUNDS=1 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$n -- python tools/spread_bench.py > $out/trace_$n.log 2>&1
done
i=0
for s in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
         "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  N=2048 ROUNDS=1 rocprofv3 --pmc $s --output-format csv -d $out/pmc_$i -- python tools/spread_bench.py > $out/pmc_$i.log 2>&1
  i=$((i+1))
done
python - <<'PY'
import csv, collections, glob
out = open('gpurun_out/spread/summary.txt', 'w')
for n in (1024, 2048):
    for f in glob.glob('gpurun_out/spread/trace_%d/*/*kernel_stats.csv' % n):
        print('--- kernel stats, N = %d (rocprofv3 --kernel-trace --stats; ns)' % n, file=out)
        for r in csv.DictReader(open(f)):
            if 'ac::' in r['Name']:
                name = r['Name'].replace('void ac::(anonymous namespace)::', '').split('(')[0]
                print('%-52s calls %4s  avg %10.0f  min %10s  max %10s' % (name, r['Calls'], float(r['AverageNs']), r['MinNs'], r['MaxNs']), file=out)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/spread/pmc_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'ac::' not in k: continue
        k = k.replace('void ac::(anonymous namespace)::', '').split('(')[0]
        agg[(k, r['VGPR_Count'], r['LDS_Block_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
print('--- counters per launch, N = 2048 (rocprofv3 --pmc, separate passes)', file=out)
for k, v in agg.items():
    print('%s  vgpr %s  lds %s' % k, file=out)
    for c, vals in sorted(v.items()): print('   %-32s %.5g (n=%d)' % (c, sum(vals) / len(vals), len(vals)), file=out)
out.close()
print(open('gpurun_out/spread/summary.txt').read())
PY
