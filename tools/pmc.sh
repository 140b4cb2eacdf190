#!/bin/bash
# usage: tools/pmc.sh <encode|transform|inverse|psy> <tag>   (run on the GPU box; writes gpurun_out/pmc_<tag>.txt)
what=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C audiocodec_amd/csrc || exit 1   # never build under the profiler (the import-time fallback refuses to)
sets=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY" \
      "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU" \
      "FETCH_SIZE" "WRITE_SIZE")
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d gpurun_out/pmc_${tag}_$i -- python tools/run_one.py $what 2 > gpurun_out/pmc_${tag}_$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
out=open('gpurun_out/pmc_${tag}.txt','w')
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc_${tag}_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'ac::' not in k: continue
        k=k.replace('void ac::(anonymous namespace)::','').split('(')[0][:48]
        agg[(k, r['VGPR_Count'], r['LDS_Block_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in agg.items():
    print(k, file=out)
    for c,vals in sorted(v.items()): print('   %-24s %.5g (n=%d)'%(c, sum(vals)/len(vals), len(vals)), file=out)
out.close()
print(open('gpurun_out/pmc_${tag}.txt').read())
PY
