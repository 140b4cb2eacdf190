#!/bin/bash
# usage: tools/pmc2.sh <encode|inverse> <tag>: memory-pipeline back-pressure counters (run on the GPU box)
what=$1; tag=$2
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
make -s -C audiocodec_amd/csrc || exit 1   # never build under the profiler (the import-time fallback refuses to)
sets=("SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_LEVEL_VMEM" \
      "TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
      "TCC_EA0_WRREQ_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum" "TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum" \
      "TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RDREQ_sum" "TCC_EA0_WRREQ_DRAM_sum TCC_EA0_WRREQ_sum" "TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE" "TCC_BUSY_sum TCC_TAG_STALL_sum")
i=0
for s in "${sets[@]}"; do
  rocprofv3 --pmc $s --output-format csv -d gpurun_out/pmc2_${tag}_$i -- python tools/run_one.py $what 2 > gpurun_out/pmc2_${tag}_$i.log 2>&1
  i=$((i+1))
done
python - <<PY
import csv, collections, glob
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob('gpurun_out/pmc2_${tag}_*/*/*_counter_collection.csv')):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name']
        if 'ac::' not in k: continue
        k=k.split('(')[0].replace('void ac::(anonymous namespace)::','')[:40]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
out=open('gpurun_out/pmc2_${tag}.txt','w')
for k,v in agg.items():
    print(k, file=out)
    for c,vals in sorted(v.items()): print('   %-40s %.5g (n=%d)'%(c, sum(vals)/len(vals), len(vals)), file=out)
out.close()
print(open('gpurun_out/pmc2_${tag}.txt').read())
PY
