#!/bin/bash
# usage: tools/sweep_env.sh VAR v1 v2 ...   -- bench.py once per value of an environment tuning hook
var=$1; shift
for v in "$@"; do
  env $var=$v python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['kernels']
        print('%-22s enc %.3f ms (%.0f GB/s)  dec %.3f ms (%.0f GB/s)  step frac %.3f  value %.1f M' % ('$var=$v', k['encode_ms'], k['encode_GBs'], k['decode_ms'], k['decode_GBs'], k['step_frac_of_hbm_peak'], d['value']/1e6))
"
done
