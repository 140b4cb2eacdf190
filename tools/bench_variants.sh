#!/bin/bash
# run bench.py once per library variant (audiocodec_amd/lib/variants/*.so) and print encode/decode ms
for so in audiocodec_amd/lib/variants/libaudiocodec_amd_*.so; do
  name=$(basename $so .so | sed 's/libaudiocodec_amd_//')
  AUDIOCODEC_AMD_LIB=$PWD/$so python bench.py --steps ${STEPS:-10} --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys
for line in sys.stdin:
    if line.startswith('{'):
        d=json.loads(line); k=d['kernels']
        print('%-14s enc %.3f ms (%.0f GB/s)  dec %.3f ms (%.0f GB/s)  step frac %.3f  value %.1f M' % ('$name', k['encode_ms'], k['encode_GBs'], k['decode_ms'], k['decode_GBs'], k['step_frac_of_hbm_peak'], d['value']/1e6))
"
done
