#!/bin/bash
# usage: tools/asm_one.sh FILE.hip 'AC_WAVE_CT(960, 64, 10, 8, 6, 0)' OUT.s   -- device assembly of one LDS-FFT instance (seconds, not minutes)
cd "$(dirname "$0")/../audiocodec_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fvisibility=hidden \
  --offload-device-only -S "-DAC_WAVE_CT_SIZES=$2" "$1" -o "$3" 2>&1 | grep -v "warning: argument unused"
grep -E "^\s+\.(vgpr_count|private_segment_fixed_size|sgpr_spill_count)|\.name:" "$3" | paste - - - - | sed 's/ \+/ /g'
