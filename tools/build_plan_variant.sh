#!/bin/bash
# usage: tools/build_plan_variant.sh NAME 'AC_WAVE_CT(800, 64, 10, 10, 4, 0) AC_WAVE_CT(...)'
#   -> audiocodec_amd/lib/variants/libaudiocodec_amd_NAME.so with ONLY those LDS-FFT instances (for tools/plan_ab.py)
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/audiocodec_amd/lib/variants"
h="$root/audiocodec_amd/lib/variants/sizes_$1.h"
echo "#define AC_WAVE_CT_SIZES $2" > "$h"
bash "$root/tools/build_variant.sh" "$1" -include "$h"
