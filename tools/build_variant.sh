#!/bin/bash
# usage: tools/build_variant.sh NAME [extra hipcc flags]  -> audiocodec_amd/lib/variants/libaudiocodec_amd_NAME.so
set -e
cd "$(dirname "$0")/../audiocodec_amd/csrc"
name=$1; shift
tmp=../lib/variants/obj_$name; mkdir -p $tmp
make -s OUTDIR=$tmp CXXFLAGS="-O3 -std=c++17 -fPIC -Wno-unused-function -ffp-contract=fast -fvisibility=hidden -fvisibility-inlines-hidden $*" -j4
cp $tmp/libaudiocodec_amd.so ../lib/variants/libaudiocodec_amd_$name.so
rm -rf $tmp
echo built variants/libaudiocodec_amd_$name.so
