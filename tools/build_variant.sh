#!/bin/bash
# Builds an experiment variant of libsiftmi.so into tools/tmp_variants/ (git-ignored; travels to the GPU box with gpurun).
# usage: tools/build_variant.sh <name> [-DFLAG=V ...]     ->  tools/tmp_variants/libsiftmi_<name>.so   (use with SIFTMI_LIB=...)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $R/tools/tmp_variants
cd $R/siftmetal_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-slp-vectorize -Wall -Wno-unused-function "$@" -shared -o $R/tools/tmp_variants/libsiftmi_$NAME.so siftmi_api.hip
echo built $R/tools/tmp_variants/libsiftmi_$NAME.so
