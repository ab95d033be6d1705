#!/bin/bash
# tools/build_alt.sh NAME -DMACRO=VALUE ... — an experimental build of the library next to the shipped one
# (highs_amd/lib/alt/lib_NAME.so, for same-box A/B runs through PDLP_MI355X_LIB); the shipped build is restored afterwards.
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
cd $R/highs_amd/csrc
mkdir -p ../lib/alt
BASE="-O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-result"
touch pdlp_kernels.hip pdlp_small.hip pdlp_check.hip pdlp_halpern.hip pdlp_mesh.hip pdlp_setup.hip
make CXXFLAGS="$BASE $*" 2>&1 | grep -E "error|warning" || true
cp ../lib/libpdlp_mi355x.so ../lib/alt/lib_$NAME.so
touch pdlp_kernels.hip pdlp_small.hip pdlp_check.hip pdlp_halpern.hip pdlp_mesh.hip pdlp_setup.hip
make 2>&1 | grep -E "error|warning" || true
ls -la ../lib/alt/lib_$NAME.so ../lib/libpdlp_mi355x.so
