#!/bin/bash
# Build the library of a given commit (default HEAD) as m17_sdr_amd/libm17gpu_base.so, for same-box A/B runs against
# the working tree's libm17gpu.so (scripts/ab_lib.py):   scripts/build_base.sh [commit]
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
C=${1:-HEAD}
T=$(mktemp -d)
git -C $R archive $C m17_sdr_amd/csrc include | tar -x -C $T
make -C $T/m17_sdr_amd/csrc ../libm17gpu.so > $T/build.log 2>&1 || { tail -20 $T/build.log; exit 1; }
cp $T/m17_sdr_amd/libm17gpu.so $R/m17_sdr_amd/libm17gpu_base.so
rm -rf $T
echo "built m17_sdr_amd/libm17gpu_base.so from $C"
