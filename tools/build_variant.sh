#!/bin/bash
# builds a diagnostic variant of libshodh_hip.so next to the product one: tools/build_variant.sh <suffix> <extra hipcc flags...>
# e.g. tools/build_variant.sh prof -DSHODH_PROF   ->  shodh_memory_amd/libshodh_hip.so.prof   (use with SHODH_HIP_LIB=...; results may be invalid)
# The objects go to a temporary directory that is removed afterwards (55 variant trees = 204 MB once rode every push to the GPU box).
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); SUF=$1; shift
OBJ=$(mktemp -d /tmp/shodh_variant_$SUF.XXXXXX); trap 'rm -rf $OBJ' EXIT
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result $@"
for f in $ROOT/shodh_memory_amd/csrc/*.hip; do /opt/rocm/bin/hipcc $FLAGS -c $f -o $OBJ/$(basename $f .hip).o & done; wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJ/*.o -o $ROOT/shodh_memory_amd/libshodh_hip.so.$SUF -lpthread
echo built $ROOT/shodh_memory_amd/libshodh_hip.so.$SUF
