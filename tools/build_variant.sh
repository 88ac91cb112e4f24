#!/bin/bash
# A variant of the library with scan_mfma.hip compiled under extra flags (diagnostics / A-B runs): tools/build_variant.sh <suffix> <flags ...>
# -> shodh_memory_amd/libshodh_hip.so.<suffix> (the other objects are the product build's; run python -m shodh_memory_amd.build first)
set -e
ROOT=$(cd $(dirname $0)/.. && pwd); suf=$1; shift
B=$ROOT/shodh_memory_amd/build; SRC=${VARIANT_SRC:-scan_mfma}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-value -Wno-unused-result "$@" -c $ROOT/shodh_memory_amd/csrc/$SRC.hip -o /tmp/${SRC}_$suf.o
objs=$(ls $B/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/${SRC}_$suf.o -o $ROOT/shodh_memory_amd/libshodh_hip.so.$suf -lpthread
echo built libshodh_hip.so.$suf "$@"
