#!/bin/bash
# tools/build_variant.sh NAME SOURCE.hip [-DMACRO=..]...: deepsir_amd/libdsir_NAME.so = the current objects with SOURCE recompiled
# under the extra macros (timing / A-B builds; run after `python -m deepsir_amd.csrc.build`)
set -e
name=$1; src=$2; shift 2
cs=deepsir_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Iinclude -I$cs "$@" -c $cs/$src -o /tmp/variant_$name.o
objs=$(ls $cs/build/*.o | grep -v "/${src%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o deepsir_amd/libdsir_$name.so $objs /tmp/variant_$name.o
echo deepsir_amd/libdsir_$name.so
