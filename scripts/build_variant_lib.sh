#!/bin/bash
# libaps_amd_<name>.so: the shipped objects with ONE object rebuilt from <source> with extra flags -- A/B
# runs on one box through APS_AMD_LIB (same ABI).  Run after the normal build.
#   scripts/build_variant_lib.sh <name> <object to replace, e.g. stft> <source.hip> [extra hipcc flags ...]
set -e
name=$1; obj=$2; src=$3; shift 3
cd "$(dirname "$0")/../aps_amd/csrc"
[ -f "$src" ] || { echo "no such source: $src"; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mcode-object-version=5 -w \
   -Xclang -target-feature -Xclang -packed-fp32-ops -I"$PWD" "$@" -c "$src" -o /tmp/variant_${name}_$obj.o
objs=$(ls _obj/*.o | grep -v "/$obj.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs /tmp/variant_${name}_$obj.o -o libaps_amd_$name.so
ls -la libaps_amd_$name.so
