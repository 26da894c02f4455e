#!/bin/bash
# a variant of the library that differs from the current build in ONE translation unit (same-box A/B of a kernel family without
# rebuilding the other seven): bash tools/build_variant_unit.sh <name> <unit> [-DFLAG=..]...   -> zokrates_amd/libzkhip_<name>.so
# (python -m zokrates_amd.build must have run for the current sources: the other objects come from zokrates_amd/_obj)
set -eu
name=$1; unit=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/zokrates_amd/_obj
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c "$root/zokrates_amd/csrc/$unit.hip" -o "$tmp/$unit.o"
objs=()
for u in bls381_g2 bls381_g1 bn254_g2 bn254_g1 curve_bn254 curve_bls381 zkhip_api ingest; do
  if [ "$u" = "$unit" ]; then objs+=("$tmp/$u.o"); else objs+=("$obj/$u.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/zokrates_amd/libzkhip_$name.so" "${objs[@]}"
rm -rf "$tmp"
echo "$root/zokrates_amd/libzkhip_$name.so"
