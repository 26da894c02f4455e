#!/bin/bash
# builds a variant of the library for a same-box A/B (tools/gpu_ab.sh): bash tools/build_variant.sh <name> [-DFLAG=..]...
# -> zokrates_amd/libzkhip_<name>.so (git-ignored; objects in zokrates_amd/_obj_<name>/, removed afterwards)
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/zokrates_amd/_obj_$name
mkdir -p "$obj"
pids=()
for u in bls381_g2 bls381_g1 bn254_g2 bn254_g1 curve_bn254 curve_bls381 zkhip_api ingest; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable "$@" -c "$root/zokrates_amd/csrc/$u.hip" -o "$obj/$u.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/zokrates_amd/libzkhip_$name.so" "$obj"/*.o
rm -rf "$obj"
echo "$root/zokrates_amd/libzkhip_$name.so"
