#!/bin/bash
# a variant of the library that differs from the current build in SOME translation units (same-box A/B of kernel families without
# rebuilding the others):   bash tools/build_variant_units.sh <name> <unit>:"<flags>" [<unit>:"<flags>" ...]
#   e.g.  bash tools/build_variant_units.sh w42 bn254_g1:"-DZK_G1_ACCUM_WPE=4" bn254_g2:"-DZK_G2_ACCUM_WPE=2"
# -> zokrates_amd/libzkhip_<name>.so (git-ignored).  python -m zokrates_amd.build must have run for the current sources: the
# other objects come from zokrates_amd/_obj.
set -eu
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
obj=$root/zokrates_amd/_obj
tmp=$(mktemp -d)
declare -A changed
pids=()
for spec in "$@"; do
  unit=${spec%%:*}; flags=${spec#*:}
  changed[$unit]=1
  # shellcheck disable=SC2086
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-unused-variable $flags -c "$root/zokrates_amd/csrc/$unit.hip" -o "$tmp/$unit.o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait "$p"; done
objs=()
for u in bls381_g2 bls381_g1 bn254_g2 bn254_g1 curve_bn254 curve_bls381 zkhip_api ingest; do
  if [ -n "${changed[$u]:-}" ]; then objs+=("$tmp/$u.o"); else objs+=("$obj/$u.o"); fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$root/zokrates_amd/libzkhip_$name.so" "${objs[@]}"
rm -rf "$tmp"
echo "$root/zokrates_amd/libzkhip_$name.so"
