#!/bin/sh
# TEST-ONLY: builds the fibre-emulated copy of libzkhip (same kernel sources, g++, -DZK_EMU) so that
# kernel indexing / LDS / barrier logic runs under `pytest -m "not gpu"`.  Never shipped, never a fallback.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
SRC="$HERE/../../zokrates_amd/csrc"
FLAGS="-U_FORTIFY_SOURCE -O2 -g -std=c++17 -fPIC -DZK_EMU ${EMU_FLAGS:-} -x c++ -Wall -Wno-unused-function -Wno-unknown-pragmas -Wno-unused-variable -Wno-restrict"
mkdir -p "$HERE/obj"
for f in curve_bn254 curve_bls381 bn254_g1 bn254_g2 bls381_g1 bls381_g2 zkhip_api ingest; do
  g++ $FLAGS -c "$SRC/$f.hip" -o "$HERE/obj/$f.o" &
done
wait
g++ -shared -o "$HERE/libzkhip_emu.so" "$HERE"/obj/*.o
# the compiled host side (csrc/host) against the emulator library: the same executable the product ships, for CPU tests
g++ -O2 -std=c++17 -Wall -pthread "$SRC/host/backend.cpp" "$SRC/host/verify.cpp" "$SRC/host/cli_main.cpp" -L"$HERE" -lzkhip_emu -Wl,-rpath,'$ORIGIN' -o "$HERE/zkhip-cli-emu"
