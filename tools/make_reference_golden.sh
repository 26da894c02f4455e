#!/bin/bash
# The parity kit for whoever has a Rust toolchain (this image has none: SURVEY.md §8c).  Builds golden directories from the REAL
# reference — `zokrates compile / setup / compute-witness / generate-proof --backend ark --entropy ...`
# (/root/reference/zokrates_cli/src/ops/generate_proof.rs:95-202; --entropy makes setup and proof deterministic:
# zokrates_proof_systems/src/rng.rs:5-20) — in the layout tests/test_reference_golden.py reads:
#     tests/golden/reference/<name>/{out, witness, proving.key, verification.key, proof.json, entropy.txt[, scheme.txt]}
# Dropping them in turns "unique for fixed (pk, z, r, s)" into observed byte equality with zokrates_ark:
#     python -m pytest tests/test_reference_golden.py            # emulator, keys up to 8 MiB
#     python -m pytest tests/test_reference_golden.py -m gpu     # on an MI355X
#
# usage: tools/make_reference_golden.sh <path to the zokrates binary> [<stdlib path>] [<output directory>]
#   e.g. (in a checkout of the reference)  cargo build --release -p zokrates_cli
#        tools/make_reference_golden.sh target/release/zokrates zokrates_stdlib/stdlib
# Cases: Groth16 on bn128 and bls12_381, GM17 on bn128 and bls12_381; a program with outputs (`~out_i` become instance variables:
# zokrates_ark/src/lib.rs:52-69), one whose variables are first seen out of order in the constraints (the allocation order of
# generate_constraints), one with a public and a private argument mixed, and the SHA-256 example of BASELINE.json configs[0].
set -euo pipefail
ZOK=$(realpath "${1:?path to the zokrates binary}")
STDLIB=$(realpath "${2:-zokrates_stdlib/stdlib}")
OUT=$(realpath -m "${3:-$(dirname "$0")/../tests/golden/reference}")
export ZOKRATES_STDLIB="$STDLIB"
mkdir -p "$OUT"
work=$(mktemp -d)
trap 'rm -rf "$work"' EXIT

# case <name> <curve> <scheme> <entropy> <arguments...>   (the program text on stdin)
case_() {
  local name=$1 curve=$2 scheme=$3 entropy=$4; shift 4
  local d="$OUT/$name"
  mkdir -p "$d"
  cat > "$work/$name.zok"
  ( cd "$work"
    "$ZOK" compile -i "$name.zok" -o "$d/out" --curve "$curve" --r1cs "$d/out.r1cs" -s "$d/abi.json" --stdlib-path "$STDLIB" > /dev/null
    "$ZOK" setup -i "$d/out" -b ark -s "$scheme" -p "$d/proving.key" -v "$d/verification.key" --entropy "$entropy setup" > /dev/null
    "$ZOK" compute-witness -i "$d/out" -s "$d/abi.json" -o "$d/witness" --circom-witness "$d/out.wtns" -a "$@" > /dev/null
    "$ZOK" generate-proof -i "$d/out" -w "$d/witness" -p "$d/proving.key" -b ark -s "$scheme" --entropy "$entropy" -j "$d/proof.json" > /dev/null
    "$ZOK" verify -b ark -v "$d/verification.key" -j "$d/proof.json" > /dev/null )
  printf '%s\n' "$entropy" > "$d/entropy.txt"
  [ "$scheme" = g16 ] || printf '%s\n' "$scheme" > "$d/scheme.txt"
  rm -f "$d/out.r1cs" "$d/out.wtns" "$d/abi.json"
  echo "wrote $d ($(du -sh "$d" | cut -f1))"
}

case_ factorize_bn128_g16 bn128 g16 "golden vector 1" 7 13 91 <<'ZOK'
def main(private field a, private field b, field n) {
    assert(a * b == n);
    return;
}
ZOK

case_ outputs_bls12_381_g16 bls12_381 g16 "golden vector 2" 3 5 <<'ZOK'
def main(field x, private field y) -> (field, field) {
    field s = x * y + 7;
    field t = s * s * x;
    return (s, t);
}
ZOK

case_ unordered_bn128_g16 bn128 g16 "golden vector 3" 2 9 4 <<'ZOK'
def main(private field a, field b, private field c) -> field {
    field u = c * c;
    field v = a * u;
    field w = b * v + a;
    assert(w != 0);
    return w * u;
}
ZOK

case_ mixed_bn128_gm17 bn128 gm17 "golden vector 4" 11 6 <<'ZOK'
def main(field p, private field s) -> field {
    field mut acc = p;
    for u32 i in 0..6 {
        acc = acc * acc + s;
    }
    return acc;
}
ZOK

# GM17 a second time, on the other curve.  Cases 4 and 6 are THE test of a property of this repository's restatement that only
# ark-gm17 0.3.0's create_proof can confirm: of its three draws (d1, d2, r) the proof depends on d1 and r only
# through r + d1, and not on d2 at all.
case_ chain_bls12_381_gm17 bls12_381 gm17 "golden vector 6" 4 17 <<'ZOK'
def main(private field a, field b) -> field {
    field mut t = a + b;
    for u32 i in 0..4 {
        t = t * t * a + b;
    }
    assert(t != a);
    return t;
}
ZOK

case_ sha256_bn128_g16 bn128 g16 "golden vector 5" 0 0 0 5 <<'ZOK'
import "hashes/sha256/512bitPacked" as sha256packed;

def main(private field a, private field b, private field c, private field d) -> field[2] {
    field[2] h = sha256packed([a, b, c, d]);
    return h;
}
ZOK

echo "done: $(ls "$OUT" | wc -l) golden directories under $OUT"
