#!/bin/bash
# Static instruction mix of the hot loop of k_msm_accum<G1> (no GPU needed): device-only assembly of csrc/bn254_g1.hip, the
# kernel's loop, and in it the path every wavefront runs for a sorted entry: the loop is cut at its WAVE-UNIFORM branch (the
# vote "does any lane have an infinite base / an equal-x case?") and the general side — which runs practically never on
# full-width scalars — is left out.  (The few blocks of the bucket-boundary handling, which only the lanes at a boundary
# execute but which are issued for the wavefront at nearly every step, ARE counted: they are part of the price of a step.)
# Usage: bash tools/isa_mix.sh [out.txt]      ISA_FLAGS="-D..." for variants
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-variable --cuda-device-only -S \
  ${ISA_FLAGS:-} "$root/zokrates_amd/csrc/bn254_g1.hip" -o "$tmp/g1.s" 2>/dev/null
python3 - "$tmp/g1.s" ${ISA_VARIANT:-dense} <<'PY' | tee "${1:-/dev/stdout}"
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
variant = "1" if sys.argv[2:] and sys.argv[2] == "sparse" else "0"      # SKIP_INF: ISA_VARIANT=sparse picks the kernel of tables with many points at infinity
start = next(i for i, l in enumerate(src) if re.match(r"^_ZN2zk11k_msm_accumINS_2FuINS_7Bn254FqEEELi\d+ELb" + variant + r"E.*:", l))
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
body = src[start:end]
regs = [l.strip() for l in src[end:end + 60] if "next_free_vgpr" in l or "private_segment_fixed" in l]
labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
inloop = [i for i in labels if "Loop" in body[i]]
first, last = min(inloop), max(inloop)
exit_ = min([i for i in labels if i > last] + [len(body)])
loop = body[first:exit_]
idx = {m.group(1): i for i, l in enumerate(loop) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
cut = None
for i, l in enumerate(loop):      # the wave-uniform branch: a scalar conditional branch over a long stretch of the loop
    m = re.match(r"\s+s_cbranch_(vccz|vccnz|scc0|scc1)\s+(\.LBB\d+_\d+)", l)
    if m and m.group(2) in idx and idx[m.group(2)] - i > 800:
        cut = (i, idx[m.group(2)])
        break
if cut:
    a, b = loop[cut[0] + 1:cut[1]], loop[cut[1]:]
    general, fast = (a, b) if len(a) > len(b) else (b, a)      # the general side holds the inlined doubling: the longer one
    hot = loop[:cut[0] + 1] + fast
    note = f"{len(general)} lines of the general path left out"
else:
    hot, note = loop, "no wave-uniform branch found: the whole loop"
mix = collections.Counter()
for l in hot:
    t = l.split(";")[0].split()
    if not t or t[0].startswith(".") or t[0].endswith(":"): continue
    mix[t[0]] += 1
total = sum(mix.values())
print(f"k_msm_accum<Fu<Bn254Fq>, SKIP_INF = {bool(sys.argv[2:] and sys.argv[2] == 'sparse')}>: {total} instructions per sorted entry on the hot path ({note}); {', '.join(regs)}")
for k, v in mix.most_common(18):
    print(f"  {v:5d}  {100.0 * v / total:5.1f} %  {k}")
PY
rm -rf "$tmp"
