#!/bin/bash
# Static instruction mix of the hot loop of k_msm_accum<G1> (no GPU needed): device-only assembly of csrc/bn254_g1.hip, the
# kernel's body, the loop blocks that run for every sorted entry (the rarely taken doubling branch is left out).
# Usage: bash tools/isa_mix.sh [out.txt]
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-function -Wno-unused-variable --cuda-device-only -S \
  ${ISA_FLAGS:-} "$root/zokrates_amd/csrc/bn254_g1.hip" -o "$tmp/g1.s" 2>/dev/null
python3 - "$tmp/g1.s" <<'PY' | tee "${1:-/dev/stdout}"
import re, sys, collections
src = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(src) if re.match(r"^_ZN2zk11k_msm_accumINS_2FuINS_7Bn254FqEEELi\d+E.*:", l))
end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
body = src[start:end]
# blocks: the loop's labels; a block between two s_swappc calls or holding one is the doubling / slow zero test (cold)
labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
inloop = [i for i in labels if "Loop" in body[i]]
first, last = min(inloop), max(inloop)
exit_ = min([i for i in labels if i > last] + [len(body)])
loop = body[first:exit_]
calls = [i for i, l in enumerate(loop) if "s_swappc" in l]
cold = set()
if calls:      # from the first call to the label after the last big block that follows the calls (the inlined doubling)
    lo = max(i for i, l in enumerate(loop[:calls[0]]) if "s_cbranch_execz" in l)
    nxt = [i for i, l in enumerate(loop) if re.match(r"^\.LBB", l) and i > calls[-1]]
    # the doubling body is the longest label-free run after the last call
    runs, prev = [], calls[-1]
    for i in nxt:
        runs.append((i - prev, prev, i)); prev = i
    big = max(runs)
    cold = set(range(lo, big[2]))
mix = collections.Counter()
for i, l in enumerate(loop):
    if i in cold: continue
    t = l.split(";")[0].split()
    if not t or t[0].startswith(".") or t[0].endswith(":"): continue
    mix[t[0]] += 1
total = sum(mix.values())
print(f"k_msm_accum<Fu<Bn254Fq>>: {total} instructions per sorted entry on the hot path ({len(cold)} lines of cold blocks left out)")
for k, v in mix.most_common(18):
    print(f"  {v:5d}  {100.0 * v / total:5.1f} %  {k}")
PY
rm -rf "$tmp"
