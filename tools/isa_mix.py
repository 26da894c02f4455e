#!/usr/bin/env python3
"""Static instruction mix of the accumulation loop of k_msm_accum (no GPU needed).

    tools/isa_mix.py g1|g2 [--curve bn254|bls381] [--sparse] [--asm FILE] [-D...]

Compiles csrc/bn254_<group>.hip to device assembly (or reads --asm), finds the kernel's loop and cuts it at its
WAVE-UNIFORM branch (the vote "does any lane have an infinite base / an equal-x case?"): the side every wavefront runs for a
sorted entry is the HOT path, the other side — the general addition with its inlined doubling, practically never taken on
full-width scalars — is reported separately.  For both it prints the instruction count, the multiply-adds, and the accesses
to scratch memory (scratch_* / buffer_* with `offen`/`off` to the private segment): the kernel's `scratch_size` is a
per-function figure, this says which path it belongs to."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def asm_for(group, flags, curve="bn254"):
    tmp = tempfile.mkdtemp()
    out = os.path.join(tmp, group + ".s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-unused-function", "-Wno-unused-variable",
                           "--cuda-device-only", "-S"] + flags + [os.path.join(ROOT, "zokrates_amd", "csrc", "%s_%s.hip" % (curve, group)), "-o", out],
                          stderr=subprocess.DEVNULL)
    return out


def is_scratch(op, line):
    return op.startswith("scratch_") or (op.startswith("buffer_") and " off" in line or op.startswith("buffer_") and "offen" in line)


def mix_of(lines):
    mix = collections.Counter()
    scratch = collections.Counter()
    for l in lines:
        t = l.split(";")[0].split()
        if not t or t[0].startswith(".") or t[0].endswith(":"):
            continue
        mix[t[0]] += 1
        if is_scratch(t[0], l):
            scratch["load" if "load" in t[0] else "store"] += 1
    return mix, scratch


def main():
    args = sys.argv[1:]
    group = args.pop(0) if args and args[0] in ("g1", "g2") else "g1"
    sparse = "--sparse" in args
    asm = None
    if "--asm" in args:
        asm = args[args.index("--asm") + 1]
    flags = [a for a in args if a.startswith("-D")]
    curve = args[args.index("--curve") + 1] if "--curve" in args else "bn254"
    path = asm or asm_for(group, flags, curve)
    src = open(path).read().split("\n")
    fq = "7Bn254Fq" if curve == "bn254" else "8Bls381Fq"
    ftype = ("2FuINS_%sEEE" if group == "g1" else "3Fu2INS_%sEEE") % fq
    pat = re.compile(r"^_ZN2zk11k_msm_accumINS_" + ftype + r"Li\d+ELb" + ("1" if sparse else "0") + r"E.*:")
    start = next(i for i, l in enumerate(src) if pat.match(l))
    end = next(i for i in range(start, len(src)) if ".amdhsa_kernel" in src[i])
    body = src[start:end]
    regs = [l.strip() for l in src[end:end + 80] if "next_free_vgpr" in l or "private_segment_fixed" in l or "accum_offset" in l]
    labels = [i for i, l in enumerate(body) if re.match(r"^\.LBB\d+_\d+:", l)]
    inloop = [i for i in labels if "Loop" in body[i]]
    first, last = min(inloop), max(inloop)
    exit_ = min([i for i in labels if i > last] + [len(body)])
    loop = body[first:exit_]
    idx = {m.group(1): i for i, l in enumerate(loop) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
    cut = None
    for i, l in enumerate(loop):      # the wave-uniform branch: a scalar conditional branch over a long stretch of the loop
        m = re.match(r"\s+s_cbranch_(vccz|vccnz|scc0|scc1)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(2) in idx and (idx[m.group(2)] - i > 800 or any("s_swappc_b64" in x for x in loop[i:idx[m.group(2)]])):
            cut = (i, idx[m.group(2)])      # (the general path inlined: a long stretch; out of line: the stretch with the call)
            break
    if cut:
        a, b = loop[cut[0] + 1:cut[1]], loop[cut[1]:]
        if any("s_swappc_b64" in x for x in a + b):
            general, fast = (a, b) if any("s_swappc_b64" in x for x in a) else (b, a)
        else:
            general, fast = (a, b) if len(a) > len(b) else (b, a)      # the general side holds the inlined doubling: the longer one
        hot = loop[:cut[0] + 1] + fast
    else:
        hot, general = loop, []
    outside = body[:first] + body[exit_:]
    name = "k_msm_accum<%s<%s>, SKIP_INF = %s>" % ("Fu" if group == "g1" else "Fu2", "Bn254Fq" if curve == "bn254" else "Bls381Fq", sparse)
    print("%s  %s  [%s]" % (name, " ".join(flags), ", ".join(regs)))
    for title, part in (("HOT path (every sorted entry)", hot), ("general path (vote taken: infinite base, doubling, cancellation)", general),
                        ("outside the loop (prologue, last store)", outside)):
        mix, scratch = mix_of(part)
        total = sum(mix.values())
        mads = mix.get("v_mad_u64_u32", 0)
        print("  %-68s %6d instructions, %5d v_mad_u64_u32, scratch loads %d stores %d" % (title + ":", total, mads, scratch["load"], scratch["store"]))
        if part is hot:
            for k, v in mix.most_common(int(os.environ.get("ISA_MIX_TOP", "14"))):
                print("      %5d  %5.1f %%  %s" % (v, 100.0 * v / max(total, 1), k))


if __name__ == "__main__":
    main()
