"""Builds libzkhip.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m zokrates_amd.build            # incremental
    python -m zokrates_amd.build --force

The three translation units (two curves + the C ABI) compile in parallel; hipcc cross-compiles for
gfx950 without a GPU.  The result `zokrates_amd/libzkhip.so` is git-ignored but travels to the GPU
box with the repository snapshot.
"""
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_obj")
LIB = os.path.join(HERE, "libzkhip.so")
UNITS = ["bls381_g2", "bls381_g1", "bn254_g2", "bn254_g1", "curve_bn254", "curve_bls381", "zkhip_api", "ingest"]   # slowest first
HEADERS = ["core.cuh", "devrt.h", "ec.cuh", "field.cuh", "fieldu.cuh", "kernels_msm.cuh", "kernels_ntt.cuh", "setup.cuh", "gm17.cuh", "group.cuh", "bind.cuh", "ingest.h"]
ARCH = "gfx950"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable"]


HOST_ONLY = ("ingest.hip", "ingest.h", "emu.h")   # the program / witness readers and the test-only emulator shim: no kernel, no launch


def csrc_hash(root=None, with_host_only=False):
    """Fingerprint of the sources of libzkhip.so's KERNELS and of the host code that launches them (the sources directly under
    csrc/ and the ABI header, names and bytes): what ties offline evidence — the rocprofv3 counter files under profiles/ — to the
    build it was taken from.  bench.py prints a counter figure only next to a build with the same fingerprint.  Left out: csrc/host
    (the compiled host layer) and HOST_ONLY — the file readers and the emulator shim hold no kernel and launch none, so a change
    there cannot move a counter (`with_host_only=True` is the definition the files were stamped under until the readers were
    reworked: tools/adopt_evidence.py --rekey checks a file against it before giving it the present one)."""
    import hashlib
    here = os.path.join(root, "zokrates_amd") if root else HERE
    files = []
    csrc = os.path.join(here, "csrc")      # (not csrc/host: the compiled host layer is not what the counters measured)
    files += [os.path.join(csrc, n) for n in os.listdir(csrc) if n.endswith((".cuh", ".hip", ".h")) and (with_host_only or n not in HOST_ONLY)]
    files.append(os.path.join(here, "..", "include", "zkhip.h"))
    h = hashlib.sha256()
    for f in sorted(files, key=lambda f: os.path.relpath(f, here)):
        h.update(os.path.relpath(f, here).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
        h.update(b"\0")
    return h.hexdigest()[:16]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(unit, force, verbose):
    src = os.path.join(CSRC, unit + ".hip")
    obj = os.path.join(OBJ, unit + ".o")
    deps = [src, os.path.join(HERE, "..", "include", "zkhip.h")] + [os.path.join(CSRC, h) for h in HEADERS]
    if not force and not _newer(obj, deps):
        return obj, False
    cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    t0 = time.time()
    subprocess.check_call(cmd)
    if verbose:
        print("  %s: %.0f s" % (unit, time.time() - t0), flush=True)
    return obj, True


def build_lib(force=False, verbose=False):
    """Compile every HIP translation unit for gfx950 and link libzkhip.so.  Returns the library path."""
    os.makedirs(OBJ, exist_ok=True)
    with ThreadPoolExecutor(max_workers=min(len(UNITS), os.cpu_count() or 4)) as ex:
        results = list(ex.map(lambda u: _compile(u, force, verbose), UNITS))
    objs = [o for o, _ in results]
    if force or any(changed for _, changed in results) or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    build_cli(force, verbose)
    return LIB


CLI = os.path.join(HERE, "zkhip-cli")


def build_cli(force=False, verbose=False):
    """The compiled host side (include/zkhip_backend.hpp, csrc/host/backend.cpp) and its `generate-proof` executable: plain C++17
    over the C ABI, linked against the in-tree libzkhip.so (found next to the executable through $ORIGIN)."""
    srcs = [os.path.join(CSRC, "host", f) for f in ("backend.cpp", "verify.cpp", "cli_main.cpp")]
    deps = srcs + [os.path.join(HERE, "..", "include", h) for h in ("zkhip.h", "zkhip_backend.hpp")] + [os.path.join(CSRC, h) for h in ("field.cuh", "fieldu.cuh", "ec.cuh")] + [LIB]
    if not force and not _newer(CLI, deps):
        return CLI
    cmd = ["g++", "-O2", "-std=c++17", "-Wall", "-pthread"] + srcs + ["-L" + HERE, "-lzkhip", "-Wl,-rpath,$ORIGIN", "-Wl,--allow-shlib-undefined", "-o", CLI]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return CLI


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
