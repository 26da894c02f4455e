#!/usr/bin/env python3
"""Device memory in use at each stage of a large-domain Groth16 run (rocm-smi --showmeminfo vram), to attribute an
out-of-resources abort: python tools/gpu_mem_probe.py [log2 n] [batch]"""
import json
import os
import subprocess
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zokrates_amd import native, synth  # noqa: E402


def used(tag):
    try:
        txt = subprocess.run(["rocm-smi", "--showmeminfo", "vram", "--json"], capture_output=True, text=True, timeout=30).stdout
        d = next(iter(json.loads(txt[txt.index("{"):]).values()))
        u = [int(v) for k, v in d.items() if "Used" in k][0]
        print("%-34s %8.2f GiB used" % (tag, u / 2**30), flush=True)
    except Exception as e:
        print(tag, "rocm-smi failed:", e, flush=True)


lg = int(sys.argv[1]) if len(sys.argv) > 1 else 22
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 3
used("start")
ctx = native.Context(0)
used("context")
circ = synth.circuit(0, n=1 << lg)
cs = native.ConstraintSystem(ctx, 0, circ.n, circ.l, circ.w, circ.mats())
used("constraint system")
raw = native.setup_g16(ctx, cs, synth.toxic_waste(0))
used("setup")
pk = native.ProvingKey(ctx, 0, raw)
used("key resident")
z = circ.assignment(0x5EED0001)
za = native.Assignment(ctx, cs, z)
t0 = time.time()
p1 = native.prove_g16_resident(ctx, pk, cs, za, 11, 13)
used("one proof (%.0f ms)" % (1000 * (time.time() - t0)))
for k in range(2):
    t0 = time.time()
    proofs, _ = native.prove_g16_resident_batch(ctx, pk, cs, [za] * batch, [(11, 13)] * batch)
    used("batch of %d (%.0f ms)" % (batch, 1000 * (time.time() - t0)))
    assert all(p == p1 for p in proofs)
print("ok")
