#!/usr/bin/env python3
"""Development probe (not bench.py): proves synthetic circuits of growing size on the GPU, checks each proof
against the oracle's closed-form trapdoor proof and prints phase timings.  Uses the oracle for setup."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from oracle import cpu
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native


def main():
    logs = [int(x) for x in (sys.argv[1:] or ["14", "16", "18"])]
    ctx = native.Context(0)
    print(ctx.describe(), "host threads", cpu.hw_threads(), flush=True)
    for curve in (BN254,):
        for lg in logs:
            t = time.time(); oc = cpu.Circuit.synth(curve.curve_id, (1 << lg) - 2, 0x5EED0000 + lg); t_syn = time.time() - t
            tox = cpu.toxic_bytes(g16.Toxic.from_seed(curve))
            t = time.time(); opk = cpu.ProvingKey.setup(oc, tox); t_setup = time.time() - t
            raw = opk.serialize()
            z = oc.assignment()
            t = time.time(); cs = native.ConstraintSystem(ctx, curve.curve_id, oc.n, oc.l, oc.w, [oc.csr(k) for k in range(3)]); t_cs = time.time() - t
            t = time.time(); pk = native.ProvingKey(ctx, curve.curve_id, raw); t_pk = time.time() - t
            want = cpu.trapdoor(oc, tox, z, 12345, 67890)
            best = None
            for it in range(4):
                t = time.time(); got, tm = native.prove_g16(ctx, pk, cs, z, 12345, 67890, want_timings=True); wall = time.time() - t
                assert got == want, "PARITY FAILURE"
                if best is None or tm["total_ms"] < best["total_ms"]:
                    best = tm
            print(f"{curve.name} 2^{lg}: synth {t_syn:.2f}s cpu-setup {t_setup:.2f}s r1cs_load {t_cs:.2f}s pk_load {t_pk:.2f}s | parity OK | "
                  + " ".join(f"{k}={v:.2f}" for k, v in best.items()), flush=True)
            if lg <= 18:
                t = time.time(); _, ctm = cpu.prove(oc, opk, z, 12345, 67890); print(f"   cpu oracle prove {time.time()-t:.2f}s", {k: round(v, 2) for k, v in ctm.items()}, flush=True)
            pk.close(); cs.close()


if __name__ == "__main__":
    main()
