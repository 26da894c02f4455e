"""`python -m zokrates_amd.cli` — the CLI-shaped shim around libzkhip: the `setup` and `generate-proof` steps of
/root/reference/zokrates_cli/src/ops/{setup,generate_proof}.rs over tool-neutral inputs (.r1cs / .wtns as written by
`zokrates export-r1cs`-style tooling, zokrates_circom), producing ark-format `proving.key`, and `verification.key` /
`proof.json` in ZoKrates' JSON.  Public inputs are listed in wire order (outputs, then public arguments), the order of
the key's gamma_abc.  Needs a gfx950 GPU (there is no CPU fallback).

    python -m zokrates_amd.cli setup          -i circuit.r1cs -p proving.key -v verification.key [--entropy TEXT]
    python -m zokrates_amd.cli generate-proof -i circuit.r1cs -w witness.wtns -p proving.key -j proof.json [--entropy TEXT]
"""
import argparse
import hashlib
import os
import sys

from . import formats, native


def _field_elems(curve_id, seed, count):
    """`count` non-zero Fr elements from a seed (SHAKE-256 stream, rejection sampling) — setup toxic waste / r, s."""
    p = formats.FR_MODULUS[curve_id]
    stream = hashlib.shake_256(seed).digest(64 * (count + 8))
    out, pos = [], 0
    while len(out) < count:
        v = int.from_bytes(stream[pos:pos + 32], "little") & ((1 << p.bit_length()) - 1)
        pos += 32
        if 0 < v < p:
            out.append(v)
    return out


def _seed(entropy):
    return entropy.encode() if entropy is not None else os.urandom(32)


def cmd_setup(args):
    r1 = formats.read_r1cs(open(args.input, "rb").read())
    ctx = native.Context(args.device)
    cs = native.ConstraintSystem(ctx, r1.curve_id, r1.n, r1.l, r1.w, r1.mats)
    toxic = _field_elems(r1.curve_id, b"zkhip-setup" + _seed(args.entropy), 5)
    pk = native.setup_g16(ctx, cs, toxic)
    open(args.proving_key_path, "wb").write(pk.tobytes())
    open(args.verification_key_path, "w").write(formats.verification_key_json(r1.curve_id, pk))
    print(f"setup: {r1.n} constraints, {r1.n_wires} wires, {r1.l - 1} public; wrote {args.proving_key_path}, {args.verification_key_path}")


def cmd_generate_proof(args):
    r1 = formats.read_r1cs(open(args.input, "rb").read())
    curve_w, z = formats.read_wtns(open(args.witness, "rb").read())
    if curve_w != r1.curve_id or z.size != 32 * r1.n_wires:
        sys.exit("witness does not match the constraint system")
    ctx = native.Context(args.device)
    cs = native.ConstraintSystem(ctx, r1.curve_id, r1.n, r1.l, r1.w, r1.mats)
    pk = native.ProvingKey(ctx, r1.curve_id, open(args.proving_key_path, "rb").read())
    r, s = _field_elems(r1.curve_id, b"zkhip-prove" + _seed(args.entropy), 2)
    raw = native.prove_g16(ctx, pk, cs, z, r, s)
    inputs = [int.from_bytes(z[32 * i:32 * i + 32].tobytes(), "little") for i in range(1, r1.l)]
    open(args.proof_path, "w").write(formats.proof_json(r1.curve_id, raw, inputs))
    print(f"generate-proof: wrote {args.proof_path}")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="zokrates_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("setup")
    s.add_argument("-i", "--input", required=True)
    s.add_argument("-p", "--proving-key-path", default="proving.key")
    s.add_argument("-v", "--verification-key-path", default="verification.key")
    s.add_argument("--entropy")
    s.add_argument("--device", type=int, default=0)
    s.set_defaults(fn=cmd_setup)
    g = sub.add_parser("generate-proof")
    g.add_argument("-i", "--input", required=True)
    g.add_argument("-w", "--witness", required=True)
    g.add_argument("-p", "--proving-key-path", default="proving.key")
    g.add_argument("-j", "--proof-path", default="proof.json")
    g.add_argument("--entropy")
    g.add_argument("--device", type=int, default=0)
    g.set_defaults(fn=cmd_generate_proof)
    args = ap.parse_args(argv)
    args.fn(args)


if __name__ == "__main__":
    main()
