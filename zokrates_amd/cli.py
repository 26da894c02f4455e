"""`python -m zokrates_amd.cli` — the CLI-shaped shim around libzkhip: the `setup` and `generate-proof` steps of
/root/reference/zokrates_cli/src/ops/{setup,generate_proof}.rs over tool-neutral inputs (.r1cs / .wtns as written by
`zokrates export-r1cs`-style tooling, zokrates_circom), producing ark-format `proving.key`, and `verification.key` /
`proof.json` in ZoKrates' JSON.  Public inputs are listed in wire order (outputs, then public arguments), the order of
the key's gamma_abc.  The inputs may also be ZoKrates' own files — the `out` program of `zokrates compile` and the
`witness` of `zokrates compute-witness` (recognised by their magic bytes) — read by `zkhip_prog_parse` /
`zkhip_prog_assignment` in ark variable order; `inputs` of proof.json is then `public_inputs_values` (public arguments,
then outputs) as in /root/reference/zokrates_ark/src/groth16.rs:33-38.  `-s gm17` selects the GM17 scheme.
Needs a gfx950 GPU (there is no CPU fallback).

    python -m zokrates_amd.cli setup          -i circuit.r1cs|out -p proving.key -v verification.key [-s g16|gm17] [--entropy TEXT]
    python -m zokrates_amd.cli generate-proof -i circuit.r1cs|out -w witness.wtns|witness -p proving.key -j proof.json [-s g16|gm17]
"""
import argparse
import hashlib
import os
import sys

from . import formats, native, rng


def _field_elems(curve_id, seed, count):
    """`count` non-zero Fr elements from a seed (SHAKE-256 stream, rejection sampling) — setup toxic waste.  (The
    reference's setup also samples random group generators from its RNG; a key made here is a valid key, not the key
    `zokrates setup --entropy` would make.  Proof randomness, in contrast, follows the reference: see rng.py.)"""
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


def _load_system(ctx, path):
    """(constraint system on the GPU, zkhip_prog or None) from an .r1cs file or a ZoKrates `out` file."""
    data = open(path, "rb").read()
    if data[:4] == b"ZOK\0":
        prog = native.Program(data, ctx.lib)
        return prog.constraint_system(ctx), prog
    r1 = formats.read_r1cs(data)
    return native.ConstraintSystem(ctx, r1.curve_id, r1.n, r1.l, r1.w, r1.mats), None


def _load_key(ctx, curve_id, path, scheme, cache_dir):
    """`proving.key` -> resident key.  With --key-cache DIR the device-layout image (`zkhip_pk_export`) is kept next to
    it, named after (path, size, mtime, scheme) — hashing a 400 MB key would cost more than parsing it — and later runs
    import the image (`zkhip_pk_import`: no parsing, no Montgomery conversion)."""
    if cache_dir:
        st = os.stat(path)
        tag = hashlib.sha256(f"{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}|{scheme}|{curve_id}".encode()).hexdigest()[:32]
        image_path = os.path.join(cache_dir, tag + ".zkhippk")
        if os.path.exists(image_path):
            try:
                return native.ProvingKey.from_image(ctx, curve_id, open(image_path, "rb").read(), scheme=scheme)
            except native.ZkhipError:
                pass                                   # stale image of another library build: fall through and rewrite it
    pk = native.ProvingKey(ctx, curve_id, open(path, "rb").read(), scheme=scheme)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = image_path + ".tmp%d" % os.getpid()
        with open(tmp, "wb") as f:
            f.write(pk.export_image().tobytes())
        os.replace(tmp, image_path)
    return pk


def cmd_setup(args):
    ctx = native.Context(args.device)
    cs, _ = _load_system(ctx, args.input)
    toxic = _field_elems(cs.curve_id, b"zkhip-setup" + _seed(args.entropy), 5)
    if args.proving_scheme == "gm17":
        toxic[2] = 1                                   # ark's generate_random_parameters: gamma = 1
        pk = native.setup_gm17(ctx, cs, (toxic[0], toxic[1], toxic[2], toxic[4]))
        vk = formats.verification_key_json_gm17(cs.curve_id, pk)
    else:
        pk = native.setup_g16(ctx, cs, toxic)
        vk = formats.verification_key_json(cs.curve_id, pk)
    open(args.proving_key_path, "wb").write(pk.tobytes())
    open(args.verification_key_path, "w").write(vk)
    print(f"setup ({args.proving_scheme}): {cs.n} constraints, {cs.m} variables, {cs.l - 1} public; wrote {args.proving_key_path}, "
          f"{args.verification_key_path}")


def cmd_generate_proof(args):
    ctx = native.Context(args.device)
    cs, prog = _load_system(ctx, args.input)
    wdata = open(args.witness, "rb").read()
    if prog is not None:
        z, inp = prog.assignment(wdata)                # ark order; inputs = public_inputs_values
        inputs = [int.from_bytes(inp[32 * i:32 * i + 32].tobytes(), "little") for i in range(inp.size // 32)]
    else:
        curve_w, z = formats.read_wtns(wdata)
        if curve_w != cs.curve_id or z.size != 32 * cs.m:
            sys.exit("witness does not match the constraint system")
        inputs = [int.from_bytes(z[32 * i:32 * i + 32].tobytes(), "little") for i in range(1, cs.l)]
    pk = _load_key(ctx, cs.curve_id, args.proving_key_path, args.proving_scheme, args.key_cache)
    # the blinding scalars are drawn as the reference draws them: StdRng seeded from --entropy (rng.rs:5-20) or from the OS,
    # then `Fr::rand` twice (Groth16: r, s) or three times (GM17: d1, d2, r) — zokrates_amd/rng.py
    gen = rng.rng_from_entropy(args.entropy) if args.entropy is not None else rng.StdRng(os.urandom(32))
    if args.proving_scheme == "gm17":
        d1, d2, r = (rng.fr_rand(gen, cs.curve_id) for _ in range(3))
        raw = native.prove_gm17(ctx, pk, cs, z, d1, d2, r)
    else:
        r, s = (rng.fr_rand(gen, cs.curve_id) for _ in range(2))
        raw = native.prove_g16(ctx, pk, cs, z, r, s)
    open(args.proof_path, "w").write(formats.proof_json(cs.curve_id, raw, inputs, scheme=args.proving_scheme))
    print(f"generate-proof ({args.proving_scheme}): wrote {args.proof_path}")


def main(argv=None):
    ap = argparse.ArgumentParser(prog="zokrates_amd.cli")
    sub = ap.add_subparsers(dest="cmd", required=True)
    s = sub.add_parser("setup")
    s.add_argument("-i", "--input", required=True)
    s.add_argument("-p", "--proving-key-path", default="proving.key")
    s.add_argument("-v", "--verification-key-path", default="verification.key")
    s.add_argument("-s", "--proving-scheme", default="g16", choices=["g16", "gm17"])
    s.add_argument("--entropy")
    s.add_argument("--device", type=int, default=0)
    s.set_defaults(fn=cmd_setup)
    g = sub.add_parser("generate-proof")
    g.add_argument("-i", "--input", required=True)
    g.add_argument("-w", "--witness", required=True)
    g.add_argument("-p", "--proving-key-path", default="proving.key")
    g.add_argument("-j", "--proof-path", default="proof.json")
    g.add_argument("-s", "--proving-scheme", default="g16", choices=["g16", "gm17"])
    g.add_argument("--key-cache", help="directory for device-layout key images (zkhip_pk_export / zkhip_pk_import)")
    g.add_argument("--entropy")
    g.add_argument("--device", type=int, default=0)
    g.set_defaults(fn=cmd_generate_proof)
    args = ap.parse_args(argv)
    args.fn(args)


if __name__ == "__main__":
    main()
