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
    python -m zokrates_amd.cli verify         -v verification.key -j proof.json        (the compiled verifier: no GPU needed)
    python -m zokrates_amd.cli print-proof    -j proof.json -f remix|json
"""
import argparse
import hashlib
import json
import os
import sys
import threading
import time

import numpy as np

from . import formats, native, rng


def _toxic_waste(curve_id, entropy):
    """(alpha, beta, gamma, delta, tau) for setup: five `Fr::rand` draws from the RNG the reference would use — `StdRng` seeded
    from --entropy (rng.rs:5-20) or from the OS —, redrawn while zero.  ark's `generate_random_parameters` starts with the same
    four draws (alpha, beta, gamma, delta) but then samples random group generators from the RNG, which this setup replaces by
    the standard generators: a key made here is a valid key, not the key `zokrates setup --entropy` would make.  The same
    recipe, draw for draw, is in the compiled host layer (csrc/host/backend.cpp): both CLIs write the same key."""
    gen = rng.rng_from_entropy(entropy) if entropy is not None else rng.StdRng(os.urandom(32))
    out = []
    while len(out) < 5:
        v = rng.fr_rand(gen, curve_id)
        if v:
            out.append(v)
    return out


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
    import the image (`zkhip_pk_import`: no parsing, no Montgomery conversion; the window multiples are recomputed on the device)."""
    if cache_dir:
        st = os.stat(path)
        tag = hashlib.sha256(f"{os.path.abspath(path)}|{st.st_size}|{st.st_mtime_ns}|{scheme}|{curve_id}".encode()).hexdigest()[:32]
        image_path = os.path.join(cache_dir, tag + ".zkhippk")
        if os.path.exists(image_path):
            try:
                return native.ProvingKey.from_image(ctx, curve_id, np.fromfile(image_path, dtype=np.uint8), scheme=scheme), "image"
            except native.ZkhipError:
                pass                                   # stale image of another library build: fall through and rewrite it
    pk = native.ProvingKey(ctx, curve_id, np.fromfile(path, dtype=np.uint8), scheme=scheme)
    if cache_dir:
        os.makedirs(cache_dir, exist_ok=True)
        tmp = image_path + ".tmp%d" % os.getpid()
        pk.export_image().tofile(tmp)
        os.replace(tmp, image_path)
    return pk, "proving.key"


def _curve_of(path):
    """Curve id from the first bytes of a program file (`out`: Field::id at offset 8; .r1cs: the prime of its header)."""
    with open(path, "rb") as f:
        head = f.read(4096)
    if head[:4] == b"ZOK\0":
        cid = {bytes.fromhex("b4f7b5bd"): 0, bytes.fromhex("40d8c1f9"): 1}.get(head[8:12])
        if cid is None:
            sys.exit("unknown curve identifier in the program file")
        return cid
    return None


def cmd_setup(args):
    ctx = native.Context(args.device)
    cs, _ = _load_system(ctx, args.input)
    toxic = _toxic_waste(cs.curve_id, args.entropy)
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
    """`zokrates generate-proof` (/root/reference/zokrates_cli/src/ops/generate_proof.rs:152-202): program + witness +
    proving key -> proof.json, one proof per process.  The three inputs are independent until the proof starts, so the
    program is read and decoded on host threads (zkhip_prog_parse: the constraint section in parallel chunks) WHILE the key
    is uploaded and its window multiples are built on the GPU; --timings prints where the wall clock went."""
    t_start = time.perf_counter()
    marks = {}

    def lap(name, t0):
        marks[name] = round(1000.0 * (time.perf_counter() - t0), 3)

    curve_hint = _curve_of(args.input)
    native.default_library()                           # loaded once, before the host-side thread needs it too
    host = {}

    def load_host_side():                              # no GPU involved: runs beside the key upload
        t0 = time.perf_counter()
        data = np.fromfile(args.input, dtype=np.uint8)
        wdata = np.fromfile(args.witness, dtype=np.uint8)
        lap("read_program_and_witness_ms", t0)
        t0 = time.perf_counter()
        if data[:4].tobytes() == b"ZOK\0":
            prog = native.Program(data)
            host["prog"] = prog
            lap("parse_program_ms", t0)
            t0 = time.perf_counter()
            host["z"], inp = prog.assignment(wdata)    # ark order; inputs = public_inputs_values
            host["inputs"] = [int.from_bytes(inp[32 * i:32 * i + 32].tobytes(), "little") for i in range(inp.size // 32)]
            lap("witness_to_assignment_ms", t0)
        else:
            host["r1cs"] = formats.read_r1cs(data.tobytes())
            lap("parse_program_ms", t0)
            t0 = time.perf_counter()
            host["wtns"] = formats.read_wtns(wdata.tobytes())
            lap("witness_to_assignment_ms", t0)

    def load_host_side_guarded():                      # an error in the worker is the run's error: kept, raised by the main thread
        try:
            load_host_side()
        except BaseException as e:                     # noqa: B902 — whatever it is, the main thread must see it
            host["error"] = e

    worker = None
    if curve_hint is not None:                         # the key's curve is known from the header: overlap
        worker = threading.Thread(target=load_host_side_guarded)
        worker.start()
    t0 = time.perf_counter()
    ctx = native.Context(args.device)
    ctx.tune("msm_sets", 64)        # one proof, then the process ends: the window-multiple tables cost ten times what they save it
    ctx.tune("serial", 1)           # ... and one stream: the streams a resident prover overlaps proofs on cost ~10 ms of set-up each
    lap("hip_init_ms", t0)
    if worker is None:
        load_host_side()
        curve_hint = host["r1cs"].curve_id
    t0 = time.perf_counter()
    pk, key_source = _load_key(ctx, curve_hint, args.proving_key_path, args.proving_scheme, args.key_cache)
    lap("key_load_ms", t0)
    if worker is not None:
        t0 = time.perf_counter()
        worker.join()
        lap("wait_for_host_side_ms", t0)               # what the host side took beyond the key upload
        if "error" in host:                            # a bad program / witness file: its own message, not a KeyError further down
            raise host["error"]
    t0 = time.perf_counter()
    if "prog" in host:
        cs = host["prog"].constraint_system(ctx)
        z, inputs = host["z"], host["inputs"]
    else:
        r1 = host["r1cs"]
        cs = native.ConstraintSystem(ctx, r1.curve_id, r1.n, r1.l, r1.w, r1.mats)
        curve_w, z = host["wtns"]
        if curve_w != cs.curve_id or z.size != 32 * cs.m:
            sys.exit("witness does not match the constraint system")
        inputs = [int.from_bytes(z[32 * i:32 * i + 32].tobytes(), "little") for i in range(1, cs.l)]
    lap("r1cs_upload_ms", t0)
    # the blinding scalars are drawn as the reference draws them: StdRng seeded from --entropy (rng.rs:5-20) or from the OS,
    # then `Fr::rand` twice (Groth16: r, s) or three times (GM17: d1, d2, r) — zokrates_amd/rng.py
    t0 = time.perf_counter()
    gen = rng.rng_from_entropy(args.entropy) if args.entropy is not None else rng.StdRng(os.urandom(32))
    if args.proving_scheme == "gm17":
        d1, d2, r = (rng.fr_rand(gen, cs.curve_id) for _ in range(3))
        raw = native.prove_gm17(ctx, pk, cs, z, d1, d2, r)
    else:
        r, s = (rng.fr_rand(gen, cs.curve_id) for _ in range(2))
        raw = native.prove_g16(ctx, pk, cs, z, r, s)
    lap("prove_ms", t0)
    t0 = time.perf_counter()
    with open(args.proof_path, "w") as f:
        f.write(formats.proof_json(cs.curve_id, raw, inputs, scheme=args.proving_scheme))
    lap("proof_json_ms", t0)
    marks["total_in_process_ms"] = round(1000.0 * (time.perf_counter() - t_start), 3)
    marks["key_source"] = key_source
    marks["constraints"] = cs.n
    print(f"generate-proof ({args.proving_scheme}): wrote {args.proof_path}")
    if args.timings:
        print("timings " + json.dumps(marks))


def cmd_native(args):
    """`verify` / `print-proof`: the pairing check and the proof printer live in the compiled host layer (csrc/host/verify.cpp,
    /root/reference/zokrates_cli/src/ops/{verify,print_proof}.rs); this shim hands the command to the executable next to the
    library and leaves with its exit status."""
    exe = os.environ.get("ZKHIP_CLI") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "zkhip-cli")
    if not os.access(exe, os.X_OK):
        raise SystemExit(f"{exe} is missing: python -m zokrates_amd.build builds it next to libzkhip.so")
    sys.stdout.flush()
    os.execv(exe, [exe, args.cmd] + args.rest)


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if argv and argv[0] in ("verify", "print-proof"):
        return cmd_native(argparse.Namespace(cmd=argv[0], rest=argv[1:]))
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
    g.add_argument("--timings", action="store_true", help="print the split of the wall clock as one JSON object")
    g.add_argument("--entropy")
    g.add_argument("--device", type=int, default=0)
    g.set_defaults(fn=cmd_generate_proof)
    args = ap.parse_args(argv)
    args.fn(args)


if __name__ == "__main__":
    main()
    # one command per process: everything is on disk, so leave without tearing down the tables and the HIP runtime
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(0)
