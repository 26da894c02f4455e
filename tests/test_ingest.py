"""`zkhip_prog_parse` / `zkhip_prog_assignment` (host-only C ABI, rows a2, a3, a5, a6 of SURVEY.md §8a) against the python
restatement in oracle/ir.py: the `out` program file -> R1CS in ark variable order, the `witness` file -> z and the
public inputs.  Runs on the TEST-ONLY emulator build for the CPU suite (the functions are pure host code, identical in
libzkhip.so); the `-m gpu` test proves from the two files end to end.

Reference: /root/reference/zokrates_ast/src/ir/serialize.rs:133-390, /root/reference/zokrates_ark/src/lib.rs:41-141,
/root/reference/zokrates_ast/src/ir/witness.rs:44-71."""
import random
import struct

import numpy as np
import pytest

from oracle import formats, ir
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

CURVES = [BN254, BLS12_381]


@pytest.fixture(scope="module")
def lib():
    from emu_util import emu_library
    return emu_library()


def rows_of(mats, n):
    out = []
    for rp, col, val in mats:
        rows = []
        for i in range(n):
            a, b = int(rp[i]), int(rp[i + 1])
            cols = [int(c) for c in col[a:b]]
            assert cols == sorted(cols) and len(set(cols)) == len(cols)          # ark keeps a row sorted, no duplicates
            rows.append({c: int.from_bytes(val[32 * q:32 * q + 32].tobytes(), "little") for q, c in zip(range(a, b), cols)})
        out.append(rows)
    return tuple(out)


def random_prog(curve, rnd, n, n_args=4, n_out=2, with_noise=True, wide=False):
    """A hand-built (non-canonical) program: duplicate variables inside a combination, zero coefficients, terms that
    cancel, outputs first seen late and out of index order, private/public/unused arguments, directives and logs with
    nested payloads between the constraints, spans and error annotations."""
    r = curve.r
    args = [ir.Parameter(id=k + 1, private=rnd.random() < 0.5) for k in range(n_args)]
    pool = [0] + [p.id for p in args] + [-(k + 1) for k in reversed(range(n_out))] + list(range(n_args + 1, n_args + 1 + 2 * n))
    coeff = lambda: rnd.choice([0, 1, r - 1, rnd.randrange(r), rnd.randrange(1 << 64)])

    def lc():
        # wide: now and then a combination of tens to hundreds of terms over few variables (every variable several times, some
        # cancelling): what the reference's optimizer leaves of an inlined lazy sum, and more than the reader's short-row merge takes
        length = rnd.choice([23, 24, 25, 26, 60, 300]) if wide and rnd.random() < 0.3 else rnd.randrange(0, 5)
        t = [(rnd.choice(pool), coeff()) for _ in range(length)]
        if t and rnd.random() < 0.3:
            v, c = t[0]
            t.append((v, (r - c) % r))                                        # cancels -> zero after merging
        if t and rnd.random() < 0.3:
            t.append((t[-1][0], coeff()))                                     # duplicate variable
        return t

    stmts = []
    for k in range(n):
        if with_noise and rnd.random() < 0.4:
            stmts.append(ir.Other("Directive", {"span": None, "inputs": [ir._lc(lc())], "outputs": [{"id": rnd.choice(pool)}],
                                                "solver": {"Ref": {"index": k, "argument_count": 2}}}))
        if with_noise and rnd.random() < 0.2:
            stmts.append(ir.Other("Log", {"span": None, "format_string": {"parts": ["x = ", ""]}, "expressions": [["Int", [ir._lc(lc())]]]}))
        span = (0x1234567890abcdef, (k + 1, 3), (k + 1, 300)) if rnd.random() < 0.5 else None
        error = rnd.choice([None, "ArkConstraint", {"SourceAssertion": {"file": "main.zok", "position": {"line": 3, "col": 9}, "message": None}}])
        stmts.append(ir.Constraint(lc(), lc(), lc(), span=span, error=error))
    return ir.Prog(curve, args, stmts, return_count=n_out)


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_parse_matches_ark_order(lib, curve):
    rnd = random.Random(77)
    for trial in range(12):
        prog = random_prog(curve, rnd, n=rnd.randrange(0, 40), n_args=rnd.randrange(0, 6), n_out=rnd.randrange(0, 4))
        p = native.Program(ir.serialize_prog(prog), lib)
        l, w, order, rows = ir.ark_order(prog)
        n = sum(isinstance(s, ir.Constraint) for s in prog.statements)
        assert (p.curve_id, p.n, p.l, p.w, p.return_count) == (curve.curve_id, n, l, w, prog.return_count)
        assert p.n_public_args == sum(not a.private for a in prog.arguments)
        assert list(p.variable_order()) == order
        assert rows_of(p.mats(), n) == rows
        # witness file -> z in ark order + public_inputs_values
        used = set(order) | {-(i + 1) for i in range(3)}
        values = {v: (1 if v == 0 else rnd.randrange(curve.r)) for v in used}
        z, inputs = p.assignment(ir.serialize_witness(values))
        assert [int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(l + w)] == [1] + [values[v] for v in order[1:]]
        want_inputs = ir.public_inputs_values(prog, values)
        assert [int.from_bytes(inputs[32 * j:32 * j + 32].tobytes(), "little") for j in range(len(inputs) // 32)] == want_inputs
        p.close()


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_wide_combinations_merge_like_short_ones(lib, curve):
    """Rows on either side of the reader's switch from the quadratic merge to the sorted one (24 terms), duplicates and
    cancellations in both: columns, order of first occurrence and values as the Python model of ark's allocation has them."""
    rnd = random.Random(4242)
    for trial in range(6):
        prog = random_prog(curve, rnd, n=rnd.randrange(5, 30), n_args=3, n_out=2, with_noise=False, wide=True)
        p = native.Program(ir.serialize_prog(prog), lib)
        l, w, order, rows = ir.ark_order(prog)
        n = sum(isinstance(s, ir.Constraint) for s in prog.statements)
        assert (p.n, p.l, p.w) == (n, l, w)
        assert list(p.variable_order()) == order
        assert rows_of(p.mats(), n) == rows
        assert max(len(r) for m in rows for r in m) > 24 or trial
        p.close()


def test_compiler_shaped_program(lib):
    """The shape the ZoKrates compiler emits: canonical combinations, outputs defined in index order — the instance
    order [ONE, public args, ~out_0, ~out_1] then equals public_inputs_values (what makes a proof verify)."""
    curve = BN254
    r = curve.r
    # def main(private field a, field b) -> (field, field): return a * b, a * b + b
    prog = ir.Prog(curve, [ir.Parameter(1, True), ir.Parameter(2, False)], [
        ir.Other("Directive", {"span": None, "inputs": [], "outputs": [{"id": 3}], "solver": "ConditionEq"}),
        ir.Constraint([(1, 1)], [(2, 1)], [(3, 1)]),
        ir.Constraint([(0, 1)], [(3, 1)], [(-1, 1)]),
        ir.Constraint([(0, 1)], [(2, 1), (3, 1)], [(-2, 1)]),
    ], return_count=2)
    p = native.Program(ir.serialize_prog(prog), lib)
    assert (p.n, p.l, p.w) == (3, 4, 2)
    assert list(p.variable_order()) == [0, 2, -1, -2, 1, 3]
    a, b = 7, 9
    z, inputs = p.assignment(ir.serialize_witness({0: 1, 1: a, 2: b, 3: a * b, -1: a * b, -2: a * b + b}))
    zi = [int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(6)]
    assert zi == [1, b, a * b, a * b + b, a, a * b]
    assert [int.from_bytes(inputs[32 * j:32 * j + 32].tobytes(), "little") for j in range(3)] == zi[1:4]
    cs = g16.R1CS(l=p.l, w=p.w)
    cs.A, cs.B, cs.C = ([list(row.items()) for row in m] for m in rows_of(p.mats(), p.n))
    assert cs.is_satisfied(zi, r)


def test_header_and_error_paths(lib):
    curve = BN254
    prog = random_prog(curve, random.Random(5), n=6, with_noise=False)
    good = ir.serialize_prog(prog)
    assert good[:4] == b"ZOK\0" and good[8:12] == bytes.fromhex("b4f7b5bd")     # zokrates_book/src/toolbox/ir.md:13-15
    assert ir.curve_id_bytes(BLS12_381) == bytes.fromhex("40d8c1f9")

    def err(data, code=-2):
        with pytest.raises(native.ZkhipError) as e:
            native.Program(data, lib)
        assert e.value.code == code, e.value
        return str(e.value)

    assert "magic" in err(b"ZOL\0" + good[4:])
    assert "version" in err(good[:4] + bytes([2, 0, 0, 0]) + good[8:])
    assert "curve" in err(good[:8] + b"\x12\x34\x56\x78" + good[12:], code=-1)  # e.g. bls12_377 / bw6_761: other backends
    err(good[:50])                                                               # truncated header
    err(good[:130])                                  # sections point outside the file
    bad = bytearray(good)
    bad[12:16] = struct.pack("<I", 99)                                           # constraint count mismatch
    assert "constraint count" in err(bytes(bad))
    # a coefficient >= r
    p2 = ir.Prog(curve, [], [ir.Constraint([(1, curve.r)], [(0, 1)], [(2, 1)])])
    raw = ir.serialize_prog(p2).replace((curve.r % (1 << 256)).to_bytes(32, "little"), curve.r.to_bytes(32, "little"))
    assert "canonical" in err(raw)
    # witness errors
    p = native.Program(good, lib)
    order = list(p.variable_order())
    values = {v: 3 for v in order}
    p.assignment(ir.serialize_witness(values))
    missing = dict(values)
    del missing[order[-1]]
    with pytest.raises(native.ZkhipError) as e:
        p.assignment(ir.serialize_witness(missing))
    assert e.value.code == -5 and "missing" in str(e.value)
    with pytest.raises(native.ZkhipError) as e:
        p.assignment(ir.serialize_witness(values)[:-1])
    assert e.value.code == -2
    with pytest.raises(native.ZkhipError) as e:
        p.assignment(ir.serialize_witness({**values, order[1]: curve.r}))
    assert e.value.code == -2


def test_indefinite_length_cbor_is_accepted(lib):
    """serde_cbor writes definite lengths, but a streaming producer may not: both decode to the same system."""
    curve = BN254
    prog = ir.Prog(curve, [ir.Parameter(1, False)], [ir.Constraint([(1, 2), (0, 5)], [(2, 1)], [(-1, 1)])], return_count=1)
    good = ir.serialize_prog(prog)
    st = ir.cbor(ir._statement(prog.statements[0]))
    lcs = [ir.cbor(ir._lc(t)) for t in ([(1, 2), (0, 5)], [(2, 1)], [(-1, 1)])]
    key = lambda s: ir.cbor(s)
    indef_lc = lambda blob_terms: b"\xbf" + key("span") + b"\xf6" + key("value") + b"\x9f" + b"".join(
        ir.cbor([{"id": v}, int(c).to_bytes(32, "little")]) for v, c in blob_terms) + b"\xff" + b"\xff"
    st2 = (b"\xbf" + key("Constraint") + b"\xbf" + key("span") + b"\xf6" + key("quad") + b"\xbf" + key("span") + b"\xf6" + key("left")
           + indef_lc([(1, 2), (0, 5)]) + key("right") + indef_lc([(2, 1)]) + b"\xff" + key("lin") + indef_lc([(-1, 1)]) + key("error") + b"\xf6"
           + b"\xff" + b"\xff")
    # the top-level statement map must announce one entry: keep it definite, make everything inside indefinite
    st2 = b"\xa1" + st2[1:-1]
    i = good.index(st)
    raw = bytearray(good[:i] + st2 + good[i + len(st):])
    delta = len(st2) - len(st)
    # fix up section table: constraints length, solvers / modules offsets
    off = 20
    secs = [list(struct.unpack_from("<IQQ", raw, off + 20 * s)) for s in range(4)]
    secs[1][2] += delta; secs[2][1] += delta; secs[3][1] += delta
    for s in range(4):
        struct.pack_into("<IQQ", raw, off + 20 * s, *secs[s])
    a, b = native.Program(good, lib), native.Program(bytes(raw), lib)
    assert rows_of(a.mats(), 1) == rows_of(b.mats(), 1) and list(a.variable_order()) == list(b.variable_order())


@pytest.mark.gpu
def test_gpu_prove_from_zokrates_files():
    """`out` + `witness` + `proving.key` -> proof, all through the C ABI; equals the oracle's closed-form proof."""
    from oracle import cpu
    curve = BN254
    ctx = native.Context(0)
    cs_py, z = g16.synthetic_chain(curve, 200, 99)
    # the same circuit as a ZoKrates program: x public, everything else private and discovered in first-use order
    stmts = [ir.Constraint(list(a), list(b), list(c)) for a, b, c in zip(cs_py.A, cs_py.B, cs_py.C)]
    # column j of the synthetic system <-> ZoKrates variable: 0 -> ~one, 1 -> _0 (public argument), j >= 2 -> _{j-1}
    prog = ir.Prog(curve, [ir.Parameter(1, False)], stmts, return_count=0)
    p = native.Program(ir.serialize_prog(prog))
    assert (p.n, p.l, p.w) == (cs_py.n, cs_py.l, cs_py.w)
    zz, inputs = p.assignment(ir.serialize_witness({j: v for j, v in enumerate(z)}))
    order = list(p.variable_order())
    z_ark = [z[v] for v in order]
    dcs = p.constraint_system(ctx)
    tox = g16.Toxic.from_seed(curve)
    raw_pk = native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
    pk = native.ProvingKey(ctx, 0, raw_pk)
    got = native.prove_g16(ctx, pk, dcs, zz, 11, 13)
    # oracle on the ark-ordered system
    l, w, _, rows = ir.ark_order(prog)
    ocs = g16.R1CS(l=l, w=w)
    ocs.A, ocs.B, ocs.C = ([list(row.items()) for row in m] for m in rows)
    assert ocs.is_satisfied(z_ark, curve.r)
    assert got == formats.proof_raw(curve, g16.trapdoor_prove(curve, ocs, tox, z_ark, 11, 13))
    assert int.from_bytes(inputs.tobytes(), "little") == z[1]
    ctx.close()


# ------------------------------------------------------------------ the CLI shim over ZoKrates' own files
def _cli_flow_zok(tmp_path, scheme):
    """setup + generate-proof from `out` / `witness` (both schemes); the JSON artefacts pass the oracle's pairing check
    with `inputs` = public arguments then outputs."""
    import json
    from oracle import gm17, pairing
    from zokrates_amd import cli
    curve = BN254
    # def main(private field a, field b) -> (field, field): return a * b, a * b + b
    prog = ir.Prog(curve, [ir.Parameter(1, True), ir.Parameter(2, False)], [
        ir.Other("Directive", {"span": None, "inputs": [], "outputs": [{"id": 3}], "solver": "ConditionEq"}),
        ir.Constraint([(1, 1)], [(2, 1)], [(3, 1)]),
        ir.Constraint([(0, 1)], [(3, 1)], [(-1, 1)]),
        ir.Constraint([(0, 1)], [(2, 1), (3, 1)], [(-2, 1)]),
    ], return_count=2)
    a, b = 1234567, 7654321
    outp = tmp_path / "out"; wit = tmp_path / "witness"; pkp = tmp_path / "proving.key"; vkp = tmp_path / "verification.key"; pj = tmp_path / "proof.json"
    outp.write_bytes(ir.serialize_prog(prog))
    wit.write_bytes(ir.serialize_witness({0: 1, 1: a, 2: b, 3: a * b, -1: a * b, -2: a * b + b}))
    cli.main(["setup", "-i", str(outp), "-p", str(pkp), "-v", str(vkp), "-s", scheme, "--entropy", "unit test"])
    cli.main(["generate-proof", "-i", str(outp), "-w", str(wit), "-p", str(pkp), "-j", str(pj), "-s", scheme, "--entropy", "abc"])
    proof = json.loads(pj.read_text())
    vk = json.loads(vkp.read_text())
    assert proof["scheme"] == vk["scheme"] == scheme and proof["curve"] == "bn128"
    h = lambda s: int(s, 16)
    g1 = lambda p: (h(p[0]), h(p[1]))
    g2 = lambda p: ((h(p[0][0]), h(p[0][1])), (h(p[1][0]), h(p[1][1])))
    pts = (g1(proof["proof"]["a"]), g2(proof["proof"]["b"]), g1(proof["proof"]["c"]))
    inputs = [h(x) for x in proof["inputs"]]
    assert inputs == [b, a * b, a * b + b]
    if scheme == "g16":
        ovk = dict(alpha_g1=g1(vk["alpha"]), beta_g2=g2(vk["beta"]), gamma_g2=g2(vk["gamma"]), delta_g2=g2(vk["delta"]),
                   gamma_abc_g1=[g1(p) for p in vk["gamma_abc"]])
        check = lambda inp: pairing.groth16_verify(curve, ovk, pts, inp)
    else:
        assert list(vk)[:2] == ["scheme", "curve"] and list(vk)[2:] == ["h", "g_alpha", "h_beta", "g_gamma", "h_gamma", "query"]
        ovk = dict(h_g2=g2(vk["h"]), g_alpha_g1=g1(vk["g_alpha"]), h_beta_g2=g2(vk["h_beta"]), g_gamma_g1=g1(vk["g_gamma"]),
                   h_gamma_g2=g2(vk["h_gamma"]), query=[g1(p) for p in vk["query"]])
        check = lambda inp: gm17.verify(curve, ovk, pts, inp)
    assert check(inputs)
    assert not check([inputs[0], inputs[1], inputs[2] + 1])


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_cli_zokrates_files_on_emulator(tmp_path, monkeypatch, scheme):
    from emu_util import emu_library
    monkeypatch.setattr(native, "_default", emu_library())
    _cli_flow_zok(tmp_path, scheme)


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["g16", "gm17"])
def test_cli_zokrates_files_on_gpu(tmp_path, scheme):
    _cli_flow_zok(tmp_path, scheme)


# ------------------------------------------------------------------ N2: device-layout key images
def _key_image_checks(ctx):
    from oracle import cpu, gm17
    curve = BN254
    cs_py, z = g16.synthetic_chain(curve, 12, 7)
    from test_gm17 import csr_of, le
    dcs = native.ConstraintSystem(ctx, 0, cs_py.n, cs_py.l, cs_py.w, [csr_of(cs_py.A), csr_of(cs_py.B), csr_of(cs_py.C)])
    tox = g16.Toxic.from_seed(curve)
    for scheme, raw in (("g16", native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))),
                        ("gm17", native.setup_gm17(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.tau)))):
        pk = native.ProvingKey(ctx, 0, raw, scheme=scheme)
        image = pk.export_image()
        pk2 = native.ProvingKey.from_image(ctx, 0, image, scheme=scheme)
        assert (pk2.m, pk2.hlen, pk2.w, pk2.l) == (pk.m, pk.hlen, pk.w, pk.l)
        assert pk2.export_image().tobytes() == image.tobytes()
        if scheme == "g16":
            assert native.prove_g16(ctx, pk2, dcs, le(z), 3, 4) == native.prove_g16(ctx, pk, dcs, le(z), 3, 4)
            with pytest.raises(native.ZkhipError):                       # the scheme travels with the image
                native.prove_gm17(ctx, pk2, dcs, le(z), 1, 2, 3)
        else:
            assert native.prove_gm17(ctx, pk2, dcs, le(z), 1, 2, 3) == native.prove_gm17(ctx, pk, dcs, le(z), 1, 2, 3)
        for bad in (image[:-1], image[:40], np.concatenate([image, image[:1]]), np.concatenate([np.frombuffer(b"ZKHIPPK0", dtype=np.uint8), image[8:]])):
            with pytest.raises(native.ZkhipError) as e:
                native.ProvingKey.from_image(ctx, 0, bad, scheme=scheme)
            assert e.value.code == -2
        # header fields the kernels would trust (index range lengths at byte offsets 88 / 104) must match the array sizes
        for off in (88, 104):
            bad = image.copy()
            bad[off:off + 8] = np.frombuffer(struct.pack("<Q", int.from_bytes(image[off:off + 8].tobytes(), "little") + 1), dtype=np.uint8)
            with pytest.raises(native.ZkhipError) as e:
                native.ProvingKey.from_image(ctx, 0, bad, scheme=scheme)
            assert e.value.code == -2
    # a shard keeps its index range
    shard = native.ProvingKey(ctx, 0, native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau)), rank=1, world=3)
    sh2 = native.ProvingKey.from_image(ctx, 0, shard.export_image())
    raw16 = native.setup_g16(ctx, dcs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
    others = [native.prove_g16_partial(ctx, native.ProvingKey(ctx, 0, raw16, rank=k, world=3), dcs, le(z), 5, 6) for k in (0, 2)]
    p_img, p_ref = native.prove_g16_partial(ctx, sh2, dcs, le(z), 5, 6), native.prove_g16_partial(ctx, shard, dcs, le(z), 5, 6)
    whole = native.prove_g16(ctx, native.ProvingKey(ctx, 0, raw16), dcs, le(z), 5, 6)
    assert native.combine_g16(ctx, shard, [others[0], p_img, others[1]], 5, 6) == whole
    assert p_img.tobytes() == p_ref.tobytes()      # canonical records: equal shares are equal bytes


def test_key_image_on_emulator(lib):
    ctx = native.Context(0, lib)
    _key_image_checks(ctx)
    ctx.close()


@pytest.mark.gpu
def test_key_image_on_gpu():
    ctx = native.Context(0)
    _key_image_checks(ctx)
    ctx.close()


def test_cli_key_cache_on_emulator(tmp_path, monkeypatch):
    from emu_util import emu_library
    from zokrates_amd import cli
    monkeypatch.setattr(native, "_default", emu_library())
    curve = BN254
    prog = ir.Prog(curve, [ir.Parameter(1, True), ir.Parameter(2, False)], [ir.Constraint([(1, 1)], [(2, 1)], [(-1, 1)])], return_count=1)
    outp = tmp_path / "out"; wit = tmp_path / "witness"; pkp = tmp_path / "proving.key"; vkp = tmp_path / "verification.key"
    outp.write_bytes(ir.serialize_prog(prog))
    wit.write_bytes(ir.serialize_witness({0: 1, 1: 6, 2: 7, -1: 42}))
    cli.main(["setup", "-i", str(outp), "-p", str(pkp), "-v", str(vkp), "--entropy", "k"])
    cache = tmp_path / "cache"
    proofs = []
    for run in range(3):                                   # first run parses and writes the image, later runs import it
        pj = tmp_path / f"proof{run}.json"
        cli.main(["generate-proof", "-i", str(outp), "-w", str(wit), "-p", str(pkp), "-j", str(pj), "--entropy", "e", "--key-cache", str(cache)])
        proofs.append(pj.read_text())
        assert len(list(cache.iterdir())) == 1
    assert proofs[0] == proofs[1] == proofs[2]
    pj = tmp_path / "proof_nocache.json"
    cli.main(["generate-proof", "-i", str(outp), "-w", str(wit), "-p", str(pkp), "-j", str(pj), "--entropy", "e"])
    assert pj.read_text() == proofs[0]


# ------------------------------------------------------------------ the reference backend's own unit test, mirrored
def _reference_backend_test(ctx, curve, scheme):
    """/root/reference/zokrates_ark/src/groth16.rs:123-161 and gm17.rs:124-160 (there over BLS12-377 / BW6-761; here over
    the two curves this backend proves on): the program `(1) * (_0) == ~out_0` with `_0` a public argument, input 42,
    setup -> generate_proof -> verify.  No witness variable at all (w = 0): every query vector of the key that ranges
    over the witness is empty."""
    from oracle import gm17, pairing
    prog = ir.Prog(curve, [ir.Parameter(1, False)], [ir.Constraint([(0, 1)], [(1, 1)], [(-1, 1)])], return_count=1)
    p = native.Program(ir.serialize_prog(prog), ctx.lib)
    assert (p.n, p.l, p.w) == (1, 3, 0) and list(p.variable_order()) == [0, 1, -1]
    z, inputs = p.assignment(ir.serialize_witness({0: 1, 1: 42, -1: 42}))
    inp = [int.from_bytes(inputs[32 * i:32 * i + 32].tobytes(), "little") for i in range(2)]
    assert inp == [42, 42]
    cs = p.constraint_system(ctx)
    tox = g16.Toxic.from_seed(curve, 0xBEEF)
    if scheme == "g16":
        raw = native.setup_g16(ctx, cs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
        pk = native.ProvingKey(ctx, curve.curve_id, raw)
        assert (pk.m, pk.w, pk.l, pk.hlen) == (3, 0, 3, 3)
        proof = formats.proof_from_raw(curve, native.prove_g16(ctx, pk, cs, z, 1111, 2222))
        vk = formats.ark_pk_deserialize(curve, raw.tobytes())["vk"]
        assert pairing.groth16_verify(curve, vk, proof, inp)
        assert not pairing.groth16_verify(curve, vk, proof, [42, 43])
    else:
        raw = native.setup_gm17(ctx, cs, (tox.alpha, tox.beta, 1, tox.tau))
        pk = native.ProvingKey(ctx, curve.curve_id, raw, scheme="gm17")
        proof = formats.proof_from_raw(curve, native.prove_gm17(ctx, pk, cs, z, 1111, 2222, 3333))
        vk = gm17.vk_from_pk_bytes(curve, raw)
        assert vk["h_g2"] == vk["h_gamma_g2"]                       # gamma = 1, as in ark's generate_random_parameters
        assert gm17.verify(curve, vk, proof, inp)
        assert not gm17.verify(curve, vk, proof, [42, 43])


@pytest.mark.parametrize("scheme", ["g16", "gm17"])
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_reference_backend_unit_test_on_emulator(lib, curve, scheme):
    ctx = native.Context(0, lib)
    _reference_backend_test(ctx, curve, scheme)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("scheme", ["g16", "gm17"])
@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_reference_backend_unit_test_on_gpu(curve, scheme):
    ctx = native.Context(0)
    _reference_backend_test(ctx, curve, scheme)
    ctx.close()


def test_empty_program_on_emulator(lib):
    """`Prog::default()` (what /root/reference/zokrates_ast/src/ir/serialize.rs:396-430 round-trips): no arguments, no
    statements — n = 0, one variable (ONE), domain of size 1, every query vector empty or a single point.  Both schemes
    still produce proofs that verify."""
    from oracle import gm17, pairing
    curve = BN254
    ctx = native.Context(0, lib)
    p = native.Program(ir.serialize_prog(ir.Prog(curve, [], [], return_count=0)), lib)
    assert (p.n, p.l, p.w) == (0, 1, 0)
    z, inputs = p.assignment(ir.serialize_witness({0: 1}))
    assert inputs.size == 0
    cs = p.constraint_system(ctx)
    tox = g16.Toxic.from_seed(curve)
    raw = native.setup_g16(ctx, cs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
    pk = native.ProvingKey(ctx, 0, raw)
    assert (pk.m, pk.w, pk.l, pk.hlen) == (1, 0, 1, 0)
    proof = formats.proof_from_raw(curve, native.prove_g16(ctx, pk, cs, z, 11, 22))
    assert pairing.groth16_verify(curve, formats.ark_pk_deserialize(curve, raw.tobytes())["vk"], proof, [])
    raw = native.setup_gm17(ctx, cs, (tox.alpha, tox.beta, 1, tox.tau))
    pk = native.ProvingKey(ctx, 0, raw, scheme="gm17")
    proof = formats.proof_from_raw(curve, native.prove_gm17(ctx, pk, cs, z, 1, 2, 3))
    assert gm17.verify(curve, gm17.vk_from_pk_bytes(curve, raw), proof, [])
    ctx.close()


def test_parser_survives_mutations(lib):
    """The C ABI never throws across the boundary and never aborts: byte flips, truncations, insertions and stray CBOR
    structure bytes in an `out` file (and garbage `witness` files) give a parsed program or a ZkhipError — nothing else;
    an absurd variable id is an error, not a multi-gigabyte allocation."""
    rnd = random.Random(1)
    prog = random_prog(BN254, rnd, n=12)
    good = ir.serialize_prog(prog)
    parsed = rejected = 0
    for _ in range(3000):
        b = bytearray(good)
        k = rnd.randrange(4)
        if k == 0:
            for _ in range(rnd.randrange(1, 4)):
                b[rnd.randrange(len(b))] = rnd.randrange(256)
        elif k == 1:
            b = b[:rnd.randrange(len(b))]
        elif k == 2:
            i = rnd.randrange(len(b))
            b[i:i] = bytes(rnd.randrange(256) for _ in range(rnd.randrange(1, 9)))
        else:
            b[rnd.randrange(120, len(b))] = rnd.choice([0x9F, 0xBF, 0xFF, 0x5F, 0x7F, 0xFB, 0x1B, 0x3B, 0xC0, 0xD8])
        try:
            p = native.Program(bytes(b), lib)
            parsed += 1
            try:
                p.assignment(bytes(rnd.randrange(256) for _ in range(8 + 40 * 3)))
            except native.ZkhipError:
                pass
            p.close()
        except native.ZkhipError:
            rejected += 1
    assert parsed > 100 and rejected > 100
    huge = ir.Prog(BN254, [], [ir.Constraint([((1 << 31) - 2, 1)], [(0, 1)], [(1, 1)])])
    with pytest.raises(native.ZkhipError) as e:
        native.Program(ir.serialize_prog(huge), lib)
    assert e.value.code == -2 and "out of range" in str(e.value)
    p = native.Program(good, lib)
    with pytest.raises(native.ZkhipError) as e:
        p.assignment(struct.pack("<Q", 1) + struct.pack("<q", (1 << 31) - 2) + bytes(32))
    assert e.value.code == -2


# ---------------------------------------------------------------- chunked / parallel decode, and the writer
def _snapshot(p):
    mats = p.mats()
    return ((p.curve_id, p.n, p.l, p.w, p.return_count, p.n_public_args, p.nnz), list(p.variable_order()),
            [(rp.tobytes(), col.tobytes(), val.tobytes()) for rp, col, val in mats])


def test_parallel_decode_equals_sequential(lib, monkeypatch):
    """The constraint section is decoded in chunks that start at the byte pattern of `{"Constraint":`; a cut is trusted only
    if the chunk before it ends exactly there.  Programs with that pattern inside directive payloads, byte strings and
    coefficients, cut every few dozen bytes, must come out exactly as from one sequential pass (variables are numbered in
    first-seen order ACROSS chunks)."""
    curve = BN254
    pat = bytes([0xa1, 0x6a]) + b"Constraint"
    rnd = random.Random(5)
    for trial in range(5):
        prog = random_prog(curve, rnd, n=150, n_args=3, n_out=2)
        poison = ir.Other("Directive", {"span": None, "inputs": [], "outputs": [],
                                        "solver": {"Zir": {"blob": pat + b"\0" * 7, "nested": {"Constraint": {"span": None, "quad": 1}}}}})
        for pos in (5, 60, 140):
            prog.statements.insert(pos, poison)
        evil = int.from_bytes(pat + bytes(20), "little")            # a coefficient whose bytes start like a statement
        prog.statements.insert(30, ir.Constraint([(7, evil)], [(8, evil), (0, 1)], [(9, 1)]))
        data = ir.serialize_prog(prog)
        monkeypatch.setenv("ZKHIP_INGEST_THREADS", "1")
        want = _snapshot(native.Program(data, lib))
        l, w, order, rows = ir.ark_order(prog)
        assert want[1] == order
        for threads, chunk in ((8, 64), (3, 1000), (64, 16), (5, 333)):
            monkeypatch.setenv("ZKHIP_INGEST_THREADS", str(threads))
            monkeypatch.setenv("ZKHIP_INGEST_MIN_CHUNK", str(chunk))
            assert _snapshot(native.Program(data, lib)) == want, (trial, threads, chunk)
        # errors are the sequential parser's errors: a truncated file, a flipped byte deep inside
        for bad in (data[:len(data) * 2 // 3], data[:700] + bytes([data[700] ^ 0xff]) + data[701:]):
            outcomes = []
            for threads in ("1", "8"):
                monkeypatch.setenv("ZKHIP_INGEST_THREADS", threads)
                try:
                    outcomes.append(_snapshot(native.Program(bad, lib)))
                except native.ZkhipError as e:
                    outcomes.append((e.code, str(e)))
            assert outcomes[0] == outcomes[1]


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_program_writer_round_trip(lib, curve, monkeypatch):
    """zkhip_prog_write (R1CS -> `out`) against the python writer byte for byte, and through the reader: the synthetic
    benchmark circuit comes back with the same matrices in the same column order."""
    from zokrates_amd import synth
    circ = synth.circuit(curve.curve_id, 9, kind="sha", seed=11)
    data = native.write_program(curve.curve_id, circ.n, circ.m, circ.mats(), args=[(1, False)], library=lib)
    # the same program through oracle/ir.py's writer
    stmts = []
    rows = rows_of(circ.mats(), circ.n) if False else None
    mats = circ.mats()
    def row(k, i):
        rp, col, val = mats[k]
        return [(int(col[q]), int.from_bytes(val[32 * q:32 * q + 32].tobytes(), "little")) for q in range(int(rp[i]), int(rp[i + 1]))]
    for i in range(circ.n):
        stmts.append(ir.Constraint(row(0, i), row(1, i), row(2, i)))
    prog = ir.Prog(curve, [ir.Parameter(1, False)], stmts, return_count=0)
    assert data.tobytes() == ir.serialize_prog(prog)
    for threads in ("1", "6"):
        monkeypatch.setenv("ZKHIP_INGEST_THREADS", threads)
        monkeypatch.setenv("ZKHIP_INGEST_MIN_CHUNK", "4096")
        p = native.Program(data, lib)
        assert (p.n, p.l, p.w) == (circ.n, circ.l, circ.w)
        # every variable of the sha-like circuit appears, in column order, except that boolean rows mention their variable
        # before the chain reaches it: compare as sets of rows under the returned order
        order = list(p.variable_order())
        assert sorted(order) == list(range(circ.m))
        perm = {zid: j for j, zid in enumerate(order)}                  # original column -> column after the reader
        got = rows_of(p.mats(), circ.n)
        for k in range(3):
            for i in range(circ.n):
                assert got[k][i] == {perm[c]: v for c, v in row(k, i)}, (k, i)
        z = circ.assignment(3)
        wit = native.write_witness(np.arange(circ.m), z)
        assert wit.tobytes() == ir.serialize_witness({j: int.from_bytes(z[32 * j:32 * j + 32].tobytes(), "little") for j in range(circ.m)})
        zz, inputs = p.assignment(wit)
        assert zz.reshape(-1, 32)[[perm[j] for j in range(circ.m)]].tobytes() == z.tobytes()
        assert inputs.tobytes() == z[32:64].tobytes()
