"""The program shapes of the reference's OTHER backend's unit tests — /root/reference/zokrates_bellman/src/lib.rs:236-474:
empty, identity, public identity, no arguments, unordered variables, constant one, "with directives" — pushed through the
native reader (`zkhip_prog_*`), the device setup and prover, and the pairing check, with `inputs` in
`public_inputs_values` order (/root/reference/zokrates_ast/src/ir/mod.rs:278-288).

The `out` and `witness` bytes of this file are assembled HERE, byte by byte, from the reference's writer —
/root/reference/zokrates_ast/src/ir/serialize.rs:134-189 (header: magic, version, curve id, counts, four sections of
u32 type + u64 offset + u64 length, in a 120-byte region), :202-279 (sections: parameters, constraints as a stream of CBOR
items, solvers, module map) and the serde derives of `Parameter` (common/parameter.rs:8-16), `Variable`
(common/flat/variable.rs:10-13), `ConstraintStatement` / `Statement` (ir/mod.rs:33-128), `QuadComb` / `LinComb`
(ir/expression.rs:10-18, 69-76), `Span` (common/position.rs:59-63), field elements as 32-byte CBOR byte strings
(zokrates_field/src/lib.rs:547-560) — NOT through oracle/ir.py, the writer the other ingest tests use: the reader is
checked here against bytes its author's writer did not produce."""
import hashlib
import struct

import numpy as np
import pytest

from emu_util import emu_library
from oracle import formats, gm17, pairing
from oracle import groth16 as g16
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native


# ---- CBOR, spelled out (RFC 8949 heads; serde_cbor 0.11 writes definite lengths, struct = map keyed by field name,
# ---- enum = single-entry map keyed by variant name, Option::None = null, bool = true/false, isize = int)
def t(s):                       # text string, length < 24
    assert len(s) < 24
    return bytes([0x60 + len(s)]) + s.encode()


def uint(v):                    # unsigned, < 2^16
    return bytes([v]) if v < 24 else (bytes([0x18, v]) if v < 256 else b"\x19" + struct.pack(">H", v))


def nint(v):                    # negative integer -1 - n, n < 24
    assert -24 <= v < 0
    return bytes([0x20 + (-1 - v)])


def integer(v):
    return uint(v) if v >= 0 else nint(v)


def array(n):
    assert n < 24
    return bytes([0x80 + n])


def amap(n):
    assert n < 24
    return bytes([0xa0 + n])


NULL, TRUE, FALSE = b"\xf6", b"\xf5", b"\xf4"


def fe(v, r):                   # a field element: byte string of 32 (0x58 0x20), canonical little-endian (ark CanonicalSerialize)
    return b"\x58\x20" + int(v % r).to_bytes(32, "little")


def variable(vid):              # Variable { id: isize }
    return amap(1) + t("id") + integer(vid)


def lincomb(terms, r):          # LinComb { span: None, value: Vec<(Variable, T)> }  (a tuple is a 2-array)
    body = b"".join(array(2) + variable(v) + fe(c, r) for v, c in terms)
    return amap(2) + t("span") + NULL + t("value") + array(len(terms)) + body


def constraint(left, right, lin, r):
    quad = amap(3) + t("span") + NULL + t("left") + lincomb(left, r) + t("right") + lincomb(right, r)
    st = amap(4) + t("span") + NULL + t("quad") + quad + t("lin") + lincomb(lin, r) + t("error") + NULL
    return amap(1) + t("Constraint") + st


def parameter(vid, private):    # Parameter { span, id, private }
    return amap(3) + t("span") + NULL + t("id") + variable(vid) + t("private") + (TRUE if private else FALSE)


def out_file(curve, params, statements, n_constraints, return_count, solvers=array(0)):
    ps = array(len(params)) + b"".join(params)
    cs = b"".join(statements)
    modules = amap(1) + t("modules") + amap(0)                     # ModuleMap { modules: Vec / map, empty }
    nbytes = (curve.r.bit_length() + 63) // 64 * 8
    curve_id = hashlib.sha256(curve.r.to_bytes(nbytes, "little")).digest()[:4]      # Field::id(), zokrates_field/src/lib.rs:283-293
    header = b"ZOK\x00" + b"\x03\x00\x00\x00" + curve_id + struct.pack("<II", n_constraints, return_count)
    off = 120                                                      # size_of::<ProgHeader>(): 100 bytes written, 120 reserved
    for ty, blob in ((1, ps), (2, cs), (3, solvers), (3, modules)):                  # (module map is tagged Solvers: serialize.rs:247)
        header += struct.pack("<IQQ", ty, off, len(blob))
        off += len(blob)
    assert len(header) == 100
    return header + b"\x00" * 20 + ps + cs + solvers + modules


def witness_file(values, r):    # ir/witness.rs:44-53: usize count, then (isize id, 32-byte value) in BTreeMap (ascending id) order
    out = struct.pack("<Q", len(values))
    for vid in sorted(values):
        out += struct.pack("<q", vid) + int(values[vid] % r).to_bytes(32, "little")
    return out


ONE = 0


def new(k):                     # Variable::new(k)  -> id k + 1
    return k + 1


def public(k):                  # Variable::public(k) -> id -k - 1   (~out_k)
    return -k - 1


def shapes(curve):
    """name -> (out bytes, witness values, expected (n, l, w), expected ark variable order, expected inputs)."""
    r = curve.r
    one = [(ONE, 1)]
    s = {}
    # empty: Prog::default()
    s["empty"] = (out_file(curve, [], [], 0, 0), {ONE: 1}, (0, 1, 0), [0], [])
    # identity: private _0; (1) * (_0) == ~out_0; input 0
    s["identity"] = (out_file(curve, [parameter(new(0), True)], [constraint(one, [(new(0), 1)], [(public(0), 1)], r)], 1, 1),
                     {ONE: 1, new(0): 0, public(0): 0}, (1, 2, 1), [0, -1, 1], [0])
    # public identity: public _0
    s["public_identity"] = (out_file(curve, [parameter(new(0), False)], [constraint(one, [(new(0), 1)], [(public(0), 1)], r)], 1, 1),
                            {ONE: 1, new(0): 0, public(0): 0}, (1, 3, 0), [0, 1, -1], [0, 0])
    # no arguments: (1) * (~one) == ~out_0
    s["no_arguments"] = (out_file(curve, [], [constraint(one, [(ONE, 1)], [(public(0), 1)], r)], 1, 1),
                         {ONE: 1, public(0): 1}, (1, 2, 0), [0, -1], [1])
    # unordered variables: private _42, public _51; (1) * (_42 + _51) == ~out_0; (1) * (~one + _42) == ~out_1; inputs 3, 4
    s["unordered_variables"] = (
        out_file(curve, [parameter(new(42), True), parameter(new(51), False)],
                 [constraint(one, [(new(42), 1), (new(51), 1)], [(public(0), 1)], r),
                  constraint(one, [(ONE, 1), (new(42), 1)], [(public(1), 1)], r)], 2, 2),
        {ONE: 1, new(42): 3, new(51): 4, public(0): 7, public(1): 4}, (2, 4, 1), [0, new(51), -1, -2, new(42)], [4, 7, 4])
    # one: public _42; (1) * (_42 + ~one) == ~out_0; input 3
    s["one"] = (out_file(curve, [parameter(new(42), False)], [constraint(one, [(new(42), 1), (ONE, 1)], [(public(0), 1)], r)], 1, 1),
                {ONE: 1, new(42): 3, public(0): 4}, (1, 3, 0), [0, new(42), -1], [3, 4])
    # "with directives" (the reference's test of that name holds one constraint and, despite its name, no directive):
    # private _42, public _51; (1) * (_42 + _51) == ~out_0
    s["with_directives"] = (
        out_file(curve, [parameter(new(42), True), parameter(new(51), False)],
                 [constraint(one, [(new(42), 1), (new(51), 1)], [(public(0), 1)], r)], 1, 1),
        {ONE: 1, new(42): 3, new(51): 4, public(0): 7}, (1, 3, 1), [0, new(51), -1, new(42)], [4, 7])
    # a real directive, as the compiler emits one for `a / b` (ir statements after SolverIndexer: the directive refers to
    # entry 0 of the solvers section, serialize.rs:214-236; common/statements.rs:155-163, common/solvers.rs:6-23):
    #   _2 = Div(_0, _1);  (_2) * (_1) == _0;  (1) * (_2) == ~out_0        inputs 12, 4 -> 3
    quad = lambda lc: amap(3) + t("span") + NULL + t("left") + lincomb(one, r) + t("right") + lincomb(lc, r)
    directive = amap(1) + t("Directive") + (
        amap(4) + t("span") + NULL + t("inputs") + array(2) + quad([(new(0), 1)]) + quad([(new(1), 1)])
        + t("outputs") + array(1) + variable(new(2))
        + t("solver") + amap(1) + t("Ref") + amap(2) + t("index") + uint(0) + t("signature") + array(2) + uint(2) + uint(1))
    s["division_with_directive"] = (
        out_file(curve, [parameter(new(0), True), parameter(new(1), True)],
                 [directive, constraint([(new(2), 1)], [(new(1), 1)], [(new(0), 1)], r), constraint(one, [(new(2), 1)], [(public(0), 1)], r)],
                 2, 1, solvers=array(1) + t("Div")),
        {ONE: 1, new(0): 12, new(1): 4, new(2): 3, public(0): 3}, (2, 2, 3), [0, -1, new(0), new(1), new(2)], [3])
    return s


NAMES = ["division_with_directive", "empty", "identity", "public_identity", "no_arguments", "unordered_variables", "one", "with_directives"]


def _run(ctx, curve, name, scheme):
    data, wit, dims, order, want_inputs = shapes(curve)[name]
    p = native.Program(np.frombuffer(data, dtype=np.uint8), ctx.lib)
    assert (p.n, p.l, p.w) == dims, name
    # ark's allocation order (zokrates_ark/src/lib.rs:80-129): ONE, public arguments, then ~out_k as first seen -> instance;
    # private arguments, then everything else as first seen -> witness
    assert list(p.variable_order()) == order, name
    z, inputs = p.assignment(np.frombuffer(witness_file(wit, curve.r), dtype=np.uint8))
    got_inputs = [int.from_bytes(inputs[32 * i:32 * i + 32].tobytes(), "little") for i in range(inputs.size // 32)]
    assert got_inputs == want_inputs, name                         # public arguments in argument order, then ~out_0, ~out_1, ...
    zi = [int.from_bytes(z[32 * i:32 * i + 32].tobytes(), "little") for i in range(p.m)]
    assert zi == [wit[v] % curve.r for v in order], name
    # the first l - 1 instance values after ONE are what the verifier is given
    assert zi[1:p.l] == want_inputs, name
    cs = p.constraint_system(ctx)
    tox = g16.Toxic.from_seed(curve, 0xBE11 + len(name))
    if scheme == "g16":
        raw = native.setup_g16(ctx, cs, (tox.alpha, tox.beta, tox.gamma, tox.delta, tox.tau))
        pk = native.ProvingKey(ctx, curve.curve_id, raw)
        proof = formats.proof_from_raw(curve, native.prove_g16(ctx, pk, cs, z, 0x1111 + len(name), 0x2222))
        vk = formats.ark_pk_deserialize(curve, raw.tobytes())["vk"]
        assert pairing.groth16_verify(curve, vk, proof, want_inputs), name
        if want_inputs:
            assert not pairing.groth16_verify(curve, vk, proof, [want_inputs[0] + 1] + want_inputs[1:]), name
    else:
        raw = native.setup_gm17(ctx, cs, (tox.alpha, tox.beta, 1, tox.tau))
        pk = native.ProvingKey(ctx, curve.curve_id, raw, scheme="gm17")
        proof = formats.proof_from_raw(curve, native.prove_gm17(ctx, pk, cs, z, 0x1111, 0x2222, 0x3333 + len(name)))
        vk = gm17.vk_from_pk_bytes(curve, raw)
        assert gm17.verify(curve, vk, proof, want_inputs), name
        if want_inputs:
            assert not gm17.verify(curve, vk, proof, [want_inputs[0] + 1] + want_inputs[1:]), name


@pytest.mark.parametrize("name", NAMES)
def test_bellman_shapes_on_emulator(name):
    ctx = native.Context(0, emu_library())
    _run(ctx, BN254, name, "g16")
    ctx.close()


@pytest.mark.parametrize("name,curve,scheme", [("unordered_variables", BN254, "gm17"), ("one", BLS12_381, "g16"), ("identity", BLS12_381, "gm17")],
                         ids=lambda v: getattr(v, "name", str(v)))
def test_bellman_shapes_other_curve_and_scheme_on_emulator(name, curve, scheme):
    ctx = native.Context(0, emu_library())
    _run(ctx, curve, name, scheme)
    ctx.close()


def test_hand_written_bytes_agree_with_the_python_writer():
    """The two independent writers (this file's and oracle/ir.py) produce the same `out` bytes for the shapes both can
    express — a check of the ORACLE's writer against the byte-level reading of the reference made here."""
    from oracle import ir
    curve = BN254
    mine = shapes(curve)
    progs = {
        "empty": ir.Prog(curve, [], [], return_count=0),
        "identity": ir.Prog(curve, [ir.Parameter(1, True)], [ir.Constraint([(0, 1)], [(1, 1)], [(-1, 1)])], return_count=1),
        "unordered_variables": ir.Prog(curve, [ir.Parameter(43, True), ir.Parameter(52, False)],
                                       [ir.Constraint([(0, 1)], [(43, 1), (52, 1)], [(-1, 1)]), ir.Constraint([(0, 1)], [(0, 1), (43, 1)], [(-2, 1)])],
                                       return_count=2),
    }
    for name, prog in progs.items():
        assert ir.serialize_prog(prog) == mine[name][0], name
    assert ir.serialize_witness({0: 1, 43: 3, 52: 4, -1: 7, -2: 4}) == witness_file(mine["unordered_variables"][1], curve.r)


def test_missing_witness_value_is_reported():
    """The reference panics with AssignmentMissing (zokrates_ark/src/lib.rs:69,111); the reader returns UNSATISFIED."""
    data, wit, _, _, _ = shapes(BN254)["unordered_variables"]
    p = native.Program(np.frombuffer(data, dtype=np.uint8), emu_library())
    short = dict(wit)
    del short[new(42)]
    with pytest.raises(native.ZkhipError) as e:
        p.assignment(np.frombuffer(witness_file(short, BN254.r), dtype=np.uint8))
    assert e.value.code == -5


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_bellman_shapes_on_gpu(name):
    ctx = native.Context(0)
    _run(ctx, BN254, name, "g16")
    _run(ctx, BN254, name, "gm17")
    ctx.close()
