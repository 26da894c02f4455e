"""`zkhip_prog_parse` fed by the THIRD writer (tests/serde_model.py: schema tables transcribed from the reference's serde
derives + a generic serde_cbor encoder): byte-equality with oracle/ir.py's writer on the shapes both express, and programs
with the statements only the tables express faithfully — `Directive` with every solver shape (`Bits(n)`, `Ref(RefCall {index,
signature})`, unit variants), `Log` with typed expressions, `Span::Embed` — which `Computation::generate_constraints` walks
past (/root/reference/zokrates_ark/src/lib.rs:115-123: only `Statement::Constraint` allocates variables and adds rows)."""
import random

import pytest

import serde_model as sm
from oracle import ir
from oracle.fields import BN254, BLS12_381
from zokrates_amd import native

from test_ingest import rows_of

CURVES = [BN254, BLS12_381]


@pytest.fixture(scope="module")
def lib():
    from emu_util import emu_library
    return emu_library()


def _span(rnd):
    k = rnd.random()
    if k < 0.4:
        return None
    if k < 0.8:
        return ("Source", {"module": rnd.getrandbits(64), "from": {"line": rnd.randrange(1, 500), "col": rnd.randrange(1, 120)},
                           "to": {"line": rnd.randrange(1, 500), "col": rnd.randrange(1, 120)}})
    return ("Embed", rnd.choice(["Unpack", "U32ToBits", "BitArrayLe"]))


def _ir_span(s):
    if s is None:
        return None
    if s[0] == "Embed":
        return {"Embed": s[1]}
    return {"Source": s[1]}


def _lc(rnd, curve, pool, span=None):
    r = curve.r
    terms = [(rnd.choice(pool), rnd.choice([0, 1, r - 1, rnd.randrange(r), rnd.randrange(1 << 32)])) for _ in range(rnd.randrange(0, 5))]
    return {"span": span, "value": [({"id": v}, c) for v, c in terms]}, terms


def _program(rnd, curve, n):
    n_args = rnd.randrange(0, 5)
    args = [{"span": _span(rnd), "id": {"id": k + 1}, "private": rnd.random() < 0.5} for k in range(n_args)]
    pool = [0] + [k + 1 for k in range(n_args)] + [-1, -2] + list(range(n_args + 1, n_args + 2 + 2 * n))
    model, plain = [], []                  # the schema-driven statements and the (left, right, lin) term lists of the constraints
    for k in range(n):
        if rnd.random() < 0.5:
            solver = rnd.choice(["ConditionEq", "Div", "Xor", "ShaCh", "EuclideanDiv", ("Bits", rnd.choice([8, 32, 254])),
                                 ("Ref", {"index": k, "signature": (2, 1)}), ("SnarkVerifyBls12377", 3)])
            ql, _ = _lc(rnd, curve, pool)
            qr, _ = _lc(rnd, curve, pool)
            model.append(("Directive", {"span": _span(rnd), "inputs": [{"span": None, "left": ql, "right": qr}],
                                        "outputs": [{"id": rnd.choice(pool)} for _ in range(rnd.randrange(0, 3))], "solver": solver}))
        if rnd.random() < 0.25:
            e, _ = _lc(rnd, curve, pool)
            model.append(("Log", {"span": None, "format_string": {"parts": ["value = ", " and ", ""]},
                                  "expressions": [(("type", "field"), [e]), (("type", "u32"), [])]}))
        sp = _span(rnd)
        (l, lt), (r_, rt), (c, ct) = _lc(rnd, curve, pool, sp), _lc(rnd, curve, pool), _lc(rnd, curve, pool)
        err = rnd.choice([None, "ArkConstraint", "Bitness", "Division"])
        model.append(("Constraint", {"span": sp, "quad": {"span": sp, "left": l, "right": r_}, "lin": c, "error": err}))
        plain.append((lt, rt, ct, sp, err))
    return args, model, plain


@pytest.mark.parametrize("curve", CURVES, ids=lambda c: c.name)
def test_third_writer_against_the_reader_and_the_other_writer(lib, curve):
    rnd = random.Random(2024)
    for trial in range(10):
        args, model, plain = _program(rnd, curve, rnd.randrange(0, 30))
        data = sm.program_file(curve.r, args, model, return_count=2)
        # what ark would build: constraints only, in order (python restatement of generate_constraints)
        prog = ir.Prog(curve, [ir.Parameter(a["id"]["id"], a["private"]) for a in args],
                       [ir.Constraint(l, r, c) for l, r, c, _, _ in plain], return_count=2)
        l, w, order, rows = ir.ark_order(prog)
        p = native.Program(data, lib)
        assert (p.curve_id, p.n, p.l, p.w, p.return_count) == (curve.curve_id, len(plain), l, w, 2)
        assert list(p.variable_order()) == order
        assert rows_of(p.mats(), p.n) == rows
        p.close()
        # on the constraint-only projection the two writers must agree byte for byte (spans and unit-variant errors included)
        only = [s for s in model if s[0] == "Constraint"]
        prog2 = ir.Prog(curve, [ir.Parameter(a["id"]["id"], a["private"]) for a in args], [], return_count=2)
        body_a = sm.program_file(curve.r, [dict(a, span=None) for a in args], only, return_count=2)
        prog2.statements = [ir.Constraint(l, r, c, span=None, error=err) for l, r, c, sp, err in plain]
        # (oracle/ir.py writes a constraint's span into the statement and its quad; give it the same spans through its dict hook)
        stmts_b = b"".join(ir.cbor({"Constraint": {"span": _ir_span(sp), "quad": {"span": _ir_span(sp), "left": ir._lc(l) | {"span": _ir_span(sp)},
                                                                                   "right": ir._lc(r)}, "lin": ir._lc(c), "error": err}})
                           for l, r, c, sp, err in plain)
        params_b = ir.cbor([{"span": None, "id": {"id": a["id"]["id"]}, "private": a["private"]} for a in args])
        assert body_a[120:120 + len(params_b)] == params_b
        assert body_a[120 + len(params_b):120 + len(params_b) + len(stmts_b)] == stmts_b


def test_schema_tables_match_the_hand_assembled_bytes():
    """One constraint, every byte spelled out (serde_cbor: map heads a1..a4, text keys, null = f6, 32-byte strings = 58 20)."""
    one = (1).to_bytes(32, "little")
    stmt = ("Constraint", {"span": None, "quad": {"span": None, "left": {"span": None, "value": [({"id": 1}, 1)]},
                                                  "right": {"span": None, "value": [({"id": -1}, 1)]}},
                           "lin": {"span": None, "value": []}, "error": "ArkConstraint"})
    var = lambda idb: b"\xa1" + b"\x62id" + idb
    lc = lambda body: b"\xa2" + b"\x64span\xf6" + b"\x65value" + body
    want = (b"\xa1" + b"\x6aConstraint" + b"\xa4" + b"\x64span\xf6" + b"\x64quad" + b"\xa3" + b"\x64span\xf6"
            + b"\x64left" + lc(b"\x81\x82" + var(b"\x01") + b"\x58\x20" + one)
            + b"\x65right" + lc(b"\x81\x82" + var(b"\x20") + b"\x58\x20" + one)
            + b"\x63lin" + lc(b"\x80") + b"\x65error" + b"\x6dArkConstraint")
    assert sm.encode(sm.Statement, stmt) == want
    assert sm.encode(sm.Solver, ("Bits", 254)) == b"\xa1\x64Bits\x18\xfe" and sm.encode(sm.Solver, "Div") == b"\x63Div"
    assert sm.encode(sm.RefCall, {"index": 3, "signature": (2, 1)}) == b"\xa2\x65index\x03\x69signature\x82\x02\x01"
    assert sm.encode(sm.ConcreteType, ("type", "field")) == b"\xa1\x64type\x65field"
