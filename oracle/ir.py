"""ZoKrates' compiled-program file (`out`) and the ark variable order, python restatement.  TEST ORACLE ONLY.

  * writer of the `out` format: /root/reference/zokrates_ast/src/ir/serialize.rs:133-148 (header layout),
    :202-279 (`ProgIterator::serialize`: 4 sections — parameters, constraints as a stream of CBOR items, solvers,
    module map; the header region is `size_of::<ProgHeader>()` bytes, only the first 100 of which are written; the
    module-map section is tagged `SectionType::Solvers`, :247) with [UPSTREAM] serde_cbor 0.11.2's encoding of the
    derived `Serialize` impls (structs = maps keyed by field name, enums externally tagged, `None` = null);
  * `ark_order`: /root/reference/zokrates_ark/src/lib.rs:80-129 (`Computation::generate_constraints`), SURVEY.md App. A.2;
  * `public_inputs_values`: /root/reference/zokrates_ast/src/ir/mod.rs:278-288;
  * the `witness` file writer: /root/reference/zokrates_ast/src/ir/witness.rs:44-53.

The reference holds no `out` fixture and cannot be compiled here, so this writer is the only producer of test inputs
for `zkhip_prog_parse`: the reader (C++) and this writer are two independent readings of the same source files.
"""
import hashlib
import struct
from dataclasses import dataclass, field

HEADER_REGION = 120          # size_of::<ProgHeader>() on x86-64: 20 bytes of scalars padded to 24 + 4 x 24-byte sections


# ---------------- minimal CBOR encoder (the subset serde_cbor emits) ----------------
def _head(major, n):
    if n < 24: return bytes([major << 5 | n])
    if n < 1 << 8: return bytes([major << 5 | 24, n])
    if n < 1 << 16: return bytes([major << 5 | 25]) + struct.pack(">H", n)
    if n < 1 << 32: return bytes([major << 5 | 26]) + struct.pack(">I", n)
    return bytes([major << 5 | 27]) + struct.pack(">Q", n)


def cbor(x):
    if x is None: return b"\xf6"
    if x is True: return b"\xf5"
    if x is False: return b"\xf4"
    if isinstance(x, int): return _head(0, x) if x >= 0 else _head(1, -1 - x)
    if isinstance(x, bytes): return _head(2, len(x)) + x
    if isinstance(x, str): return _head(3, len(x.encode())) + x.encode()
    if isinstance(x, (list, tuple)): return _head(4, len(x)) + b"".join(cbor(v) for v in x)
    if isinstance(x, dict): return _head(5, len(x)) + b"".join(cbor(k) + cbor(v) for k, v in x.items())
    raise TypeError(type(x))


# ---------------- the IR subset that matters to the prover ----------------
@dataclass
class Parameter:
    id: int                 # ZoKrates variable id: 0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1}  (flat/variable.rs:10-31)
    private: bool


@dataclass
class Constraint:
    left: list              # [(variable id, coefficient)]  stored order, duplicates allowed
    right: list
    lin: list
    span: object = None     # optional source span (ignored by the prover)
    error: object = None    # optional RuntimeError (ignored by the prover)


@dataclass
class Other:
    """A Directive or Log statement: opaque to the prover; `body` is any CBOR-able python value."""
    variant: str
    body: object


@dataclass
class Prog:
    curve: object
    arguments: list
    statements: list
    return_count: int = 0
    solvers: list = field(default_factory=list)


def curve_id_bytes(curve):
    """Field::id(): first 4 bytes of sha256(modulus little-endian)   (zokrates_field/src/lib.rs:283-293)."""
    nbytes = (curve.r.bit_length() + 63) // 64 * 8
    return hashlib.sha256(curve.r.to_bytes(nbytes, "little")).digest()[:4]


def _span(s):
    if s is None:
        return None
    module, (l0, c0), (l1, c1) = s
    return {"Source": {"module": module, "from": {"line": l0, "col": c0}, "to": {"line": l1, "col": c1}}}


def _lc(terms, span=None):
    return {"span": _span(span), "value": [[{"id": v}, int(c).to_bytes(32, "little")] for v, c in terms]}


def _statement(s):
    if isinstance(s, Constraint):
        return {"Constraint": {"span": _span(s.span), "quad": {"span": _span(s.span), "left": _lc(s.left, s.span), "right": _lc(s.right)},
                               "lin": _lc(s.lin), "error": s.error}}
    return {s.variant: s.body}


def serialize_prog(prog):
    """Bytes of the `out` file."""
    params = cbor([{"span": None, "id": {"id": p.id}, "private": p.private} for p in prog.arguments])
    stmts = b"".join(cbor(_statement(s)) for s in prog.statements)
    solvers = cbor(prog.solvers)
    modules = cbor({"modules": {}})
    body = params + stmts + solvers + modules
    off = HEADER_REGION
    secs = []
    for ty, blob in ((1, params), (2, stmts), (3, solvers), (3, modules)):
        secs.append(struct.pack("<IQQ", ty, off, len(blob)))
        off += len(blob)
    count = sum(isinstance(s, Constraint) for s in prog.statements)
    header = b"ZOK\0" + bytes([3, 0, 0, 0]) + curve_id_bytes(prog.curve) + struct.pack("<II", count, prog.return_count) + b"".join(secs)
    return header + b"\0" * (HEADER_REGION - len(header)) + body


def serialize_witness(values):
    """{variable id: value} -> the `witness` file (BTreeMap order = ascending signed id)."""
    out = struct.pack("<Q", len(values))
    for vid in sorted(values):
        out += struct.pack("<q", vid) + int(values[vid]).to_bytes(32, "little")
    return out


# ---------------- Computation::generate_constraints, restated ----------------
def ark_order(prog):
    """Returns (l, w, order, rows): order[j] = ZoKrates id of column j; rows[k] = [{col: coeff}] for k = A, B, C
    (duplicates summed, zero coefficients dropped)."""
    r = prog.curve.r
    inst, wit, sym = [0], [], {0: ("i", 0)}
    for p in prog.arguments:
        if p.private:
            sym[p.id] = ("w", len(wit)); wit.append(p.id)
        else:
            sym[p.id] = ("i", len(inst)); inst.append(p.id)
    raw = ([], [], [])
    for s in prog.statements:
        if not isinstance(s, Constraint):
            continue
        for k, lc in enumerate((s.left, s.right, s.lin)):
            row = {}
            for v, c in lc:
                if v not in sym:
                    if v < 0:
                        sym[v] = ("i", len(inst)); inst.append(v)
                    else:
                        sym[v] = ("w", len(wit)); wit.append(v)
                row[sym[v]] = (row.get(sym[v], 0) + c) % r
            raw[k].append(row)
    l, w = len(inst), len(wit)
    colof = lambda t: t[1] if t[0] == "i" else l + t[1]
    rows = tuple([{colof(t): c for t, c in row.items() if c} for row in mat] for mat in raw)
    return l, w, inst + wit, rows


def public_inputs_values(prog, witness):
    outs = sorted((v for v in witness if v < 0), reverse=True)          # ~out_0 = -1, ~out_1 = -2, ...
    assert outs == [-(i + 1) for i in range(len(outs))]
    return [witness[p.id] for p in prog.arguments if not p.private] + [witness[v] for v in outs]
