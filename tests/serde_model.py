"""A third, table-driven writer of ZoKrates' `out` program file — TEST INFRASTRUCTURE.

`zkhip_prog_parse` (csrc/ingest.hip) is checked against bytes from three writers that share no code: oracle/ir.py (hand-built
python dictionaries), tests/test_ingest_reference_shapes.py (bytes assembled by hand) and this one: a generic encoder of
[UPSTREAM] serde_cbor 0.11.2's self-describing format driven by SCHEMA TABLES that transcribe the reference's `derive(Serialize)`
declarations one to one — so the knowledge "which fields, in which order, tagged how" lives in tables that can be diffed against
the Rust sources line by line, not in encoder code.  (The reference holds no `out` fixture and cannot be compiled here.)

serde's data model -> serde_cbor (non-packed, as `serde_cbor::to_writer` is called in serialize.rs:217,244,257,268):
  struct            map, text keys = field names, declaration order          unit variant      text string = variant name
  newtype variant   {variant: value}                                         tuple / Vec       definite-length array
  Option            null | the value                                         usize / isize     shortest-form integer
  bytes             byte string (field elements: zokrates_field/src/lib.rs:547-560, 32 bytes little-endian canonical)

Tables (reference file:line of the declaration they transcribe):
  Variable           zokrates_ast/src/common/flat/variable.rs:10-13       Parameter   common/flat/parameter.rs:9-16
  LinComb, QuadComb  zokrates_ast/src/ir/expression.rs:69-76, 10-18       Position, SourceSpan, Span   common/position.rs:53-104
  ConstraintStatement, Statement   zokrates_ast/src/ir/mod.rs:34-41, 118-128 (`Block` is #[serde(skip)])
  DirectiveStatement, LogStatement common/statements.rs:155-163, 110-118   RefCall, Solver   common/solvers.rs:5-28
  FormatString       common/format_string.rs:5-8                          RuntimeError (unit variants used)   common/error.rs:7-
  ConcreteType       typed/types.rs:694-731 (custom impl: {"type": "field" | "bool" | "u32" ...})
  FlatEmbed          common/embed.rs:39-56
"""
import hashlib
import struct

# ---------------- type constructors ----------------
USIZE, ISIZE, U64, BOOL, STR, FIELD = "usize", "isize", "u64", "bool", "str", "field"


def Struct(*fields):
    return ("struct", fields)


def Enum(**variants):
    """variant -> None (unit) | a type (newtype variant)."""
    return ("enum", variants)


def Opt(t):
    return ("opt", t)


def Vec(t):
    return ("vec", t)


def Tuple(*ts):
    return ("tuple", ts)


# ---------------- the schema tables ----------------
Variable = Struct(("id", ISIZE))
Position = Struct(("line", USIZE), ("col", USIZE))
SourceSpan = Struct(("module", U64), ("from", Position), ("to", Position))
FlatEmbed = Enum(FieldToBoolUnsafe=None, BitArrayLe=None, Unpack=None, U8ToBits=None, U16ToBits=None, U32ToBits=None, U64ToBits=None,
                 U8FromBits=None, U16FromBits=None, U32FromBits=None, U64FromBits=None, SnarkVerifyBls12377=None)
Span = Enum(Source=SourceSpan, Embed=FlatEmbed)
Parameter = Struct(("span", Opt(Span)), ("id", Variable), ("private", BOOL))
LinComb = Struct(("span", Opt(Span)), ("value", Vec(Tuple(Variable, FIELD))))
QuadComb = Struct(("span", Opt(Span)), ("left", LinComb), ("right", LinComb))
RuntimeError = Enum(BellmanConstraint=None, ArkConstraint=None, ArkOneBinding=None, ArkInputBinding=None, Bitness=None, Sum=None, Equal=None,
                    Le=None, BranchIsolation=None, Or=None, Xor=None, Inverse=None, Euclidean=None, Division=None)
ConstraintStatement = Struct(("span", Opt(Span)), ("quad", QuadComb), ("lin", LinComb), ("error", Opt(RuntimeError)))
RefCall = Struct(("index", USIZE), ("signature", Tuple(USIZE, USIZE)))
Solver = Enum(ConditionEq=None, Bits=USIZE, Div=None, Xor=None, Or=None, ShaAndXorAndXorAnd=None, ShaCh=None, EuclideanDiv=None, Ref=RefCall,
              SnarkVerifyBls12377=USIZE)
DirectiveStatement = Struct(("span", Opt(Span)), ("inputs", Vec(QuadComb)), ("outputs", Vec(Variable)), ("solver", Solver))
FormatString = Struct(("parts", Vec(STR)))
ConcreteType = Enum(type=STR)                                  # GType's hand-written impl: newtype variant "type" around the type name
LogStatement = Struct(("span", Opt(Span)), ("format_string", FormatString), ("expressions", Vec(Tuple(ConcreteType, Vec(LinComb)))))
Statement = Enum(Constraint=ConstraintStatement, Directive=DirectiveStatement, Log=LogStatement)


# ---------------- the generic encoder ----------------
def _head(major, n):
    if n < 24:
        return bytes([major << 5 | n])
    for ai, fmt, lim in ((24, ">B", 1 << 8), (25, ">H", 1 << 16), (26, ">I", 1 << 32), (27, ">Q", 1 << 64)):
        if n < lim:
            return bytes([major << 5 | ai]) + struct.pack(fmt, n)
    raise ValueError(n)


def encode(t, v):
    """Value `v` of schema type `t`: structs are dicts (or tuples in field order), enum values are "Variant" or ("Variant", payload)."""
    if t in (USIZE, U64):
        assert v >= 0
        return _head(0, v)
    if t == ISIZE:
        return _head(0, v) if v >= 0 else _head(1, -1 - v)
    if t == BOOL:
        return b"\xf5" if v else b"\xf4"
    if t == STR:
        b = v.encode()
        return _head(3, len(b)) + b
    if t == FIELD:
        return _head(2, 32) + int(v).to_bytes(32, "little")
    kind = t[0]
    if kind == "opt":
        return b"\xf6" if v is None else encode(t[1], v)
    if kind == "vec":
        return _head(4, len(v)) + b"".join(encode(t[1], x) for x in v)
    if kind == "tuple":
        assert len(v) == len(t[1])
        return _head(4, len(v)) + b"".join(encode(ti, x) for ti, x in zip(t[1], v))
    if kind == "struct":
        fields = t[1]
        vals = [v[name] for name, _ in fields] if isinstance(v, dict) else list(v)
        assert len(vals) == len(fields)
        return _head(5, len(fields)) + b"".join(encode(STR, name) + encode(ft, x) for (name, ft), x in zip(fields, vals))
    if kind == "enum":
        name, payload = (v, None) if isinstance(v, str) else v
        vt = t[1][name]
        if vt is None:
            return encode(STR, name)
        return _head(5, 1) + encode(STR, name) + encode(vt, payload)
    raise TypeError(t)


def field_id(modulus):
    """Field::id (zokrates_field/src/lib.rs:283-293): first 4 bytes of sha256 of the modulus' little-endian limbs."""
    nbytes = (modulus.bit_length() + 63) // 64 * 8
    return hashlib.sha256(modulus.to_bytes(nbytes, "little")).digest()[:4]


def program_file(modulus, arguments, statements, return_count, solvers=()):
    """`ProgIterator::serialize` (serialize.rs:202-279): 120-byte header region, then parameters, the statements one CBOR
    item after the other, the solver list, the module map (tagged as a second Solvers section, :247)."""
    params = encode(Vec(Parameter), arguments)
    stmts = b"".join(encode(Statement, s) for s in statements)
    solv = encode(Vec(Solver), list(solvers))
    modmap = _head(5, 1) + encode(STR, "modules") + _head(5, 0)
    off, secs = 120, b""
    for ty, blob in ((1, params), (2, stmts), (3, solv), (3, modmap)):
        secs += struct.pack("<IQQ", ty, off, len(blob))
        off += len(blob)
    count = sum(1 for s in statements if not isinstance(s, str) and s[0] == "Constraint")
    header = b"ZOK\0" + bytes([3, 0, 0, 0]) + field_id(modulus) + struct.pack("<II", count, return_count) + secs
    return header + b"\0" * (120 - len(header)) + params + stmts + solv + modmap
