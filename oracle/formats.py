"""Byte formats on the boundary, python restatement.  TEST ORACLE ONLY.

  * ark ``ProvingKey::serialize_unchecked`` / ``deserialize_unchecked`` as used at
    /root/reference/zokrates_ark/src/groth16.rs:40-42,97-98  ([UPSTREAM] layout)
  * proof point -> hex strings: /root/reference/zokrates_ark/src/lib.rs:150-226
  * proof.json envelope: /root/reference/zokrates_proof_systems/src/tagged.rs:14-37,
    /root/reference/zokrates_proof_systems/src/scheme/groth16.rs:8-16
  * zkhip C-ABI raw proof encoding (include/zkhip.h: ``zkhip_prove_g16``)
"""
import json
import struct

INF_FLAG = 0x40  # ark-serialize 0.3 SWFlags::Infinity = bit 6 of the last byte


def _fq_le(v, nb): return int(v).to_bytes(nb, "little")


def ser_g1(curve, P):
    nb = curve.fq_bytes
    if P is None:
        b = bytearray(_fq_le(0, nb) + _fq_le(1, nb))
        b[-1] |= INF_FLAG
        return bytes(b)
    return _fq_le(P[0], nb) + _fq_le(P[1], nb)


def ser_g2(curve, P):
    nb = curve.fq_bytes
    if P is None:
        b = bytearray(_fq_le(0, nb) * 2 + _fq_le(1, nb) + _fq_le(0, nb))
        b[-1] |= INF_FLAG
        return bytes(b)
    (x0, x1), (y0, y1) = P
    return _fq_le(x0, nb) + _fq_le(x1, nb) + _fq_le(y0, nb) + _fq_le(y1, nb)


def ser_vec(items, f):
    return struct.pack("<Q", len(items)) + b"".join(f(x) for x in items)


def ark_pk_serialize(curve, pk):
    g1 = lambda P: ser_g1(curve, P)
    g2 = lambda P: ser_g2(curve, P)
    vk = pk["vk"]
    out = g1(vk["alpha_g1"]) + g2(vk["beta_g2"]) + g2(vk["gamma_g2"]) + g2(vk["delta_g2"])
    out += ser_vec(vk["gamma_abc_g1"], g1)
    out += g1(pk["beta_g1"]) + g1(pk["delta_g1"])
    out += ser_vec(pk["a_query"], g1) + ser_vec(pk["b_g1_query"], g1) + ser_vec(pk["b_g2_query"], g2)
    out += ser_vec(pk["h_query"], g1) + ser_vec(pk["l_query"], g1)
    return out


class _Rd:
    def __init__(self, b): self.b = memoryview(b); self.o = 0
    def take(self, n):
        if self.o + n > len(self.b): raise ValueError("truncated proving key")
        v = self.b[self.o:self.o + n]; self.o += n
        return bytes(v)


def _de_fq_flag(rd, nb):
    raw = bytearray(rd.take(nb))
    flag = raw[-1] & 0xC0
    raw[-1] &= 0x3F
    return int.from_bytes(raw, "little"), flag


def de_g1(curve, rd):
    nb = curve.fq_bytes
    x = int.from_bytes(rd.take(nb), "little")
    y, flag = _de_fq_flag(rd, nb)
    return None if flag & INF_FLAG else (x, y)


def de_g2(curve, rd):
    nb = curve.fq_bytes
    x0 = int.from_bytes(rd.take(nb), "little")
    x1 = int.from_bytes(rd.take(nb), "little")
    y0 = int.from_bytes(rd.take(nb), "little")
    y1, flag = _de_fq_flag(rd, nb)
    return None if flag & INF_FLAG else ((x0, x1), (y0, y1))


def de_vec(rd, f):
    (n,) = struct.unpack("<Q", rd.take(8))
    return [f() for _ in range(n)]


def ark_pk_deserialize(curve, data):
    rd = _Rd(data)
    g1 = lambda: de_g1(curve, rd)
    g2 = lambda: de_g2(curve, rd)
    vk = dict(alpha_g1=g1(), beta_g2=g2(), gamma_g2=g2(), delta_g2=g2())
    vk["gamma_abc_g1"] = de_vec(rd, g1)
    pk = dict(vk=vk, beta_g1=g1(), delta_g1=g1())
    pk["a_query"] = de_vec(rd, g1)
    pk["b_g1_query"] = de_vec(rd, g1)
    pk["b_g2_query"] = de_vec(rd, g2)
    pk["h_query"] = de_vec(rd, g1)
    pk["l_query"] = de_vec(rd, g1)
    if rd.o != len(data):
        raise ValueError("trailing bytes in proving key")
    return pk


# ---- raw proof encoding of the C ABI: 8 Fq canonical LE + 3 infinity flag bytes ----
def proof_raw(curve, proof):
    nb = curve.fq_bytes
    A, B, C = proof
    z1 = (0, 0); z2 = ((0, 0), (0, 0))
    a = A or z1; b = B or z2; c = C or z1
    out = _fq_le(a[0], nb) + _fq_le(a[1], nb)
    out += _fq_le(b[0][0], nb) + _fq_le(b[0][1], nb) + _fq_le(b[1][0], nb) + _fq_le(b[1][1], nb)
    out += _fq_le(c[0], nb) + _fq_le(c[1], nb)
    out += bytes([A is None, B is None, C is None])
    return out


def proof_from_raw(curve, raw):
    nb = curve.fq_bytes
    assert len(raw) == 8 * nb + 3
    v = [int.from_bytes(raw[i * nb:(i + 1) * nb], "little") for i in range(8)]
    fa, fb, fc = raw[8 * nb:]
    A = None if fa else (v[0], v[1])
    B = None if fb else ((v[2], v[3]), (v[4], v[5]))
    C = None if fc else (v[6], v[7])
    return A, B, C


# ---- proof.json ----
def hex_be(v, nb): return "0x" + int(v).to_bytes(nb, "big").hex()


def proof_json(curve, proof, inputs, scheme="g16"):
    """Byte-for-byte what ``serde_json::to_string_pretty(TaggedProof)`` emits
    (/root/reference/zokrates_cli/src/ops/generate_proof.rs:188-201)."""
    nb = curve.fq_bytes
    A, B, C = proof
    a = A or (0, 0); b = B or ((0, 0), (0, 0)); c = C or (0, 0)
    doc = {
        "scheme": scheme,
        "curve": curve.name,
        "proof": {
            "a": [hex_be(a[0], nb), hex_be(a[1], nb)],
            "b": [[hex_be(b[0][0], nb), hex_be(b[0][1], nb)], [hex_be(b[1][0], nb), hex_be(b[1][1], nb)]],
            "c": [hex_be(c[0], nb), hex_be(c[1], nb)],
        },
        "inputs": [hex_be(x, curve.fr_bytes) for x in inputs],
    }
    return json.dumps(doc, indent=2)
