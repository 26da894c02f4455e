"""File formats either side of the hot path (host side, product code; "next" rows N1/N4 of SURVEY.md §8f).

* iden3 `.r1cs` / `.wtns` exactly as ZoKrates' circom exporter writes them
  (/root/reference/zokrates_circom/src/r1cs.rs:130-231, witness.rs:27-104): a tool-neutral way to hand a constraint
  system and a witness to the prover.  Wire order = [ONE, outputs, public inputs, private], i.e. instance first — the
  column convention of `zkhip_r1cs_load`.
* ZoKrates' native `witness` file (/root/reference/zokrates_ast/src/ir/witness.rs:44-71).
* `proof.json` / `verification.key` as the CLI writes them (/root/reference/zokrates_proof_systems/src/tagged.rs:14-37,
  scheme/groth16.rs:11-25): "0x" + lower-case hex, big-endian, zero-padded to the field width
  (/root/reference/zokrates_ark/src/lib.rs:150-226); serde_json pretty printing = 2-space indent.
"""
import json
import struct

import numpy as np

CURVE_NAMES = {0: "bn128", 1: "bls12_381"}
FR_MODULUS = {
    0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    1: 52435875175126190479447740508185965837690552500527637822603658699938581184513,
}
FQ_BYTES = {0: 32, 1: 48}


class FormatError(ValueError):
    pass


def curve_of_modulus(prime_bytes):
    p = int.from_bytes(prime_bytes, "little")
    for cid, r in FR_MODULUS.items():
        if r == p:
            return cid
    raise FormatError("unsupported scalar field modulus")


# ------------------------------------------------------------------ .r1cs
class R1cs:
    """n constraints over n_wires wires; mats = [(rowptr u64[n+1], col u32[nnz], val u8[nnz*32])] x 3 (A, B, C)."""

    def __init__(self, curve_id, n_wires, n_pub_out, n_pub_in, n_prv_in, mats):
        self.curve_id, self.n_wires = curve_id, n_wires
        self.n_pub_out, self.n_pub_in, self.n_prv_in = n_pub_out, n_pub_in, n_prv_in
        self.mats = mats
        self.n = len(mats[0][0]) - 1
        self.l = 1 + n_pub_out + n_pub_in        # ONE + public wires: the instance part
        self.w = n_wires - self.l


def read_r1cs(data):
    data = bytes(data)
    if data[:4] != b"r1cs":
        raise FormatError("not an .r1cs file")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version != 1:
        raise FormatError("unsupported .r1cs version")
    pos, sections = 12, {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, pos)
        sections[typ] = (pos + 12, size)
        pos += 12 + size
    if 1 not in sections or 2 not in sections:
        raise FormatError(".r1cs: header or constraint section missing")
    hp, _ = sections[1]
    (fs,) = struct.unpack_from("<I", data, hp)
    if fs != 32:
        raise FormatError("field size must be 32 bytes")
    curve_id = curve_of_modulus(data[hp + 4:hp + 4 + fs])
    n_wires, n_pub_out, n_pub_in, n_prv_in, _n_labels, n_cons = struct.unpack_from("<IIIIQI", data, hp + 4 + fs)
    cp, csize = sections[2]
    end = cp + csize
    mats = [([0], [], bytearray()) for _ in range(3)]
    p = cp
    for _ in range(n_cons):
        for k in range(3):
            (cnt,) = struct.unpack_from("<I", data, p)
            p += 4
            rp, col, val = mats[k]
            for _t in range(cnt):
                (wire,) = struct.unpack_from("<I", data, p)
                if wire >= n_wires:
                    raise FormatError(".r1cs: wire index out of range")
                col.append(wire)
                val += data[p + 4:p + 4 + fs]
                p += 4 + fs
            rp.append(len(col))
    if p != end:
        raise FormatError(".r1cs: constraint section size mismatch")
    out = [(np.array(rp, dtype=np.uint64), np.array(col, dtype=np.uint32), np.frombuffer(bytes(val), dtype=np.uint8)) for rp, col, val in mats]
    return R1cs(curve_id, n_wires, n_pub_out, n_pub_in, n_prv_in, out)


def write_r1cs(curve_id, n_wires, n_pub_out, n_pub_in, n_prv_in, mats):
    """Inverse of read_r1cs; byte-for-byte what zokrates_circom::write_r1cs emits (3 sections: constraints, header, wire map)."""
    prime = FR_MODULUS[curve_id].to_bytes(32, "little")
    n = len(mats[0][0]) - 1
    body = bytearray()
    rps = [np.asarray(m[0], dtype=np.uint64) for m in mats]
    cols = [np.asarray(m[1], dtype=np.uint32) for m in mats]
    vals = [bytes(np.asarray(m[2], dtype=np.uint8)) for m in mats]
    for i in range(n):
        for k in range(3):
            a, b = int(rps[k][i]), int(rps[k][i + 1])
            body += struct.pack("<I", b - a)
            for q in range(a, b):
                body += struct.pack("<I", int(cols[k][q])) + vals[k][32 * q:32 * q + 32]
    out = bytearray(b"r1cs") + struct.pack("<II", 1, 3)
    out += struct.pack("<IQ", 2, len(body)) + body
    out += struct.pack("<IQ", 1, 64) + struct.pack("<I", 32) + prime + struct.pack("<IIIIQI", n_wires, n_pub_out, n_pub_in, n_prv_in, n_wires, n)
    out += struct.pack("<IQ", 3, 8 * n_wires) + b"".join(struct.pack("<Q", i) for i in range(n_wires))
    return bytes(out)


# ------------------------------------------------------------------ .wtns
def read_wtns(data):
    """Returns (curve_id, values uint8[n*32]) — canonical little-endian, in wire order."""
    data = bytes(data)
    if data[:4] != b"wtns":
        raise FormatError("not a .wtns file")
    version, nsec = struct.unpack_from("<II", data, 4)
    if version != 2:
        raise FormatError("unsupported .wtns version")
    pos, sections = 12, {}
    for _ in range(nsec):
        typ, size = struct.unpack_from("<IQ", data, pos)
        sections[typ] = (pos + 12, size)
        pos += 12 + size
    hp, _ = sections[1]
    (fs,) = struct.unpack_from("<I", data, hp)
    if fs != 32:
        raise FormatError("field size must be 32 bytes")
    curve_id = curve_of_modulus(data[hp + 4:hp + 4 + fs])
    (count,) = struct.unpack_from("<I", data, hp + 4 + fs)
    wp, wsize = sections[2]
    if wsize != count * fs:
        raise FormatError(".wtns: witness section size mismatch")
    return curve_id, np.frombuffer(data[wp:wp + wsize], dtype=np.uint8)


def write_wtns(curve_id, values):
    values = bytes(np.asarray(values, dtype=np.uint8))
    prime = FR_MODULUS[curve_id].to_bytes(32, "little")
    count = len(values) // 32
    return (b"wtns" + struct.pack("<II", 2, 2) + struct.pack("<IQ", 1, 40) + struct.pack("<I", 32) + prime + struct.pack("<I", count)
            + struct.pack("<IQ", 2, len(values)) + values)


# ------------------------------------------------------------------ ZoKrates `witness`
def read_zokrates_witness(data):
    """{variable id (0 = ~one, k > 0 = _{k-1}, -k = ~out_{k-1}): int value}  (ir/witness.rs:55-71, flat/variable.rs:42-49)."""
    data = bytes(data)
    (count,) = struct.unpack_from("<Q", data, 0)
    if len(data) != 8 + 40 * count:
        raise FormatError("witness file size mismatch")
    out = {}
    for i in range(count):
        (vid,) = struct.unpack_from("<q", data, 8 + 40 * i)
        out[vid] = int.from_bytes(data[16 + 40 * i:48 + 40 * i], "little")
    return out


# ------------------------------------------------------------------ JSON artefacts
def _hex_be(le_bytes):
    return "0x" + bytes(le_bytes)[::-1].hex()


def _g1(buf, nb):
    return [_hex_be(buf[:nb]), _hex_be(buf[nb:2 * nb])]


def _g2(buf, nb):
    return [[_hex_be(buf[:nb]), _hex_be(buf[nb:2 * nb])], [_hex_be(buf[2 * nb:3 * nb]), _hex_be(buf[3 * nb:4 * nb])]]


def proof_json(curve_id, raw_proof, inputs, scheme="g16"):
    """raw_proof: the 8*sz(Fq)+3 bytes of zkhip_prove_g16 / zkhip_prove_gm17; inputs: public values (ints).  The text of
    `proof.json` (both schemes have proof points a (G1), b (G2), c (G1): scheme/groth16.rs:8-16, scheme/gm17.rs:12-17)."""
    nb = FQ_BYTES[curve_id]
    raw = bytearray(raw_proof)
    # A point at infinity arrives as all-zero coordinates plus its flag byte; the reference prints ark's `zero()`, which is
    # (x, y) = (0, 1) (G2: y = 1 + 0u), through parse_g1 / parse_g2 (zokrates_ark/src/lib.rs:150-218).
    one = (1).to_bytes(nb, "little")
    for flag, y_at in ((8 * nb, nb), (8 * nb + 1, 4 * nb), (8 * nb + 2, 7 * nb)):
        if len(raw) > flag and raw[flag]:
            raw[y_at:y_at + nb] = one
    raw = bytes(raw)
    doc = {"scheme": scheme, "curve": CURVE_NAMES[curve_id],
           "proof": {"a": _g1(raw[0:2 * nb], nb), "b": _g2(raw[2 * nb:6 * nb], nb), "c": _g1(raw[6 * nb:8 * nb], nb)},
           "inputs": ["0x" + int(v).to_bytes(32, "big").hex() for v in inputs]}
    return json.dumps(doc, indent=2)


def _strip_flags(pt):
    pt = bytearray(pt)
    pt[-1] &= 0x3f
    return bytes(pt)


def verification_key_json(curve_id, pk_bytes):
    """`verification.key` from the vk that leads an ark proving key (alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1)."""
    nb = FQ_BYTES[curve_id]
    pk = bytes(pk_bytes)
    g1, g2 = 2 * nb, 4 * nb
    alpha, beta, gamma, delta = pk[:g1], pk[g1:g1 + g2], pk[g1 + g2:g1 + 2 * g2], pk[g1 + 2 * g2:g1 + 3 * g2]
    off = g1 + 3 * g2
    (n_abc,) = struct.unpack_from("<Q", pk, off)
    abc = [pk[off + 8 + i * g1:off + 8 + (i + 1) * g1] for i in range(n_abc)]
    doc = {"scheme": "g16", "curve": CURVE_NAMES[curve_id], "alpha": _g1(_strip_flags(alpha), nb), "beta": _g2(_strip_flags(beta), nb),
           "gamma": _g2(_strip_flags(gamma), nb), "delta": _g2(_strip_flags(delta), nb), "gamma_abc": [_g1(_strip_flags(p), nb) for p in abc]}
    return json.dumps(doc, indent=2)


def verification_key_json_gm17(curve_id, pk_bytes):
    """`verification.key` from the vk that leads an ark-gm17 proving key (h_g2, g_alpha_g1, h_beta_g2, g_gamma_g1,
    h_gamma_g2, query[]) with the field names of /root/reference/zokrates_proof_systems/src/scheme/gm17.rs:19-27."""
    nb = FQ_BYTES[curve_id]
    g1, g2 = 2 * nb, 4 * nb
    pk = bytes(pk_bytes[:3 * g2 + 2 * g1 + 8])
    h, pos = pk[:g2], g2
    g_alpha, pos = pk[pos:pos + g1], pos + g1
    h_beta, pos = pk[pos:pos + g2], pos + g2
    g_gamma, pos = pk[pos:pos + g1], pos + g1
    h_gamma, pos = pk[pos:pos + g2], pos + g2
    (nq,) = struct.unpack_from("<Q", pk, pos)
    q = bytes(pk_bytes[pos + 8:pos + 8 + nq * g1])
    doc = {"scheme": "gm17", "curve": CURVE_NAMES[curve_id], "h": _g2(_strip_flags(h), nb), "g_alpha": _g1(_strip_flags(g_alpha), nb),
           "h_beta": _g2(_strip_flags(h_beta), nb), "g_gamma": _g1(_strip_flags(g_gamma), nb), "h_gamma": _g2(_strip_flags(h_gamma), nb),
           "query": [_g1(_strip_flags(q[i * g1:(i + 1) * g1]), nb) for i in range(nq)]}
    return json.dumps(doc, indent=2)
