"""SHA-256 workload (BASELINE.json configs[0]: "stdlib sha256/512bitPacked.zok, BN254") — host-side input generator,
product code like synth.py and poseidon.py.

The circuit is a restatement of what ZoKrates makes of
    def main(field[4] preimage) -> field[2]            /root/reference/zokrates_stdlib/stdlib/hashes/sha256/512bitPacked.zok:8-20
(unpack four 128-bit field elements, sha256 of the 512 bits with the padding block of 512bitPadded.zok:6-33, pack the digest
into two field elements), `hashes` independent calls side by side.  There is no compiler in this image, so the constraint
system is produced by following the reference's passes for exactly the operations shaRound.zok uses — nothing here is
statistical (synth.py's "sha" kind is), every wire is the wire of a SHA-256 computation, and tests/test_sha256_circuit.py checks
the public outputs against hashlib:

* u32 values are lazy sums with a tracked maximum (zokrates_analysis/src/uint_optimizer.rs:153-190: an addition only adds the
  maxima, nothing is reduced at a definition, :443-446); a value is decomposed when a bitwise operation needs its bits
  (`force_reduce`, :250-270, :330-380) — `max(bitlen(max), 32)` boolean constraints b*b = b and one sum check
  1 * (sum 2^i b_i) = value (zokrates_codegen/src/lib.rs:1946-2003), once per variable (the bits cache, :1394, :2415-2419);
* xor costs one constraint per bit, (2x)*y = x + y - name (lib.rs:1220-1276; a constant bit costs nothing); shifts and the
  `(x >> n) | (x << 32-n)` rotation only move bits (lib.rs:1486-1537, :1810-1818);
* the two SHA-specific patterns: ch = (e & f) ^ (!e & g) as a*(b - c) = ch - c, one constraint per bit (lib.rs:1590-1631), and
  maj = (a & b) ^ (a & c) ^ (b & c) as b*c = bc, (2bc - b - c)*a = bc - maj, two per bit (lib.rs:1665-1713);
* `unpack128` is the embed's decomposition (zokrates_ast/src/common/embed.rs:560-640: bit checks from the last bit up, then
  input * 1 = sum), `pack128` a linear combination of the digest's bits;
* linear definitions do not survive the optimizer (zokrates_core/src/optimizer/redefinition.rs:24-40: `lin == k*v` becomes a
  substitution), so the lazy sums are INLINED into the sum checks: the value of `e` in round i is the combination of every
  wire it was ever added up from.  These are the widest rows any workload here has (up to 7 041 terms in C): the
  opposite corner of the sparse mat-vec from synth.py's two-term rows;
* sub-expressions over constants only (the IV in the first rounds, the whole message schedule of the padding block) are
  folded, as the reference's propagation does before flattening.
Variables in ark order (/root/reference/zokrates_ark/src/lib.rs:41-129): column 0 = ONE, the public inputs, the public outputs
`~out_i`, then the witness in the order the constraints first name them (a, b, c of each constraint in turn) — so that the
same system read back from an `out` file by `zkhip_prog_parse` has the same columns, and a key made for one fits the other.  What is NOT claimed: equality constraint for constraint with a compiled
`out` file (there is none to compare with); the count per hash is what the rules above give.
"""
import functools
import hashlib

import numpy as np

from .synth import CURVE_IDS, FR_MODULUS, splitmix64

K256 = [
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01, 0x243185be,
    0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa,
    0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967, 0x27b70a85,
    0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85, 0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3,
    0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f,
    0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2]     # FIPS 180-4 §4.2.2
IV256 = [0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19]            # §5.3.3
M32 = 0xffffffff

ONE = "one"
ZERO_BIT, ONE_BIT = -1, -2        # constant bits; witness indices are >= 0


def _acc(lc, bit, coeff):
    """lc += coeff * bit (a witness index or a constant bit)."""
    if bit >= 0:
        lc[bit] = lc.get(bit, 0) + coeff
    elif bit == ONE_BIT:
        lc[ONE] = lc.get(ONE, 0) + coeff


class _U32:
    """A u32 of the flattener: a constant, or a lazy field value (`lc`, maximum `max`) and / or its 32 bits, most significant
    first.  `lc` maps witness index / ("in", j) / ONE to an integer; `src` says how the tape finds the value."""
    __slots__ = ("const", "lc", "bits", "max", "src")

    def __init__(self, const=None, lc=None, bits=None, max_=None, src=None):
        self.const, self.lc, self.bits, self.max, self.src = const, lc, bits, max_, src


class _Flattener:
    """Allocates witness variables in order, collects the rows (A, B, C) and a tape that computes a witness.  The tape works on
    whole words: an entry produces the values of consecutive variables from earlier ones, for many hashes at once."""

    def __init__(self):
        self.nvar = 0
        self.rows = []
        self.tape = []

    def fresh(self, count):
        first = self.nvar
        self.nvar += count
        return first

    # -- values ---------------------------------------------------------------------------------------------------------
    def constant(self, v):
        v &= M32
        return _U32(const=v, max_=v, bits=[ONE_BIT if (v >> (31 - i)) & 1 else ZERO_BIT for i in range(32)])

    def from_bits(self, bits):
        """u32_from_bits: free, the bits are known (lib.rs:2741-2752), maximum 2^32 - 1 (uint_optimizer.rs:465-474)."""
        return _U32(bits=list(bits), max_=M32)

    def field_of(self, u):
        """The value as a linear combination: the lazy one if there is one (`get_field_unchecked`), else the sum of the bits."""
        if u.const is not None:
            return {ONE: u.const} if u.const else {}
        if u.lc is not None:
            return u.lc
        lc = {}
        for i, b in enumerate(u.bits):
            _acc(lc, b, 1 << (31 - i))
        return lc

    def add(self, a, b):
        if a.const is not None and b.const is not None:
            return self.constant(a.const + b.const)      # (propagation: wrapping u32 addition of constants)
        lc = dict(self.field_of(a))
        for k, v in self.field_of(b).items():
            lc[k] = lc.get(k, 0) + v
        return _U32(lc=lc, max_=a.max + b.max, src=("add", a, b))

    def bits_of(self, u):
        """The 32 bits a bitwise operation works on: known, or decomposed now (once: the object keeps them)."""
        if u.bits is not None:
            return u.bits
        width = max(u.max.bit_length(), 32)
        first = self.fresh(width)
        all_bits = list(range(first, first + width))
        for b in all_bits:
            self.rows.append(({b: 1}, {b: 1}, {b: 1}))
        self.rows.append(({ONE: 1}, {b: 1 << (width - 1 - i) for i, b in enumerate(all_bits)}, dict(u.lc)))
        self.tape.append(("bits", first, width, u))
        u.bits = all_bits[width - 32:]
        return u.bits

    # -- bitwise operations ---------------------------------------------------------------------------------------------------
    def xor(self, a, b):
        if a.const is not None and b.const is not None:
            return self.constant(a.const ^ b.const)
        ab, bb = self.bits_of(a), self.bits_of(b)
        out = []
        for x, y in zip(ab, bb):
            if x < 0 and y < 0:
                out.append(ONE_BIT if (x == ONE_BIT) != (y == ONE_BIT) else ZERO_BIT)
            elif x < 0 or y < 0:
                c, e = (x, y) if x < 0 else (y, x)
                # a constant 0 bit passes the other through (lib.rs:1238-1240); a constant 1 would be 1 - e (never met: the
                # only constant bits SHA-256 xors are the zeros a shift brought in)
                assert c == ZERO_BIT
                out.append(e)
            else:
                name = self.fresh(1)
                self.rows.append(({x: 2}, {y: 1}, {x: 1, y: 1, name: -1}))
                out.append(name)
        self.tape.append(("xor", out, ab, bb))
        return _U32(bits=out, max_=M32)

    def ch(self, e, f, g):
        if all(u.const is not None for u in (e, f, g)):
            return self.constant((e.const & f.const) ^ (~e.const & g.const))
        a, b, c = self.bits_of(e), self.bits_of(f), self.bits_of(g)
        out = []
        for x, y, z in zip(a, b, c):
            name = self.fresh(1)
            left, rhs, res = {}, {}, {name: 1}
            _acc(left, x, 1)
            _acc(rhs, y, 1)
            _acc(rhs, z, -1)
            _acc(res, z, -1)
            self.rows.append((left, rhs, res))
            out.append(name)
        self.tape.append(("ch", out, a, b, c))
        return _U32(bits=out, max_=M32)

    def maj(self, p, q, r):
        if all(u.const is not None for u in (p, q, r)):
            return self.constant((p.const & q.const) ^ (p.const & r.const) ^ (q.const & r.const))
        a, b, c = self.bits_of(p), self.bits_of(q), self.bits_of(r)
        out, bcs = [], []
        for x, y, z in zip(a, b, c):
            m = self.fresh(1)
            bc = self.fresh(1)
            l1, r1, l2, r2 = {}, {}, {bc: 2}, {}
            _acc(l1, y, 1)
            _acc(r1, z, 1)
            self.rows.append((l1, r1, {bc: 1}))
            _acc(l2, y, -1)
            _acc(l2, z, -1)
            _acc(r2, x, 1)
            self.rows.append((l2, r2, {bc: 1, m: -1}))
            out.append(m)
            bcs.append(bc)
        self.tape.append(("maj", out, bcs, a, b, c))
        return _U32(bits=out, max_=M32)

    def rotr(self, u, n):
        if u.const is not None:
            return self.constant((u.const >> n) | (u.const << (32 - n)))
        bits = self.bits_of(u)
        return _U32(bits=bits[32 - n:] + bits[:32 - n], max_=M32)

    def shr(self, u, n):
        if u.const is not None:
            return self.constant(u.const >> n)
        bits = self.bits_of(u)
        return _U32(bits=[ZERO_BIT] * n + bits[:32 - n], max_=M32 >> n)


def _sha_round(fl, block, current):
    """shaRound.zok:49-101, statement by statement."""
    w = list(block)
    for i in range(16, 64):
        x, y = w[i - 15], w[i - 2]
        s0 = fl.xor(fl.xor(fl.rotr(x, 7), fl.rotr(x, 18)), fl.shr(x, 3))
        s1 = fl.xor(fl.xor(fl.rotr(y, 17), fl.rotr(y, 19)), fl.shr(y, 10))
        w.append(fl.add(fl.add(fl.add(w[i - 16], s0), w[i - 7]), s1))
    a, b, c, d, e, f, g, h = current
    for i in range(64):
        ch = fl.ch(e, f, g)
        big_s1 = fl.xor(fl.xor(fl.rotr(e, 6), fl.rotr(e, 11)), fl.rotr(e, 25))
        t1 = fl.add(fl.add(fl.add(fl.add(h, big_s1), ch), fl.constant(K256[i])), w[i])
        maj = fl.maj(a, b, c)
        big_s0 = fl.xor(fl.xor(fl.rotr(a, 2), fl.rotr(a, 13)), fl.rotr(a, 22))
        t2 = fl.add(big_s0, maj)
        h, g, f, e, d, c, b, a = g, f, e, fl.add(d, t1), c, b, a, fl.add(t1, t2)
    return [fl.add(x, y) for x, y in zip(current, (a, b, c, d, e, f, g, h))]


@functools.lru_cache(maxsize=None)
def template():
    """One `512bitPacked` call with symbolic inputs ("in", 0..3) and outputs ("out", 0..1): (rows over the columns of
    generate_constraints, the tape over the flattener's own variable numbers, witness count, column of each tape variable)."""
    fl = _Flattener()
    words = []
    for j in range(4):                                   # unpack128: embed.rs:560-640
        first = fl.fresh(128)
        bits = list(range(first, first + 128))
        for b in reversed(bits):
            fl.rows.append(({b: 1}, {b: 1}, {b: 1}))
        fl.rows.append(({("in", j): 1}, {ONE: 1}, {b: 1 << (127 - i) for i, b in enumerate(bits)}))
        fl.tape.append(("unpack", first, j))
        words += [fl.from_bits(bits[32 * k:32 * k + 32]) for k in range(4)]
    state = _sha_round(fl, words, [fl.constant(v) for v in IV256])
    pad = [fl.constant(v) for v in [0x80000000] + [0] * 14 + [0x200]]     # 512bitPadded.zok:10-31
    digest = _sha_round(fl, pad, state)
    for j in range(2):                                   # pack128 of u32_to_bits of the (lazy) digest words; the return statement
        lc = {}
        for k in range(4):
            for i, b in enumerate(fl.bits_of(digest[4 * j + k])):
                _acc(lc, b, 1 << (127 - 32 * k - i))
        fl.rows.append(({ONE: 1}, lc, {("out", j): 1}))
    # columns in the order ark's generate_constraints allocates them (zokrates_ark/src/lib.rs:41-75, :112-119): a variable gets
    # its column when a constraint first names it, walking a, then b, then c of every constraint in program order — not when
    # the flattener created it (unpack128 checks its bits from the last one up; a maj bit is named by its bc product first).
    # Within one combination the terms are kept sorted by column, so new variables of one combination take consecutive columns.
    new_of = {}
    for row in fl.rows:
        for lc in row:
            for k in sorted(k for k in lc if isinstance(k, int) and k not in new_of):
                new_of[k] = len(new_of)
    assert len(new_of) == fl.nvar
    renamed = [tuple({(new_of[k] if isinstance(k, int) else k): v for k, v in lc.items()} for lc in row) for row in fl.rows]
    column_of = np.array([new_of[k] for k in range(fl.nvar)], dtype=np.int64)
    return renamed, fl.tape, fl.nvar, column_of


def sha256_packed(preimage):
    """The function the circuit computes, on integers: four values below 2^128 -> two (hashlib is the arithmetic)."""
    data = b"".join(int(v).to_bytes(16, "big") for v in preimage)
    d = hashlib.sha256(data).digest()
    return [int.from_bytes(d[:16], "big"), int.from_bytes(d[16:], "big")]


class Sha256Packed:
    """CSR matrices and assignments of `hashes` independent calls; same interface as synth.SynthCircuit."""

    def __init__(self, curve, hashes=1):
        self.curve_id = CURVE_IDS.get(curve, curve)
        self.hashes, self.kind = hashes, "sha256"
        p = self.p = FR_MODULUS[self.curve_id]
        rows, _, V, _ = template()
        H = hashes
        self.vars_per_hash = V
        self.l, self.w = 1 + 6 * H, V * H
        self.m = self.l + self.w
        self.n = len(rows) * H
        N = 1
        while N < self.n + self.l:
            N *= 2
        self.N = N
        base = self.l

        # template entries sorted by (class, index): ONE < inputs < outputs < witness is the column order of every hash
        def order(k):
            if isinstance(k, tuple):
                return (1 if k[0] == "in" else 2, k[1])
            return (0, 0) if k == ONE else (3, k)

        mats = []
        for which in range(3):
            ents = [sorted(((order(k), v % p) for k, v in row[which].items() if v % p), key=lambda t: t[0]) for row in rows]
            counts = np.array([len(e) for e in ents], dtype=np.uint64)
            cls = np.array([o[0] for e in ents for o, _ in e], dtype=np.int64)
            idx = np.array([o[1] for e in ents for o, _ in e], dtype=np.int64)
            vals = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for e in ents for _, v in e), dtype=np.uint8).reshape(-1, 32)
            h = np.arange(H, dtype=np.int64)[:, None]
            col = np.where(cls == 0, 0, np.where(cls == 1, 1 + 4 * h + idx, np.where(cls == 2, 1 + 4 * H + 2 * h + idx, base + V * h + idx)))
            rp = np.zeros(self.n + 1, dtype=np.uint64)
            np.cumsum(np.tile(counts, H), out=rp[1:])
            mats.append((rp, np.ascontiguousarray(col.reshape(-1), dtype=np.uint32), np.ascontiguousarray(np.tile(vals, (H, 1))).reshape(-1)))
        self.A, self.B, self.C = mats

    def mats(self):
        return [self.A, self.B, self.C]

    def witness(self, preimages):
        """The witness block of every hash, (hashes, vars_per_hash) int64 (every wire is a bit), for preimages[h] = four
        integers below 2^128: the tape of the template, run for all hashes at once."""
        _, tape, V, column_of = template()
        H = len(preimages)
        Z = np.zeros((V + 2, H), dtype=np.int64)         # rows V, V + 1: the constant bits 0 and 1
        Z[V + 1] = 1
        ix = lambda bits: np.array([b if b >= 0 else (V if b == ZERO_BIT else V + 1) for b in bits], dtype=np.int64)
        memo = {}

        def value(u):
            """The lazy (unreduced) value of a u32, as the prover's solver sees it."""
            if u.const is not None:
                return np.full(H, u.const, dtype=np.int64)
            k = id(u)
            if k not in memo:
                if u.lc is not None:
                    memo[k] = value(u.src[1]) + value(u.src[2])
                else:
                    memo[k] = (Z[ix(u.bits)] << np.arange(31, -1, -1, dtype=np.int64)[:, None]).sum(axis=0)
            return memo[k]

        for op in tape:
            if op[0] == "unpack":
                _, first, j = op
                for h, pre in enumerate(preimages):
                    v = int(pre[j])
                    assert 0 <= v < 1 << 128
                    Z[first:first + 128, h] = [(v >> (127 - i)) & 1 for i in range(128)]
            elif op[0] == "bits":
                _, first, width, u = op
                v = value(u)
                assert int(v.max()) >> width == 0
                Z[first:first + width] = (v[None, :] >> np.arange(width - 1, -1, -1, dtype=np.int64)[:, None]) & 1
            elif op[0] == "xor":
                _, out, a, b = op
                res = Z[ix(a)] ^ Z[ix(b)]
                sel = [i for i, o in enumerate(out) if o >= 0]
                Z[[out[i] for i in sel]] = res[sel]
            elif op[0] == "ch":
                _, out, a, b, c = op
                x, y, z = Z[ix(a)], Z[ix(b)], Z[ix(c)]
                Z[out] = (x & y) | ((1 - x) & z)
            elif op[0] == "maj":
                _, out, bcs, a, b, c = op
                x, y, z = Z[ix(a)], Z[ix(b)], Z[ix(c)]
                Z[bcs] = y & z
                Z[out] = (x & y) ^ (x & z) ^ (y & z)
        out = np.empty((H, V), dtype=np.int64)
        out[:, column_of] = Z[:V].T
        return out

    def values(self, preimages):
        """[1, inputs, outputs, witness...] as one uint8 array (canonical LE, 32 bytes per variable)."""
        outs = [sha256_packed(pre) for pre in preimages]
        pub = [1] + [int(v) for pre in preimages for v in pre] + [v for o in outs for v in o]
        z = np.zeros((self.m, 32), dtype=np.uint8)
        z[:self.l] = np.frombuffer(b"".join(v.to_bytes(32, "little") for v in pub), dtype=np.uint8).reshape(-1, 32)
        z[self.l:, 0] = self.witness(preimages).reshape(-1)
        return z.reshape(-1)

    def preimages(self, wseed):
        raw = splitmix64(wseed, 8 * self.hashes).tolist()
        return [[(raw[8 * h + 2 * j] << 64) | raw[8 * h + 2 * j + 1] for j in range(4)] for h in range(self.hashes)]

    def assignment(self, wseed):
        """A satisfying assignment z (uint8[m*32], canonical LE); the preimages are drawn from SplitMix64(wseed)."""
        return self.values(self.preimages(wseed))


def circuit(curve, hashes=1):
    return Sha256Packed(curve, hashes)
