"""Poseidon hash-chain workload (BASELINE.json configs[3]: "stdlib Poseidon hash chain (depth 1024), BLS12-381") —
host-side input generator, product code like synth.py.

The circuit is what ZoKrates' flattener makes of
    state_{h+1} = poseidon([state_h, h])          /root/reference/zokrates_stdlib/stdlib/hashes/poseidon/poseidon.zok:32-62
iterated `depth` times with t = 3 (8 full + 57 partial rounds): additions of round constants and the MDS mix are linear
and stay inside the linear combinations; every x^5 S-box costs three constraints (a*a = x2, x2*x2 = x4, x4*a = x5 with
`a` a linear combination), 3*(8*3 + 57) = 243 constraints and 243 witness variables per hash.  In the 57 partial rounds
two of the three state words are never re-linearised, so the combinations grow to ~60 terms: a far wider sparse mat-vec
than the multiplication chain of synth.py.  Variables in ark order (/root/reference/zokrates_ark/src/lib.rs:80-129):
column 0 = ONE, 1 = the public input state_0, 2 = the public output `~out_0`, then the witness in allocation order.

Round constants and the MDS matrix are not copied from the reference's constants.zok: they are regenerated with the
Poseidon paper's Grain-LFSR procedure (the generator circomlib's — and hence ZoKrates' — tables came from; parameters
field = 1, sbox = 0, n = 254, t, R_F = 8, R_P over the BN254 scalar field) and checked against the reference's hash
known-answer tests (zokrates_stdlib/tests/tests/hashes/poseidon/poseidon_{1,2,3}.json) in tests/test_poseidon.py.  As in
the reference, the same integers are used whatever the curve ("constants are BN254-derived": over BLS12-381 the
workload is valid, the hash non-standard).
"""
import functools

import numpy as np

from .synth import CURVE_IDS, FR_MODULUS, _Stream

BN254_R = FR_MODULUS[0]
ROUNDS_P = [56, 57, 56, 60, 60, 63, 64, 63]       # poseidon.zok:36, indexed by t - 2
ROUNDS_F = 8


@functools.lru_cache(maxsize=None)
def parameters(t):
    """(round constants [t * (R_F + R_P)], MDS matrix t x t) for state width t."""
    p, n, rp = BN254_R, 254, ROUNDS_P[t - 2]
    bits = []
    for value, width in ((1, 2), (0, 4), (n, 12), (t, 12), (ROUNDS_F, 10), (rp, 10)):
        bits.extend(int(b) for b in bin(value)[2:].zfill(width))
    bits.extend([1] * 30)

    def update():
        nb = bits[62] ^ bits[51] ^ bits[38] ^ bits[23] ^ bits[13] ^ bits[0]
        bits.pop(0)
        bits.append(nb)
        return nb

    for _ in range(160):
        update()

    def next_bit():                       # self-shrinking: a 0 discards the following bit
        nb = update()
        while nb == 0:
            update()
            nb = update()
        return update()

    def next_int():
        v = 0
        for _ in range(n):
            v = (v << 1) | next_bit()
        return v

    consts = []
    while len(consts) < t * (ROUNDS_F + rp):
        v = next_int()
        while v >= p:
            v = next_int()
        consts.append(v)
    xy = [next_int() % p for _ in range(2 * t)]
    mds = [[pow((xy[i] + xy[t + j]) % p, p - 2, p) for j in range(t)] for i in range(t)]
    return consts, mds


def poseidon(inputs, modulus=BN254_R):
    """The hash of poseidon.zok over the field of `modulus` (constants as integers, see the module docstring)."""
    t = len(inputs) + 1
    consts, mds = parameters(t)
    rp = ROUNDS_P[t - 2]
    s = [0] + [int(v) % modulus for v in inputs]
    for r in range(ROUNDS_F + rp):
        s = [(x + consts[r * t + i]) % modulus for i, x in enumerate(s)]
        full = r < ROUNDS_F // 2 or r >= ROUNDS_F // 2 + rp
        s = [pow(x, 5, modulus) if (i == 0 or full) else x for i, x in enumerate(s)]
        s = [sum(mds[i][j] * s[j] for j in range(t)) % modulus for i in range(t)]
    return s[0]


VARS_PER_HASH = 3 * (ROUNDS_F * 3 + ROUNDS_P[1])      # 243 at t = 3


def _template(p):
    """One hash with symbolic inputs: rows [(A, B, C)] of linear combinations {key: coeff} over the keys
    0..242 (this hash's variables, allocation order), "one", "S" (incoming state) and "K" (chain index); and the
    outgoing state as a combination over 0..242 and "one"."""
    consts, mds = parameters(3)
    rp = ROUNDS_P[1]
    rows, nvar = [], 0
    state = [{}, {"S": 1}, {"K": 1}]
    for r in range(ROUNDS_F + rp):
        full = r < ROUNDS_F // 2 or r >= ROUNDS_F // 2 + rp
        nxt = []
        for i in range(3):
            a = dict(state[i])
            a["one"] = (a.get("one", 0) + consts[3 * r + i]) % p
            if i == 0 or full:
                x2, x4, x5 = nvar, nvar + 1, nvar + 2
                nvar += 3
                rows += [(a, a, {x2: 1}), ({x2: 1}, {x2: 1}, {x4: 1}), ({x4: 1}, a, {x5: 1})]
                nxt.append({x5: 1})
            else:
                nxt.append(a)
        state = []
        for i in range(3):
            acc = {}
            for j in range(3):
                for k, v in nxt[j].items():
                    acc[k] = (acc.get(k, 0) + mds[i][j] * v) % p
            state.append(acc)
    assert nvar == VARS_PER_HASH
    return rows, state[0]


class PoseidonChain:
    """CSR matrices and assignments of the depth-`depth` chain; same interface as synth.SynthCircuit."""

    def __init__(self, curve, depth):
        self.curve_id = CURVE_IDS.get(curve, curve)
        self.depth, self.kind = depth, "poseidon"
        p = self.p = FR_MODULUS[self.curve_id]
        rows, out = _template(p)
        assert "S" not in out and "K" not in out
        V, base = VARS_PER_HASH, 3
        self.l, self.w = 3, V * depth
        self.m = self.l + self.w
        self.n = len(rows) * depth + 1
        N = 1
        while N < self.n + self.l:
            N *= 2
        self.N = N

        def resolve(lc, h):
            """Template combination -> {column: coeff} for hash h."""
            res = {}

            def add(col, v):
                res[col] = (res.get(col, 0) + v) % p

            for k, v in lc.items():
                if k == "one":
                    add(0, v)
                elif k == "K":
                    add(0, v * h)
                elif k == "S":
                    if h == 0:
                        add(1, v)
                    else:
                        for k2, v2 in out.items():
                            add(0 if k2 == "one" else base + V * (h - 1) + k2, v * v2)
                else:
                    add(base + V * h + k, v)
            return sorted((c, v) for c, v in res.items() if v)

        # hashes 0 and 1 are resolved term by term; hash h >= 2 is hash 1 with every witness column shifted by V*(h-1),
        # except for the ONE coefficient (constant + h) of the combinations that see the chain index K
        def pack(blocks):
            counts = np.array([len(e) for e in blocks], dtype=np.uint64)
            cols = np.array([c for e in blocks for c, _ in e], dtype=np.int64)
            vals = np.frombuffer(b"".join(int(v).to_bytes(32, "little") for e in blocks for _, v in e), dtype=np.uint8).reshape(-1, 32)
            return counts, cols, vals

        mats = []
        for which in range(3):
            c0, col0, val0 = pack([resolve(row[which], 0) for row in rows])
            parts_cnt, parts_col, parts_val = [c0], [col0], [val0]
            if depth > 1:
                blocks1 = [resolve(row[which], 1) for row in rows]
                c1, col1, val1 = pack(blocks1)
                reps = depth - 1
                shift = (np.arange(reps, dtype=np.int64) * V)[:, None]
                colt = np.where(col1[None, :] >= base, col1[None, :] + shift, col1[None, :]).reshape(-1)
                valt = np.tile(val1, (reps, 1))
                # patch the K combinations: entry (ONE, const + 1) of hash 1 becomes (ONE, const + h)
                starts = np.concatenate([[0], np.cumsum(c1)]).astype(np.int64)
                for ri, row in enumerate(rows):
                    if "K" not in row[which]:
                        continue
                    ent = blocks1[ri]
                    pos = [q for q, (c, _) in enumerate(ent) if c == 0]
                    assert len(pos) == 1 and row[which]["K"] == 1
                    v1 = ent[pos[0]][1]
                    for h in range(2, depth):
                        v = (v1 - 1 + h) % p
                        assert v
                        valt[(h - 1) * len(col1) + starts[ri] + pos[0]] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint8)
                parts_cnt.append(np.tile(c1, reps)); parts_col.append(colt); parts_val.append(valt)
            final = [sorted((0 if k == "one" else base + V * (depth - 1) + k, v) for k, v in out.items() if v), [(0, 1)], [(2, 1)]]
            cf, colf, valf = pack([final[which]])
            parts_cnt.append(cf); parts_col.append(colf); parts_val.append(valf)
            rp_arr = np.zeros(self.n + 1, dtype=np.uint64)
            np.cumsum(np.concatenate(parts_cnt), out=rp_arr[1:])
            mats.append((rp_arr, np.concatenate(parts_col).astype(np.uint32), np.ascontiguousarray(np.concatenate(parts_val)).reshape(-1)))
        self.A, self.B, self.C = mats

    def mats(self):
        return [self.A, self.B, self.C]

    def values(self, s0):
        """[1, s0, out, witness...] as python ints for the initial state s0."""
        p = self.p
        consts, mds = parameters(3)
        rp = ROUNDS_P[1]
        wit, s = [], s0 % p
        for h in range(self.depth):
            st = [0, s, h % p]
            for r in range(ROUNDS_F + rp):
                full = r < ROUNDS_F // 2 or r >= ROUNDS_F // 2 + rp
                nxt = []
                for i in range(3):
                    a = (st[i] + consts[3 * r + i]) % p
                    if i == 0 or full:
                        x2 = a * a % p
                        x4 = x2 * x2 % p
                        x5 = x4 * a % p
                        wit += [x2, x4, x5]
                        nxt.append(x5)
                    else:
                        nxt.append(a)
                st = [(mds[i][0] * nxt[0] + mds[i][1] * nxt[1] + mds[i][2] * nxt[2]) % p for i in range(3)]
            s = st[0]
        return [1, s0 % p, s] + wit

    def assignment(self, wseed):
        """A satisfying assignment z (uint8[m*32], canonical LE); the initial state is drawn from SplitMix64(wseed)."""
        s0 = _Stream(wseed).field(self.curve_id)
        return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in self.values(s0)), dtype=np.uint8)


def chain(curve, depth=1024):
    return PoseidonChain(curve, depth)
