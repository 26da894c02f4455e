"""Synthetic R1CS workloads for the Groth16 hot path (host-side input generator, product code).

These are the circuits BASELINE.json's configs 2-5 are quoted on: a chain of
multiplication gates  (w_k + c_k * ONE) * (w_{k+1} + d_k * x) = w_{k+2}  with 64-bit coefficients, so
every witness value is a full-width field element ("dense": the worst case for the MSMs), and a
"sha" variant in which 90 % of the rows are boolean constraints  b * (b - 1) = 0  with coin-flip
values — the wire statistics of the reference's SHA-256 example (config 1,
/root/reference/zokrates_cli/examples/book/sha256_tutorial/hashexample.zok).

Variables are in ark order (/root/reference/zokrates_ark/src/lib.rs:80-129): column 0 = ONE,
column 1 = the public input x (so num_instance l = 2), then the witness in allocation order.
The circuit (matrices) depends only on `seed`; assignments for it are drawn from an independent
`wseed`, so one proving key serves any number of distinct witnesses.

No oracle code is used here: tests feed these matrices to the oracle through its CSR loader.
"""
import numpy as np

CURVE_IDS = {"bn128": 0, "bls12_381": 1}
FR_MODULUS = {
    0: 21888242871839275222246405745257275088548364400416034343698204186575808495617,
    1: 52435875175126190479447740508185965837690552500527637822603658699938581184513,
}
FR_BITS = {0: 254, 1: 255}

_GAMMA = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_MASK64 = (1 << 64) - 1


def splitmix64(seed, count, start=0):
    """Outputs start+1 .. start+count of SplitMix64(seed) as uint64 (the generator is counter based)."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & _MASK64) + idx * _GAMMA
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        return z ^ (z >> np.uint64(31))


class _Stream:
    """Sequential view of SplitMix64 for the few scalar draws (field elements by rejection sampling)."""

    def __init__(self, seed):
        self.seed, self.pos = seed, 0

    def next(self):
        v = int(splitmix64(self.seed, 1, self.pos)[0])
        self.pos += 1
        return v

    def field(self, curve_id):
        p, bits = FR_MODULUS[curve_id], FR_BITS[curve_id]
        while True:
            v = 0
            for i in range(4):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < p:
                return v


def _limbs(value):
    return [(value >> (64 * i)) & _MASK64 for i in range(4)]


class SynthCircuit:
    """CSR matrices A, B, C (rowptr uint64[n+1], col uint32[nnz], val uint8[nnz*32] canonical LE)."""

    def __init__(self, curve_id, n, kind, seed):
        self.curve_id, self.n, self.kind, self.seed = curve_id, n, kind, seed
        self.l = 2
        p = FR_MODULUS[curve_id]
        raw = splitmix64(seed, 3 * n)
        flag, cc, dd = raw[0::3], raw[1::3], raw[2::3]
        boolean = (flag % np.uint64(10) != 0) if kind == "sha" else np.zeros(n, dtype=bool)
        chain = ~boolean
        col = np.arange(4, 4 + n, dtype=np.uint32)          # row k allocates variable 4 + k
        # the two most recent chain variables before each row (variables 2, 3 seed the chain)
        chain_cols = np.concatenate([np.array([2, 3], dtype=np.uint32), col[chain]])
        rank = np.cumsum(chain) - chain                      # chain rows strictly before row k
        i0, i1 = chain_cols[rank], chain_cols[rank + 1]
        self.boolean, self.cc, self.dd, self.i0, self.i1 = boolean, cc, dd, i0, i1
        self.w = n + 2
        self.m = self.l + self.w
        N = 1
        while N < n + self.l:
            N *= 2
        self.N = N

        one = np.array(_limbs(1), dtype=np.uint64)
        minus_one = np.array(_limbs(p - 1), dtype=np.uint64)

        def build(counts, cols, vals):
            rp = np.zeros(n + 1, dtype=np.uint64)
            np.cumsum(counts, out=rp[1:])
            return rp, np.ascontiguousarray(cols, dtype=np.uint32), np.ascontiguousarray(vals, dtype=np.uint64).view(np.uint8).reshape(-1)

        # A: chain rows (i0, 1), (0, c_k); boolean rows (col, 1)
        nnz_a = np.where(chain, 2, 1)
        off_a = np.concatenate([[0], np.cumsum(nnz_a)])[:-1]
        ca = np.zeros(int(nnz_a.sum()), dtype=np.uint32)
        va = np.zeros((ca.size, 4), dtype=np.uint64)
        ca[off_a[chain]] = i0[chain]; va[off_a[chain]] = one
        ca[off_a[chain] + 1] = 0; va[off_a[chain] + 1, 0] = cc[chain]
        ca[off_a[boolean]] = col[boolean]; va[off_a[boolean]] = one
        self.A = build(nnz_a, ca, va)
        # B: chain rows (i1, 1), (1, d_k); boolean rows (col, 1), (0, -1)
        off_b = 2 * np.arange(n)
        cb = np.zeros(2 * n, dtype=np.uint32)
        vb = np.zeros((2 * n, 4), dtype=np.uint64)
        cb[off_b[chain]] = i1[chain]; vb[off_b[chain]] = one
        cb[off_b[chain] + 1] = 1; vb[off_b[chain] + 1, 0] = dd[chain]
        cb[off_b[boolean]] = col[boolean]; vb[off_b[boolean]] = one
        cb[off_b[boolean] + 1] = 0; vb[off_b[boolean] + 1] = minus_one
        self.B = build(np.full(n, 2), cb, vb)
        # C: chain rows (col, 1); boolean rows empty
        vc = np.zeros((int(chain.sum()), 4), dtype=np.uint64)
        vc[:] = one
        self.C = build(chain.astype(np.int64), col[chain], vc)

    def mats(self):
        return [self.A, self.B, self.C]

    def assignment(self, wseed):
        """A satisfying assignment z (uint8[m*32], canonical LE) drawn from SplitMix64(wseed)."""
        p = FR_MODULUS[self.curve_id]
        st = _Stream(wseed)
        x, w0, w1 = st.field(self.curve_id), st.field(self.curve_id), st.field(self.curve_id)
        z = [1, x, w0, w1]
        bits = (splitmix64(wseed ^ 0xB001, self.n) & np.uint64(1)).tolist()
        boolean = self.boolean.tolist()
        cc, dd = self.cc.tolist(), self.dd.tolist()
        a, b = w0, w1
        for k in range(self.n):
            if boolean[k]:
                z.append(bits[k])
            else:
                v = (a + cc[k]) * (b + dd[k] * x) % p
                z.append(v)
                a, b = b, v
        return np.frombuffer(b"".join(v.to_bytes(32, "little") for v in z), dtype=np.uint8)


def toxic_waste(curve, seed=0xC0FFEE):
    """(alpha, beta, gamma, delta, tau): fixed 'toxic waste' for benchmark keys — five non-zero Fr elements."""
    curve_id = CURVE_IDS.get(curve, curve)
    st = _Stream(seed)
    vals = []
    while len(vals) < 5:
        v = st.field(curve_id)
        if v:
            vals.append(v)
    return tuple(vals)


def circuit(curve, log_domain=None, n=None, kind="dense", seed=0xC1C0):
    """`log_domain` = k gives n = 2^k - 2 constraints, i.e. a QAP domain of exactly 2^k (n + l = 2^k)."""
    curve_id = CURVE_IDS.get(curve, curve)
    if n is None:
        n = (1 << log_domain) - 2
    return SynthCircuit(curve_id, n, kind, seed)
