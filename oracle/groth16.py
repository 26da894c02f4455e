"""Groth16 over an R1CS: setup, algorithmic prover (O2) and closed-form trapdoor prover (O1).
TEST ORACLE ONLY — python big ints, small sizes.

Follows, step by step:
  * /root/reference/zokrates_ark/src/groth16.rs:20-53 (sequence: inputs -> pk -> prove),
  * [UPSTREAM] ark-groth16 0.3.0 ``create_random_proof`` / ``LibsnarkReduction::witness_map``
    (SURVEY.md App. A.3), ark-poly 0.3.0 ``Radix2EvaluationDomain`` (App. A.4),
    ark-groth16 ``generate_random_parameters`` (App. A.6; here with *fixed* generators and
    caller-supplied toxic waste so that O1 is computable),
  * SURVEY.md App. A.9 for the trapdoor closed form.
"""
from dataclasses import dataclass, field
from .fields import inv, SplitMix64
from .curves import groups


@dataclass
class R1CS:
    """Matrices in ark variable order: column 0 = ONE, columns < l are instance, the rest witness.
    Rows are lists of (col, coeff)."""
    l: int
    w: int
    A: list = field(default_factory=list)
    B: list = field(default_factory=list)
    C: list = field(default_factory=list)

    @property
    def n(self): return len(self.A)
    @property
    def m(self): return self.l + self.w

    def domain_size(self):
        N = 1
        while N < self.n + self.l:
            N *= 2
        return N

    def is_satisfied(self, z, r):
        ev = lambda row: sum(c * z[j] for j, c in row) % r
        return all(ev(a) * ev(b) % r == ev(c) for a, b, c in zip(self.A, self.B, self.C))


def synthetic_chain(curve, n, seed, kind="dense"):
    """Synthetic R1CS of SURVEY.md §8(d) config 2.
    l = 2 (ONE, x); witness w_0..w_{n+1}; constraint k: (w_k + c_k*ONE) * (w_{k+1} + d_k*x) = w_{k+2}.
    kind="sha": ~90% of constraints are boolean checks on fresh Bernoulli(1/2) wires
    (w*(w-1) = 0); the remainder is the dense chain.
    Returns (R1CS, z) with z = [1, x, w...]; PRNG = SplitMix64(seed).
    Coefficient stream order (must match the C++ oracle and zokrates_amd.synthetic):
      x, w0, w1, then per constraint k: [sel,] c_k, d_k.
    """
    r = curve.r
    rng = SplitMix64(seed)
    x = rng.field(r)
    z = [1, x, rng.field(r), rng.field(r)]
    cs = R1CS(l=2, w=0)
    last2 = [2, 3]            # columns of the two most recent chain wires
    for k in range(n):
        boolean = False
        if kind == "sha":
            boolean = (rng.next() % 10) != 0
        if boolean:
            bit = rng.next() & 1
            col = len(z)
            z.append(bit)
            cs.A.append([(col, 1)])
            cs.B.append([(col, 1), (0, r - 1)])
            cs.C.append([])
        else:
            c = rng.next()
            d = rng.next()
            i0, i1 = last2
            col = len(z)
            val = (z[i0] + c) * (z[i1] + d * x) % r
            z.append(val)
            cs.A.append([(i0, 1), (0, c)])
            cs.B.append([(i1, 1), (1, d)])
            cs.C.append([(col, 1)])
            last2 = [i1, col]
    cs.w = len(z) - cs.l
    return cs, z


# ---------------- radix-2 domain (App. A.4) ----------------
class Domain:
    def __init__(self, curve, N):
        assert N & (N - 1) == 0
        self.N = N
        self.r = curve.r
        k = N.bit_length() - 1
        assert k <= curve.two_adicity
        self.omega = pow(curve.two_adic_root, 1 << (curve.two_adicity - k), curve.r)
        self.g = curve.fr_generator

    def _fft(self, a, root):
        r, N = self.r, self.N
        a = list(a) + [0] * (N - len(a))
        # iterative radix-2 DIT, natural in / natural out
        j = 0
        for i in range(1, N):
            bit = N >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j |= bit
            if i < j:
                a[i], a[j] = a[j], a[i]
        length = 2
        while length <= N:
            wl = pow(root, N // length, r)
            for s in range(0, N, length):
                w = 1
                for t in range(length // 2):
                    u = a[s + t]
                    v = a[s + t + length // 2] * w % r
                    a[s + t] = (u + v) % r
                    a[s + t + length // 2] = (u - v) % r
                    w = w * wl % r
            length *= 2
        return a

    def fft(self, a): return self._fft(a, self.omega)

    def ifft(self, a):
        ni = inv(self.N, self.r)
        return [x * ni % self.r for x in self._fft(a, inv(self.omega, self.r))]

    def coset_fft(self, a):
        r = self.r
        out, p = [], 1
        for x in list(a) + [0] * (self.N - len(a)):
            out.append(x * p % r)
            p = p * self.g % r
        return self.fft(out)

    def coset_ifft(self, a):
        r = self.r
        gi = inv(self.g, r)
        out, p = [], 1
        for x in self.ifft(a):
            out.append(x * p % r)
            p = p * gi % r
        return out

    def lagrange_at(self, tau):
        """u_k = L_k(tau) for the size-N domain, k < N."""
        r, N = self.r, self.N
        zt = (pow(tau, N, r) - 1) % r
        if zt == 0:
            raise ValueError("tau in domain")
        ni = inv(N, r)
        out, wk = [], 1
        for _ in range(N):
            out.append(zt * ni % r * wk % r * inv((tau - wk) % r, r) % r)
            wk = wk * self.omega % r
        return out


# ---------------- setup with known toxic waste (App. A.6) ----------------
@dataclass
class Toxic:
    alpha: int
    beta: int
    gamma: int
    delta: int
    tau: int

    @staticmethod
    def from_seed(curve, seed=0xC0FFEE):
        rng = SplitMix64(seed)
        vals = []
        for _ in range(5):
            v = 0
            while v == 0:
                v = rng.field(curve.r)
            vals.append(v)
        return Toxic(*vals)


def qap_at_tau(curve, cs, tox):
    """a_i(tau), b_i(tau), c_i(tau) per variable incl. the input-consistency rows (u_{n+i} into a_i, i<l)."""
    r = curve.r
    dom = Domain(curve, cs.domain_size())
    u = dom.lagrange_at(tox.tau)
    a = [0] * cs.m; b = [0] * cs.m; c = [0] * cs.m
    for k in range(cs.n):
        for j, v in cs.A[k]: a[j] = (a[j] + v * u[k]) % r
        for j, v in cs.B[k]: b[j] = (b[j] + v * u[k]) % r
        for j, v in cs.C[k]: c[j] = (c[j] + v * u[k]) % r
    for i in range(cs.l):
        a[i] = (a[i] + u[cs.n + i]) % r
    zt = (pow(tox.tau, dom.N, r) - 1) % r
    return a, b, c, zt, dom


def setup(curve, cs, tox):
    """Returns (pk, vk) as dicts of affine points (None = infinity).  pk field names/order follow
    [UPSTREAM] ark_groth16::ProvingKey (App. B.3)."""
    r = curve.r
    G1, G2 = groups(curve)
    a, b, c, zt, dom = qap_at_tau(curve, cs, tox)
    t1 = G1.fixed_base_table(G1.gen, r.bit_length())
    t2 = G2.fixed_base_table(G2.gen, r.bit_length())
    g1 = lambda k: G1.to_affine(G1.fixed_mul(t1, k % r))
    g2 = lambda k: G2.to_affine(G2.fixed_mul(t2, k % r))
    gi, di = inv(tox.gamma, r), inv(tox.delta, r)
    abc = [(tox.beta * a[i] + tox.alpha * b[i] + c[i]) % r for i in range(cs.m)]
    vk = dict(
        alpha_g1=g1(tox.alpha), beta_g2=g2(tox.beta), gamma_g2=g2(tox.gamma), delta_g2=g2(tox.delta),
        gamma_abc_g1=[g1(abc[i] * gi) for i in range(cs.l)],
    )
    N = dom.N
    hq, p = [], zt * di % r
    for _ in range(N - 1):
        hq.append(g1(p))
        p = p * tox.tau % r
    pk = dict(
        vk=vk,
        beta_g1=g1(tox.beta), delta_g1=g1(tox.delta),
        a_query=[g1(a[i]) for i in range(cs.m)],
        b_g1_query=[g1(b[i]) for i in range(cs.m)],
        b_g2_query=[g2(b[i]) for i in range(cs.m)],
        h_query=hq,
        l_query=[g1(abc[i] * di) for i in range(cs.l, cs.m)],
    )
    return pk, vk


# ---------------- O2: algorithmic prover (App. A.3) ----------------
def witness_map(curve, cs, z):
    """h coefficients, length N (h[N-1] == 0 for a satisfying assignment)."""
    r = curve.r
    N = cs.domain_size()
    dom = Domain(curve, N)
    ev = lambda row: sum(cf * z[j] for j, cf in row) % r
    a = [ev(row) for row in cs.A] + [z[j] for j in range(cs.l)]
    b = [ev(row) for row in cs.B]
    c = [ev(row) for row in cs.C]
    a = dom.coset_fft(dom.ifft(a))
    b = dom.coset_fft(dom.ifft(b))
    c = dom.coset_fft(dom.ifft(c))
    zinv = inv((pow(dom.g, N, r) - 1) % r, r)
    ab = [((x * y - w) % r) * zinv % r for x, y, w in zip(a, b, c)]
    return dom.coset_ifft(ab)


def prove(curve, cs, pk, z, r_, s_):
    """Returns (A, B, C) affine.  Mirrors ark_groth16::create_proof_with_reduction."""
    r = curve.r
    G1, G2 = groups(curve)
    h = witness_map(curve, cs, z)
    N = len(h)
    H = G1.msm(pk["h_query"], h[:N - 1])
    L = G1.msm(pk["l_query"], z[cs.l:])
    J1, J2 = G1.to_jac, G2.to_jac
    vk = pk["vk"]

    def coeff(G, J, query, vk_param, delta, rs):
        acc = G.msm(query[1:], z[1:])
        acc = G.add(acc, J(query[0]))
        acc = G.add(acc, G.mul(J(delta), rs))
        return G.add(acc, J(vk_param))

    gA = coeff(G1, J1, pk["a_query"], vk["alpha_g1"], pk["delta_g1"], r_)
    if r_ % r != 0:
        gB1 = coeff(G1, J1, pk["b_g1_query"], pk["beta_g1"], pk["delta_g1"], s_)
    else:
        gB1 = J1(None)
    gB2 = coeff(G2, J2, pk["b_g2_query"], vk["beta_g2"], vk["delta_g2"], s_)
    gC = G1.mul(gA, s_)
    gC = G1.add(gC, G1.mul(gB1, r_))
    gC = G1.add(gC, G1.neg(G1.mul(J1(pk["delta_g1"]), r_ * s_ % r)))
    gC = G1.add(gC, L)
    gC = G1.add(gC, H)
    return G1.to_affine(gA), G2.to_affine(gB2), G1.to_affine(gC)


# ---------------- O1: trapdoor closed form (App. A.9) ----------------
def trapdoor_scalars(curve, cs, tox, z, r_, s_):
    """Discrete logs (wrt the fixed generators) of A (G1), B (G2), C (G1)."""
    r = curve.r
    a, b, c, zt, _ = qap_at_tau(curve, cs, tox)
    Az = sum(x * y for x, y in zip(z, a)) % r
    Bz = sum(x * y for x, y in zip(z, b)) % r
    Cz = sum(x * y for x, y in zip(z, c)) % r
    di = inv(tox.delta, r)
    ht_z = (Az * Bz - Cz) % r          # = h(tau) * Z(tau)
    la = (tox.alpha + Az + r_ * tox.delta) % r
    lb = (tox.beta + Bz + s_ * tox.delta) % r
    priv = sum(z[j] * (tox.beta * a[j] + tox.alpha * b[j] + c[j]) for j in range(cs.l, cs.m)) % r
    lc = (priv * di + ht_z * di + s_ * la + r_ * lb - r_ * s_ % r * tox.delta) % r
    return la, lb, lc


def trapdoor_prove(curve, cs, tox, z, r_, s_):
    G1, G2 = groups(curve)
    la, lb, lc = trapdoor_scalars(curve, cs, tox, z, r_, s_)
    return G1.amul(G1.gen, la), G2.amul(G2.gen, lb), G1.amul(G1.gen, lc)
