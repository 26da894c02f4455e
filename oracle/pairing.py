"""Ate pairing check for BN254, BLS12-381 and (pairing only, for the reference's GM17 golden triple) BLS12-377 in python big ints.  TEST ORACLE ONLY (O3).

Implements the Groth16 verification equation of
/root/reference/zokrates_proof_systems/src/scheme/groth16.rs:156-172
(``pairingProd4(A, B, -vk_x, gamma, -C, delta, -alpha, beta)``) and, upstream, of
[UPSTREAM] ``ark_groth16::verify_proof`` called at /root/reference/zokrates_ark/src/groth16.rs:85.

Fq12 is represented as Fq[w]/(w^12 - c6*w^6 + c0) with w^6 = xi (xi = 9+u for BN254, 1+u for
BLS12-381); simple and slow (affine Miller loop, generic exponentiation), which is all a
checker needs.  Only "product of pairings == 1" is exposed, so the sign convention of the loop
parameter does not matter.
"""
from .fields import BN254, BLS12_381, inv


class Fq12Ctx:
    def __init__(self, curve):
        self.q = curve.q
        if curve is BN254:
            self.c6, self.c0, self.xi0 = 18, 82, 9     # w^12 = 18 w^6 - 82 ; u = w^6 - 9
            self.twist_d = True
            t = 4965661367192848881
            self.loop = 6 * t + 2
            self.bn = True
        elif curve is BLS12_381:
            self.c6, self.c0, self.xi0 = 2, 2, 1       # w^12 = 2 w^6 - 2 ; u = w^6 - 1
            self.twist_d = False
            self.loop = 0xd201000000010000
            self.bn = False
        elif curve.name == "bls12_377":
            # Fq2 = Fq[u]/(u^2 + 5), Fq6 = Fq2[v]/(v^3 - u), Fq12 = Fq6[w]/(w^2 - v)  =>  w^6 = u, w^12 = -5
            self.c6, self.c0, self.xi0 = 0, 5, 0
            self.twist_d = True
            self.loop = 0x8508c00000000001
            self.bn = False
        else:
            raise ValueError("unsupported curve")
        self.r = curve.r
        self.b = curve.b1

    # --- Fq12 arithmetic on 12-lists ---
    def one(self): return [1] + [0] * 11
    def zero(self): return [0] * 12
    def scalar(self, k): return [k % self.q] + [0] * 11

    def add(self, a, b): q = self.q; return [(x + y) % q for x, y in zip(a, b)]
    def sub(self, a, b): q = self.q; return [(x - y) % q for x, y in zip(a, b)]
    def neg(self, a): q = self.q; return [(-x) % q for x in a]

    def mul(self, a, b):
        q = self.q
        c = [0] * 23
        for i, x in enumerate(a):
            if x:
                for j, y in enumerate(b):
                    c[i + j] += x * y
        for i in range(22, 11, -1):
            t = c[i] % q
            if t:
                c[i - 6] += self.c6 * t
                c[i - 12] -= self.c0 * t
        return [x % q for x in c[:12]]

    def sqr(self, a): return self.mul(a, a)

    def pow(self, a, e):
        r = self.one()
        for bit in bin(e)[2:]:
            r = self.sqr(r)
            if bit == "1":
                r = self.mul(r, a)
        return r

    def inv(self, a):
        """Polynomial extended Euclid over Fq[w] modulo the degree-12 modulus."""
        q = self.q
        mod = [self.c0 % q, 0, 0, 0, 0, 0, (-self.c6) % q, 0, 0, 0, 0, 0, 1]

        def deg(p):
            d = len(p) - 1
            while d and p[d] == 0:
                d -= 1
            return d

        lm, hm = [1] + [0] * 12, [0] * 13
        low, high = list(a) + [0], mod
        while deg(low):
            dl, dh = deg(low), deg(high)
            # r = high / low (poly division, quotient only)
            r = [0] * 13
            temp = list(high)
            il = inv(low[dl], q)
            for i in range(dh - dl, -1, -1):
                c = temp[dl + i] * il % q
                r[i] = c
                if c:
                    for j in range(dl + 1):
                        temp[i + j] = (temp[i + j] - c * low[j]) % q
            nm, new = list(hm), list(high)
            for i in range(13):
                if lm[i] or low[i]:
                    for j in range(13 - i):
                        if r[j]:
                            nm[i + j] -= lm[i] * r[j]
                            new[i + j] -= low[i] * r[j]
            nm = [x % q for x in nm]
            new = [x % q for x in new]
            lm, low, hm, high = nm, new, lm, low
        il = inv(low[0], q)
        return [x * il % q for x in lm[:12]]

    def div(self, a, b): return self.mul(a, self.inv(b))

    def eq(self, a, b): return all((x - y) % self.q == 0 for x, y in zip(a, b))

    # --- embeddings ---
    def from_fq2(self, a):
        # a0 + a1*u with u = w^6 - xi0
        r = self.zero()
        r[0] = (a[0] - self.xi0 * a[1]) % self.q
        r[6] = a[1] % self.q
        return r

    def w_pow(self, k):
        r = self.zero(); r[k] = 1
        return r

    def twist(self, Q):
        """G2 affine point over Fq2 -> point on E(Fq12)."""
        x, y = self.from_fq2(Q[0]), self.from_fq2(Q[1])
        if self.twist_d:
            return (self.mul(x, self.w_pow(2)), self.mul(y, self.w_pow(3)))
        return (self.div(x, self.w_pow(2)), self.div(y, self.w_pow(3)))

    def cast_g1(self, P):
        return (self.scalar(P[0]), self.scalar(P[1]))

    # --- curve ops over Fq12 (affine) ---
    def pdbl(self, P):
        x, y = P
        m = self.div(self.mul(self.scalar(3), self.sqr(x)), self.add(y, y))
        nx = self.sub(self.sqr(m), self.add(x, x))
        ny = self.sub(self.mul(m, self.sub(x, nx)), y)
        return (nx, ny)

    def padd(self, P1, P2):
        if P1 is None: return P2
        if P2 is None: return P1
        x1, y1 = P1; x2, y2 = P2
        if self.eq(x1, x2):
            if self.eq(y1, y2):
                return self.pdbl(P1)
            return None
        m = self.div(self.sub(y2, y1), self.sub(x2, x1))
        nx = self.sub(self.sub(self.sqr(m), x1), x2)
        ny = self.sub(self.mul(m, self.sub(x1, nx)), y1)
        return (nx, ny)

    def line(self, P1, P2, T):
        x1, y1 = P1; x2, y2 = P2; xt, yt = T
        if not self.eq(x1, x2):
            m = self.div(self.sub(y2, y1), self.sub(x2, x1))
        elif self.eq(y1, y2):
            m = self.div(self.mul(self.scalar(3), self.sqr(x1)), self.add(y1, y1))
        else:
            return self.sub(xt, x1)
        return self.sub(self.mul(m, self.sub(xt, x1)), self.sub(yt, y1))

    def frob_point(self, P):
        return (self.pow(P[0], self.q), self.pow(P[1], self.q))

    def miller(self, Q2, P1):
        """Miller loop f_{loop,Q}(P) with the BN Frobenius corrections; inputs affine, non-infinity."""
        return self.miller_embedded(self.twist(Q2), P1)

    def miller_embedded(self, Q, P1):
        """The same with Q already on E(Fq12) (the image of a G2 point under `twist`, or a sum of such images)."""
        P = self.cast_g1(P1)
        R = Q
        f = self.one()
        for bit in bin(self.loop)[3:]:
            f = self.mul(self.sqr(f), self.line(R, R, P))
            R = self.pdbl(R)
            if bit == "1":
                f = self.mul(f, self.line(R, Q, P))
                R = self.padd(R, Q)
        if self.bn:
            Q1 = self.frob_point(Q)
            nQ2 = self.frob_point(Q1)
            nQ2 = (nQ2[0], self.neg(nQ2[1]))
            f = self.mul(f, self.line(R, Q1, P))
            R = self.padd(R, Q1)
            f = self.mul(f, self.line(R, nQ2, P))
        return f

    def final_exp(self, f):
        return self.pow(f, (self.q ** 12 - 1) // self.r)

    def pairing_product_is_one(self, pairs, embedded=False):
        """pairs: [(P in G1 affine, Q in G2 affine)], None = infinity (skipped).  embedded=True: every Q is already a
        point of E(Fq12) (see miller_embedded)."""
        f = self.one()
        for P, Q in pairs:
            if P is None or Q is None:
                continue
            f = self.mul(f, self.miller_embedded(Q, P) if embedded else self.miller(Q, P))
        return self.eq(self.final_exp(f), self.one())


def groth16_verify(curve, vk, proof, public_inputs):
    """vk = dict(alpha_g1, beta_g2, gamma_g2, delta_g2, gamma_abc_g1[list]); proof = (A, B, C) affine;
    public_inputs = [x_1..x_{l-1}] (x_0 = 1 implied).
    e(A,B) * e(-vk_x, gamma) * e(-C, delta) * e(-alpha, beta) == 1
    (zokrates_proof_systems/src/scheme/groth16.rs:167-171)."""
    from .curves import groups
    G1, _ = groups(curve)
    ctx = Fq12Ctx(curve)
    A, B, C = proof
    gabc = vk["gamma_abc_g1"]
    assert len(gabc) == len(public_inputs) + 1
    acc = G1.to_jac(gabc[0])
    for x, P in zip(public_inputs, gabc[1:]):
        acc = G1.add(acc, G1.mul(G1.to_jac(P), x % curve.r))
    vk_x = G1.to_affine(acc)
    return ctx.pairing_product_is_one([
        (A, B),
        (G1.aneg(vk_x), vk["gamma_g2"]),
        (G1.aneg(C), vk["delta_g2"]),
        (G1.aneg(vk["alpha_g1"]), vk["beta_g2"]),
    ])
