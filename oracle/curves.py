"""Short-Weierstrass (a = 0) group arithmetic over Fq / Fq2, python big ints.  TEST ORACLE ONLY.

Restates the group law used by [UPSTREAM] ark-ec 0.3.0 ``short_weierstrass_jacobian``
(SURVEY.md App. A.1/A.5).  Points: affine = (x, y) or None (infinity); Jacobian = (X, Y, Z),
Z == 0 is infinity.  Results are representation independent (exact group law), so only the
normalised affine outputs are ever compared.
"""
from .fields import FqOps, Fq2Ops


class Group:
    def __init__(self, F, b, gen):
        self.F = F
        self.b = b
        self.gen = gen

    # -- predicates --
    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), self.b)) == F.zero

    # -- Jacobian --
    def to_jac(self, P):
        F = self.F
        return (F.one, F.one, F.zero) if P is None else (P[0], P[1], F.one)

    def to_affine(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z):
            return None
        zi = F.inv(Z)
        zi2 = F.mul(zi, zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def dbl(self, J):
        F = self.F
        X, Y, Z = J
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.mul(X, X)
        B = F.mul(Y, Y)
        C = F.mul(B, B)
        t = F.add(X, B)
        D = F.sub(F.sub(F.mul(t, t), A), C)
        D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fv = F.mul(E, E)
        X3 = F.sub(Fv, F.add(D, D))
        C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def add(self, P, Q):
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Q
        if F.is_zero(Z1):
            return Q
        if F.is_zero(Z2):
            return P
        Z1Z1 = F.mul(Z1, Z1)
        Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2)
        U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2)
        S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        if U1 == U2:
            if S1 == S2:
                return self.dbl(P)
            return (F.one, F.one, F.zero)
        H = F.sub(U2, U1)
        R = F.sub(S2, S1)
        HH = F.mul(H, H)
        HHH = F.mul(H, HH)
        V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(R, R), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(R, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def neg(self, J):
        return (J[0], self.F.neg(J[1]), J[2])

    def mul(self, J, k):
        F = self.F
        R = (F.one, F.one, F.zero)
        if k < 0:
            J, k = self.neg(J), -k
        for bit in bin(k)[2:] if k else "":
            R = self.dbl(R)
            if bit == "1":
                R = self.add(R, J)
        return R

    # -- affine-level helpers --
    def amul(self, P, k):
        return self.to_affine(self.mul(self.to_jac(P), k))

    def aadd(self, P, Q):
        return self.to_affine(self.add(self.to_jac(P), self.to_jac(Q)))

    def aneg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    def msm(self, bases, scalars):
        """Naive MSM over affine bases (None allowed) -> Jacobian."""
        F = self.F
        acc = (F.one, F.one, F.zero)
        for P, k in zip(bases, scalars):
            if P is None or k == 0:
                continue
            acc = self.add(acc, self.mul(self.to_jac(P), k))
        return acc

    def fixed_base_table(self, P, bits, w=4):
        """table[j][d] = d * 2^(w*j) * P (Jacobian) for windowed fixed-base mult."""
        tbl = []
        base = self.to_jac(P)
        for _ in range((bits + w - 1) // w):
            row = [(self.F.one, self.F.one, self.F.zero)]
            for d in range(1, 1 << w):
                row.append(self.add(row[-1], base))
            tbl.append(row)
            for _ in range(w):
                base = self.dbl(base)
        return tbl

    def fixed_mul(self, tbl, k, w=4):
        acc = (self.F.one, self.F.one, self.F.zero)
        j = 0
        while k:
            d = k & ((1 << w) - 1)
            if d:
                acc = self.add(acc, tbl[j][d])
            k >>= w
            j += 1
        return acc


def groups(curve):
    """(G1, G2) for a fields.Curve."""
    g1 = Group(FqOps(curve.q), curve.b1, curve.g1)
    g2 = Group(Fq2Ops(curve.q), curve.b2, curve.g2)
    return g1, g2
