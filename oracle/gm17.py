"""GM17 (Groth-Maller 2017) over an R1CS: R1CS->SAP reduction, setup, algorithmic prover, closed-form
trapdoor prover and the two verification equations.  TEST ORACLE ONLY — python big ints, small sizes.

Config 5 of BASELINE.json ("GM17 scheme ... second proof system behind same Backend trait").

Follows, step by step:
  * /root/reference/zokrates_ark/src/gm17.rs:19-41 (setup: `circuit_specific_setup`, `serialize_unchecked`,
    the vk fields h_g2, g_alpha_g1, h_beta_g2, g_gamma_g1, h_gamma_g2, query), :43-78 (generate_proof: inputs ->
    `deserialize_unchecked` -> `GM17::prove` -> a, b, c);
  * the verification equations of /root/reference/zokrates_proof_systems/src/scheme/gm17.rs:168-184
    (`pairingProd4(g_alpha, h_beta, vk_x, h_gamma, C, h, -(A + g_alpha), B + h_beta)` and
    `pairingProd2(A, h_gamma, -g_gamma, B)`);
  * [UPSTREAM] ark-gm17 0.3.0 (`/root/reference/Cargo.lock:202-218`; source not vendored):
    `R1CStoSAP::instance_map_with_evaluation`, `R1CStoSAP::witness_map`, `generate_parameters`,
    `create_proof` — restated from the published algorithm.

Parity status: **unpinned** like G16 — the reference holds no GM17 proof or proving key for BN254/BLS12-381
(its only GM17 golden triple, zokrates_stdlib/tests/tests/snark/gm17.json, is over BLS12-377 and pins the JSON
encoding, not these curves).  What pins this module: the in-tree verification equations above (O3), the closed-form
trapdoor proof (O1) that must equal the algorithmic prover (O2) bit for bit, and the uniqueness of a GM17 proof for
fixed (pk, z, r + d1).  Two details of the *key* layout are restated from memory of the upstream source and cannot be
checked here: `c_query_2[i]` carries the factor 2 (`double_gamma2_z`), and `g_gamma2_z_t` has D + 1 entries.

SAP layout (D0 = 2n + 2(l-1) + 1 rows, variables [1, x_1..x_{l-1}, aux_0..aux_{w-1}, e_0..e_{n-1}, f_1..f_{l-1}]):
  row 2k      : (A_k + B_k)^2 = 4 C_k + e_k          e_k = (<A_k,z> - <B_k,z>)^2
  row 2k+1    : (A_k - B_k)^2 = e_k
  row 2n      : 1^2 = 1
  row 2n+2i-1 : (x_i + 1)^2 = 4 x_i + f_i             f_i = (x_i - 1)^2,   i = 1..l-1
  row 2n+2i   : (x_i - 1)^2 = f_i
"""
from dataclasses import dataclass

from .curves import groups
from .fields import SplitMix64, inv
from .formats import ser_g1, ser_g2, ser_vec
from .groth16 import Domain


def sap_shape(cs):
    """(number of SAP variables incl. ONE, rows before padding, domain size)."""
    M = 1 + 2 * (cs.l - 1) + cs.w + cs.n            # sap_num_variables + 1
    D0 = 2 * cs.n + 2 * (cs.l - 1) + 1
    D = 1
    while D < D0:
        D *= 2
    return M, D0, D


def extend_assignment(curve, cs, z):
    """full_input_assignment of R1CStoSAP::witness_map: z ++ e ++ f."""
    r = curve.r
    ev = lambda row: sum(cf * z[j] for j, cf in row) % r
    ext = list(z)
    ext += [pow(ev(a) - ev(b), 2, r) for a, b in zip(cs.A, cs.B)]
    ext += [pow(z[i] - 1, 2, r) for i in range(1, cs.l)]
    return ext


def sap_at_t(curve, cs, t):
    """R1CStoSAP::instance_map_with_evaluation: a_i = u_i(t), c_i = w_i(t) per SAP variable, Z(t), D."""
    r = curve.r
    M, D0, D = sap_shape(cs)
    dom = Domain(curve, D)
    u = dom.lagrange_at(t)
    n, l, m = cs.n, cs.l, cs.m
    a = [0] * M
    c = [0] * M
    off_e, off_f, off_rows = m, m + n - 1, 2 * n
    for k in range(n):
        u_add = (u[2 * k] + u[2 * k + 1]) % r
        u_sub = (u[2 * k] - u[2 * k + 1]) % r
        for j, v in cs.A[k]: a[j] = (a[j] + u_add * v) % r
        for j, v in cs.B[k]: a[j] = (a[j] + u_sub * v) % r
        for j, v in cs.C[k]: c[j] = (c[j] + 4 * u[2 * k] * v) % r
        c[off_e + k] = (c[off_e + k] + u_add) % r
    a[0] = (a[0] + u[off_rows]) % r
    c[0] = (c[0] + u[off_rows]) % r
    for i in range(1, l):
        u1, u2 = u[off_rows + 2 * i - 1], u[off_rows + 2 * i]
        a[i] = (a[i] + u1 + u2) % r
        a[0] = (a[0] + u1 - u2) % r
        c[i] = (c[i] + 4 * u1) % r
        c[off_f + i] = (c[off_f + i] + u1 + u2) % r
    zt = (pow(t, D, r) - 1) % r
    return a, c, zt, D


@dataclass
class Toxic:
    alpha: int
    beta: int
    gamma: int      # ark's generate_random_parameters fixes gamma = 1; generate_parameters takes any
    t: int

    @staticmethod
    def from_seed(curve, seed=0xC0FFEE, gamma_one=False):
        rng = SplitMix64(seed)
        vals = []
        for _ in range(4):
            v = 0
            while v == 0:
                v = rng.field(curve.r)
            vals.append(v)
        if gamma_one:
            vals[2] = 1
        return Toxic(*vals)


def key_scalars(curve, cs, tox):
    """Discrete logs of every key element (wrt the generators g, h): shared by setup and by tests of the device setup."""
    r = curve.r
    a, c, zt, D = sap_at_t(curve, cs, tox.t)
    M = len(a)
    g, ab = tox.gamma, (tox.alpha + tox.beta) % r
    gz = g * zt % r
    g2z = g * gz % r
    ks = dict(
        a_query=[x * g % r for x in a],
        c_query_1=[(c[i] * g % r * g + a[i] * ab % r * g) % r for i in range(cs.l, M)],
        c_query_2=[2 * g2z * x % r for x in a],
        query=[(g * c[i] + ab * a[i]) % r for i in range(cs.l)],
        g_gamma_z=gz, ab_gamma_z=ab * gz % r, gamma2_z2=gz * gz % r,
        g_gamma2_z_t=[], zt=zt, D=D,
    )
    p = g2z
    for _ in range(D + 1):
        ks["g_gamma2_z_t"].append(p)
        p = p * tox.t % r
    return ks


def setup(curve, cs, tox):
    """(pk, vk) as dicts of affine points.  Field names/order: [UPSTREAM] ark_gm17::{ProvingKey, VerifyingKey}."""
    r = curve.r
    G1, G2 = groups(curve)
    ks = key_scalars(curve, cs, tox)
    t1 = G1.fixed_base_table(G1.gen, r.bit_length())
    t2 = G2.fixed_base_table(G2.gen, r.bit_length())
    g1 = lambda k: G1.to_affine(G1.fixed_mul(t1, k % r))
    g2 = lambda k: G2.to_affine(G2.fixed_mul(t2, k % r))
    vk = dict(h_g2=g2(1), g_alpha_g1=g1(tox.alpha), h_beta_g2=g2(tox.beta), g_gamma_g1=g1(tox.gamma), h_gamma_g2=g2(tox.gamma),
              query=[g1(k) for k in ks["query"]])
    pk = dict(
        vk=vk,
        a_query=[g1(k) for k in ks["a_query"]],
        b_query=[g2(k) for k in ks["a_query"]],
        c_query_1=[g1(k) for k in ks["c_query_1"]],
        c_query_2=[g1(k) for k in ks["c_query_2"]],
        g_gamma_z=g1(ks["g_gamma_z"]), h_gamma_z=g2(ks["g_gamma_z"]),
        g_ab_gamma_z=g1(ks["ab_gamma_z"]), g_gamma2_z2=g1(ks["gamma2_z2"]),
        g_gamma2_z_t=[g1(k) for k in ks["g_gamma2_z_t"]],
    )
    return pk, vk


def pk_serialize(curve, pk):
    """ark `serialize_unchecked` of ark_gm17::ProvingKey (derive order = struct field order)."""
    g1 = lambda P: ser_g1(curve, P)
    g2 = lambda P: ser_g2(curve, P)
    vk = pk["vk"]
    out = g2(vk["h_g2"]) + g1(vk["g_alpha_g1"]) + g2(vk["h_beta_g2"]) + g1(vk["g_gamma_g1"]) + g2(vk["h_gamma_g2"])
    out += ser_vec(vk["query"], g1)
    out += ser_vec(pk["a_query"], g1) + ser_vec(pk["b_query"], g2) + ser_vec(pk["c_query_1"], g1) + ser_vec(pk["c_query_2"], g1)
    out += g1(pk["g_gamma_z"]) + g2(pk["h_gamma_z"]) + g1(pk["g_ab_gamma_z"]) + g1(pk["g_gamma2_z2"])
    out += ser_vec(pk["g_gamma2_z_t"], g1)
    return out


def vk_from_pk_bytes(curve, data):
    """The verifying key embedded at the head of a serialized ark_gm17::ProvingKey (`data` may be a prefix)."""
    from .formats import _Rd, de_g1, de_g2, de_vec
    nb = curve.fq_bytes
    rd = _Rd(bytes(data[:5 * 4 * nb + 8]))
    vk = dict(h_g2=de_g2(curve, rd), g_alpha_g1=de_g1(curve, rd), h_beta_g2=de_g2(curve, rd), g_gamma_g1=de_g1(curve, rd),
              h_gamma_g2=de_g2(curve, rd))
    off = rd.o
    l = int.from_bytes(bytes(data[off:off + 8]), "little")
    rd = _Rd(bytes(data[off:off + 8 + l * 2 * nb]))
    vk["query"] = de_vec(rd, lambda: de_g1(curve, rd))
    return vk


# ---------------- O2: algorithmic prover ----------------
def witness_map(curve, cs, z, d1, d2):
    """R1CStoSAP::witness_map: (extended assignment, h with D + 1 coefficients)."""
    r = curve.r
    M, D0, D = sap_shape(cs)
    dom = Domain(curve, D)
    n, l, m = cs.n, cs.l, cs.m
    ev = lambda row: sum(cf * z[j] for j, cf in row) % r
    ext = extend_assignment(curve, cs, z)
    a = [0] * D
    c = [0] * D
    for k in range(n):
        ak, bk, ck = ev(cs.A[k]), ev(cs.B[k]), ev(cs.C[k])
        a[2 * k], a[2 * k + 1] = (ak + bk) % r, (ak - bk) % r
        c[2 * k], c[2 * k + 1] = (4 * ck + ext[m + k]) % r, ext[m + k]
    a[2 * n] = c[2 * n] = 1
    for i in range(1, l):
        f = ext[m + n - 1 + i]
        a[2 * n + 2 * i - 1], a[2 * n + 2 * i] = (z[i] + 1) % r, (z[i] - 1) % r
        c[2 * n + 2 * i - 1], c[2 * n + 2 * i] = (4 * z[i] + f) % r, f
    a = dom.ifft(a)
    h = [2 * d1 * x % r for x in a]
    h[0] = (h[0] - d2 - d1 * d1) % r
    h.append(d1 * d1 % r)
    a = dom.coset_fft(a)
    c = dom.coset_fft(dom.ifft(c))
    zinv = inv((pow(dom.g, D, r) - 1) % r, r)
    q = dom.coset_ifft([(x * x - y) % r * zinv % r for x, y in zip(a, c)])
    for i in range(D - 1):
        h[i] = (h[i] + q[i]) % r
    return ext, h


def prove(curve, cs, pk, z, d1, d2, r_):
    """(A, B, C) affine.  Mirrors ark_gm17::create_proof."""
    r = curve.r
    G1, G2 = groups(curve)
    J1, J2 = G1.to_jac, G2.to_jac
    ext, h = witness_map(curve, cs, z, d1, d2)
    l = cs.l
    gA = G1.mul(J1(pk["g_gamma_z"]), r_)
    gA = G1.add(gA, J1(pk["a_query"][0]))
    gA = G1.add(gA, G1.mul(J1(pk["g_gamma_z"]), d1))
    gA = G1.add(gA, G1.msm(pk["a_query"][1:], ext[1:]))
    gB = G2.mul(J2(pk["h_gamma_z"]), r_)
    gB = G2.add(gB, J2(pk["b_query"][0]))
    gB = G2.add(gB, G2.mul(J2(pk["h_gamma_z"]), d1))
    gB = G2.add(gB, G2.msm(pk["b_query"][1:], ext[1:]))
    c1 = G1.msm(pk["c_query_1"], ext[l:])
    c2 = G1.msm(pk["c_query_2"][1:], ext[1:])
    gacc = G1.msm(pk["g_gamma2_z_t"], h)
    gC = c1
    gC = G1.add(gC, G1.mul(J1(pk["g_gamma2_z2"]), r_ * r_ % r))
    gC = G1.add(gC, G1.mul(J1(pk["g_ab_gamma_z"]), r_))
    gC = G1.add(gC, G1.mul(J1(pk["g_ab_gamma_z"]), d1))
    gC = G1.add(gC, G1.mul(J1(pk["c_query_2"][0]), r_))
    gC = G1.add(gC, G1.mul(J1(pk["g_gamma2_z2"]), 2 * r_ * d1 % r))
    gC = G1.add(gC, G1.mul(c2, r_))
    gC = G1.add(gC, G1.mul(J1(pk["g_gamma2_z_t"][0]), d2))
    gC = G1.add(gC, gacc)
    return G1.to_affine(gA), G2.to_affine(gB), G1.to_affine(gC)


# ---------------- O1: trapdoor closed form ----------------
def trapdoor_scalars(curve, cs, tox, z, d1, r_):
    """Discrete logs of A (wrt g), B (wrt h), C (wrt g).  With U = sum ext_i u_i(t), W = sum ext_i w_i(t), rho = r + d1:
       A = B = gamma (U + rho Z),
       C = gamma^2 W_aux + (alpha+beta) gamma U_aux + gamma^2 (U^2 - W) + gamma^2 Z (2 rho U + rho^2 Z) + (alpha+beta) gamma rho Z
    (from verification equation 1; d2 cancels)."""
    r = curve.r
    a, c, zt, _ = sap_at_t(curve, cs, tox.t)
    ext = extend_assignment(curve, cs, z)
    dot = lambda v, lo: sum(x * y for x, y in zip(ext[lo:], v[lo:])) % r
    U, W = dot(a, 0), dot(c, 0)
    Ua, Wa = dot(a, cs.l), dot(c, cs.l)
    g, ab, rho = tox.gamma, (tox.alpha + tox.beta) % r, (r_ + d1) % r
    la = g * (U + rho * zt) % r
    lc = (g * g * Wa + ab * g * Ua + g * g * (U * U - W) + g * g * zt * (2 * rho * U + rho * rho * zt) + ab * g * rho * zt) % r
    return la, la, lc


def trapdoor_prove(curve, cs, tox, z, d1, r_):
    G1, G2 = groups(curve)
    la, lb, lc = trapdoor_scalars(curve, cs, tox, z, d1, r_)
    return G1.amul(G1.gen, la), G2.amul(G2.gen, lb), G1.amul(G1.gen, lc)


# ---------------- O3: verification ----------------
def verify(curve, vk, proof, public_inputs):
    """The two checks of zokrates_proof_systems/src/scheme/gm17.rs:168-184."""
    from .pairing import Fq12Ctx
    G1, G2 = groups(curve)
    ctx = Fq12Ctx(curve)
    A, B, C = proof
    q = vk["query"]
    assert len(q) == len(public_inputs) + 1
    acc = G1.to_jac(q[0])
    for x, P in zip(public_inputs, q[1:]):
        acc = G1.add(acc, G1.mul(G1.to_jac(P), x % curve.r))
    vk_x = G1.to_affine(acc)
    ok1 = ctx.pairing_product_is_one([
        (vk["g_alpha_g1"], vk["h_beta_g2"]),
        (vk_x, vk["h_gamma_g2"]),
        (C, vk["h_g2"]),
        (G1.aneg(G1.aadd(A, vk["g_alpha_g1"])), G2.aadd(B, vk["h_beta_g2"])),
    ])
    ok2 = ctx.pairing_product_is_one([(A, vk["h_gamma_g2"]), (G1.aneg(vk["g_gamma_g1"]), B)])
    return ok1 and ok2


def verify_embedded(curve, vk, proof, public_inputs):
    """The same two checks with every G2 operation carried out on E(Fq12) after the twist embedding — needs no Fq2
    arithmetic for the curve, only its Fq12 tower (used for BLS12-377, whose Fq2 is Fq[u]/(u^2+5))."""
    from .curves import Group
    from .fields import FqOps
    from .pairing import Fq12Ctx
    G1 = Group(FqOps(curve.q), curve.b1, None)
    ctx = Fq12Ctx(curve)
    A, B, C = proof
    q = vk["query"]
    assert len(q) == len(public_inputs) + 1
    acc = G1.to_jac(q[0])
    for x, P in zip(public_inputs, q[1:]):
        acc = G1.add(acc, G1.mul(G1.to_jac(P), x % curve.r))
    vk_x = G1.to_affine(acc)
    tw = ctx.twist
    ok1 = ctx.pairing_product_is_one([
        (vk["g_alpha_g1"], tw(vk["h_beta_g2"])),
        (vk_x, tw(vk["h_gamma_g2"])),
        (C, tw(vk["h_g2"])),
        (G1.aneg(G1.aadd(A, vk["g_alpha_g1"])), ctx.padd(tw(B), tw(vk["h_beta_g2"]))),
    ], embedded=True)
    ok2 = ctx.pairing_product_is_one([(A, tw(vk["h_gamma_g2"])), (G1.aneg(vk["g_gamma_g1"]), tw(B))], embedded=True)
    return ok1 and ok2
