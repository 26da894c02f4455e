"""Field and curve constants + Fq2 arithmetic (python big ints).  TEST ORACLE ONLY.

Restates [UPSTREAM] ark-ff 0.3.0 / ark-bn254 0.3.0 / ark-bls12-381 0.3.0 parameters
(SURVEY.md App. A.1, A.4, App. C).  In-tree pins for the same numbers:
  * BN254 r: /root/reference/zokrates_proof_systems/src/scheme/groth16.rs:157
  * BN254 q, twist b', generators: /root/reference/zokrates_proof_systems/src/solidity.rs:24-26,430-441
"""
from dataclasses import dataclass


@dataclass(frozen=True)
class Curve:
    name: str            # zokrates curve name (zokrates_common/src/constants.rs)
    curve_id: int        # zkhip C-ABI id
    r: int               # scalar field modulus
    q: int               # base field modulus
    fr_generator: int    # Fr::GENERATOR (coset generator g)
    two_adicity: int
    two_adic_root: int   # GENERATOR^((r-1)/2^S)
    b1: int              # G1: y^2 = x^3 + b1
    b2: tuple            # G2: y^2 = x^3 + b2 over Fq2 = Fq[u]/(u^2+1)
    g1: tuple            # affine generator
    g2: tuple            # affine generator ((x0,x1),(y0,y1))
    fq_bytes: int
    fr_bytes: int = 32


BN254_R = 21888242871839275222246405745257275088548364400416034343698204186575808495617
BN254_Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583

BN254 = Curve(
    name="bn128", curve_id=0,
    r=BN254_R, q=BN254_Q,
    fr_generator=5, two_adicity=28,
    two_adic_root=19103219067921713944291392827692070036145651957329286315305642004821462161904,
    b1=3,
    b2=(19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690),
    g1=(1, 2),
    g2=((10857046999023057135944570762232829481370756359578518086990519993285655852781,
         11559732032986387107991004021392285783925812861821192530917403151452391805634),
        (8495653923123431417604973247489272438418190587263600148770280649306958101930,
         4082367875863433681332203403145435568316851327593401208105741076214120093531)),
    fq_bytes=32,
)

BLS381_R = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
BLS381_Q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab

BLS12_381 = Curve(
    name="bls12_381", curve_id=1,
    r=BLS381_R, q=BLS381_Q,
    fr_generator=7, two_adicity=32,
    two_adic_root=0x16a2a19edfe81f20d09b681922c813b4b63683508c2280b93829971f439f0d2b,
    b1=4, b2=(4, 4),
    g1=(0x17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb,
        0x08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1),
    g2=((0x024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8,
         0x13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e),
        (0x0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801,
         0x0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be)),
    fq_bytes=48,
)

CURVES = {"bn128": BN254, "bls12_381": BLS12_381}

# BLS12-377: not a curve libzkhip proves over — present only so that the oracle's GM17 verification equations can be
# pinned on the reference's one golden (proof, vk, inputs) triple, which is over this curve
# (/root/reference/zokrates_stdlib/tests/tests/snark/gm17.json; [UPSTREAM] ark-bls12-377 0.3.0 parameters).
BLS12_377 = Curve(
    name="bls12_377", curve_id=-1,
    r=8444461749428370424248824938781546531375899335154063827935233455917409239041,
    q=258664426012969094010652733694893533536393512754914660539884262666720468348340822774968888139573360124440321458177,
    fr_generator=22, two_adicity=47, two_adic_root=0,
    b1=1, b2=None, g1=None, g2=None, fq_bytes=48,
)


def inv(a, p):
    return pow(a, p - 2, p)


# ---- Fq2 = Fq[u]/(u^2+1), elements are (c0, c1) ----
def f2_add(a, b, q): return ((a[0] + b[0]) % q, (a[1] + b[1]) % q)
def f2_sub(a, b, q): return ((a[0] - b[0]) % q, (a[1] - b[1]) % q)
def f2_neg(a, q): return ((-a[0]) % q, (-a[1]) % q)
def f2_mul(a, b, q):
    return ((a[0] * b[0] - a[1] * b[1]) % q, (a[0] * b[1] + a[1] * b[0]) % q)
def f2_sqr(a, q): return f2_mul(a, a, q)
def f2_inv(a, q):
    n = inv((a[0] * a[0] + a[1] * a[1]) % q, q)
    return (a[0] * n % q, (-a[1]) * n % q)
def f2_scalar(a, k, q): return (a[0] * k % q, a[1] * k % q)


class FqOps:
    """Uniform interface so curve code is generic over Fq (ints) and Fq2 (pairs)."""
    def __init__(self, q): self.q = q; self.zero = 0; self.one = 1
    def add(self, a, b): return (a + b) % self.q
    def sub(self, a, b): return (a - b) % self.q
    def mul(self, a, b): return a * b % self.q
    def neg(self, a): return (-a) % self.q
    def inv(self, a): return inv(a, self.q)
    def is_zero(self, a): return a % self.q == 0
    def small(self, k): return k % self.q


class Fq2Ops:
    def __init__(self, q): self.q = q; self.zero = (0, 0); self.one = (1, 0)
    def add(self, a, b): return f2_add(a, b, self.q)
    def sub(self, a, b): return f2_sub(a, b, self.q)
    def mul(self, a, b): return f2_mul(a, b, self.q)
    def neg(self, a): return f2_neg(a, self.q)
    def inv(self, a): return f2_inv(a, self.q)
    def is_zero(self, a): return a[0] % self.q == 0 and a[1] % self.q == 0
    def small(self, k): return (k % self.q, 0)


class SplitMix64:
    """Deterministic PRNG shared by oracle, C++ oracle and bench."""
    M = (1 << 64) - 1

    def __init__(self, seed): self.s = seed & self.M

    def next(self):
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def field(self, p):
        """Uniform element of Z_p by rejection sampling on bit_length(p) bits."""
        bits = p.bit_length()
        nl = (bits + 63) // 64
        while True:
            v = 0
            for i in range(nl):
                v |= self.next() << (64 * i)
            v &= (1 << bits) - 1
            if v < p:
                return v
