"""An independent point set for the MSM: the snarkjs powers-of-tau file the reference ships for its JS tests
(/root/reference/zokrates_js/tests/powersOfTau5_0000.ptau -> tests/golden/ptau5_points.json, by tests/golden/make_golden.py;
SURVEY.md §8c lists it among the fixtures).  None of these points came out of this repository's setup code or its oracle.
The file is the ceremony's starting file (tau = 1), which makes it an UNFRIENDLY base set rather than a random one: the 63
tauG1 entries are all the generator (every addition inside a bucket is a doubling), and the Lagrange-basis sections are 57
points at infinity, six more generators and one domain's worth of distinct points, with the same indices in G1 and G2 — so
    e(sum c_i L_i G1, G2) = e(G1, sum c_i L_i G2)
ties a device G1 MSM to a device G2 MSM over matching bases through the pairing (the compiled verifier's, csrc/host/verify.cpp)."""
import json
import os
import random
import subprocess

import numpy as np
import pytest

from oracle import cpu, formats
from oracle.curves import groups
from oracle.fields import BN254
from zokrates_amd import native

HERE = os.path.dirname(os.path.abspath(__file__))
CLI_EMU = os.path.join(HERE, "_emu", "zkhip-cli-emu")


def le(vals, nb=32):
    return np.frombuffer(b"".join(int(v).to_bytes(nb, "little") for v in vals), dtype=np.uint8)


@pytest.fixture(scope="module")
def ptau(golden_dir):
    d = json.load(open(os.path.join(golden_dir, "ptau5_points.json")))
    g1 = lambda v: None if v == ["0", "0"] else (int(v[0]), int(v[1]))
    g2 = lambda v: None if v == ["0"] * 4 else ((int(v[0]), int(v[1])), (int(v[2]), int(v[3])))
    return {"tau_g1": [g1(v) for v in d["tau_g1"]], "tau_g2": [g2(v) for v in d["tau_g2"]],
            "lagrange_g1": [g1(v) for v in d["lagrange_g1"]], "lagrange_g2": [g2(v) for v in d["lagrange_g2"]], "q": int(d["q"])}


def test_fixture_points_are_what_the_format_says(ptau):
    """On the curve / on the twist, tau = 1, and the G1 / G2 Lagrange sections describe the same scalars (oracle pairing)."""
    G1, G2 = groups(BN254)
    assert ptau["q"] == BN254.q
    assert all(P == BN254.g1 for P in ptau["tau_g1"]) and all(Q == BN254.g2 for Q in ptau["tau_g2"])
    lg1, lg2 = ptau["lagrange_g1"], ptau["lagrange_g2"]
    assert sum(P is None for P in lg1) == 57 and len({P for P in lg1 if P}) == 65
    for P in lg1:
        assert P is None or G1.on_curve(P)
    for Q in lg2:
        assert Q is None or G2.on_curve(Q)
    # the domains 2^0 .. 2^5 exist in both groups: L_0(1) = 1 (the generator), every other L_i(1) = 0 (infinity), index by index;
    # the 64 distinct points are the G1-only block of the domain 2^6
    assert [P is None for P in lg1[:63]] == [Q is None for Q in lg2]
    assert all(P in (None, BN254.g1) for P in lg1[:63]) and all(Q in (None, BN254.g2) for Q in lg2)
    assert len({P for P in lg1[63:]}) == 64 and None not in lg1[63:]


def _msm_cases(ptau, rnd):
    lg1, lg2 = ptau["lagrange_g1"], ptau["lagrange_g2"]
    r = BN254.r
    full = [rnd.randrange(r) for _ in lg1]
    return {
        "tau_g1: 63 equal bases": (1, ptau["tau_g1"], [rnd.randrange(r) for _ in ptau["tau_g1"]]),
        "tau_g1: equal bases, equal scalars (one bucket per window)": (1, ptau["tau_g1"], [0x1234567] * 63),
        "tau_g2: 32 equal bases": (2, ptau["tau_g2"], [rnd.randrange(r) for _ in ptau["tau_g2"]]),
        "lagrange_g1: infinities, repeats, 64 distinct": (1, lg1, full),
        "lagrange_g1: small scalars": (1, lg1, [rnd.randrange(1 << 12) for _ in lg1]),
        "lagrange_g2": (2, lg2, full[:63]),
    }


def _check_msms(ctx, ptau):
    G1, G2 = groups(BN254)
    rnd = random.Random(505)
    for name, (grp, pts, ks) in _msm_cases(ptau, rnd).items():
        ser = formats.ser_g1 if grp == 1 else formats.ser_g2
        bases = np.frombuffer(b"".join(ser(BN254, P) for P in pts), dtype=np.uint8)
        got = ctx.msm(0, grp, bases, le(ks))
        assert got == cpu.msm(0, grp, bases, le(ks)), name                      # the C++ restatement of ark's Pippenger
        G = G1 if grp == 1 else G2                                              # and the plain sum, in Python
        acc = None
        for P, k in zip(pts, ks):
            if P is not None and k:
                acc = G.aadd(acc, G.amul(P, k))
        assert got == ser(BN254, acc) + (b"\1" if acc is None else b"\0"), name


def test_msm_over_the_fixture_on_the_emulator(ptau):
    from emu_util import emu_library
    ctx = native.Context(0, emu_library())
    try:
        _check_msms(ctx, ptau)
    finally:
        ctx.close()


@pytest.mark.gpu
def test_msm_over_the_fixture_on_the_gpu(ptau, tmp_path):
    ctx = native.Context(0)
    try:
        _check_msms(ctx, ptau)
        # bilinearity through two device MSMs over the matching Lagrange bases and the compiled verifier's pairing
        rnd = random.Random(606)
        ks = [rnd.randrange(BN254.r) for _ in range(63)]
        G1, _ = groups(BN254)
        b1 = np.frombuffer(b"".join(formats.ser_g1(BN254, P) for P in ptau["lagrange_g1"][:63]), dtype=np.uint8)
        b2 = np.frombuffer(b"".join(formats.ser_g2(BN254, Q) for Q in ptau["lagrange_g2"]), dtype=np.uint8)
        s1, s2 = ctx.msm(0, 1, b1, le(ks)), ctx.msm(0, 2, b2, le(ks))
        assert s1[-1] == 0 and s2[-1] == 0
        P = (int.from_bytes(s1[:32], "little"), int.from_bytes(s1[32:64], "little"))
        Q = tuple((int.from_bytes(s2[64 * c:64 * c + 32], "little"), int.from_bytes(s2[64 * c + 32:64 * c + 64], "little")) for c in range(2))
        hx = lambda v: "0x" + int(v).to_bytes(32, "big").hex()
        line = lambda A, B: " ".join([hx(A[0]), hx(A[1]), hx(B[0][0]), hx(B[0][1]), hx(B[1][0]), hx(B[1][1])])
        good, bad = tmp_path / "good.txt", tmp_path / "bad.txt"
        good.write_text(line(P, BN254.g2) + "\n" + line(G1.aneg(BN254.g1), Q) + "\n")
        bad.write_text(line(G1.aadd(P, BN254.g1), BN254.g2) + "\n" + line(G1.aneg(BN254.g1), Q) + "\n")
        exe = os.path.join(os.path.dirname(HERE), "zokrates_amd", "zkhip-cli")
        for path, want in ((good, "ONE"), (bad, "NOT-ONE")):
            r = subprocess.run([exe, "pairing-check", "bn128", str(path)], capture_output=True, text=True)
            assert r.returncode == 0 and r.stdout.strip() == want, (r.stdout, r.stderr)
    finally:
        ctx.close()
