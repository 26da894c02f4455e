"""Unsaturated-limb field (zokrates_amd/csrc/fieldu.cuh) against the saturated Montgomery field of field.cuh, compiled for
the host with g++: every operation the MSM kernels use (conversions, mul, mul2, add/sub with bias, weak reduction, the
zero test, Fq2 mul/sqr, XYZZ mixed add / add / doubling incl. the exceptional branches) on both curves."""
import os
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "zokrates_amd", "csrc")


@pytest.mark.parametrize("flags", [["-DZK_FQ2_KARATSUBA=1", "-DZK_LAZY_Y3_G1=1"], []], ids=["every-variant", "product-defaults"])
@pytest.mark.parametrize("name", ["fieldu_ops", "fieldu_curve", "fieldu_fused"])
def test_unsaturated_field_host(name, flags, tmp_path):
    """-DZK_CHECK_OVERFLOW: every product also runs its column sums in 128 bits and aborts if one leaves 64 bits — the bounds
    arguments of fieldu.cuh / ec.cuh (lazy negations, fused sums) are checked on every multiplication of these runs."""
    exe = str(tmp_path / name)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-Wno-unknown-pragmas", "-DZK_CHECK_OVERFLOW"] + flags + [ "-I", CSRC, os.path.join(HERE, "host", name + ".cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "0 failures" in out.stdout
