"""The bounds arguments of the unsaturated field INSIDE the kernels.  fieldu.cuh's products add up to four limb products per
column in 64-bit accumulators with no carries, and round 4 feeds them operands that skipped their carry round (the negated y of
a negative bucket digit, PPP in the fused Y3, the one-round X3 numerator).  Whether a column can leave 64 bits is a pencil-and-
paper argument in the comments; `-DZK_CHECK_OVERFLOW` turns it into a run-time check on host builds: every product also runs
its column sums in 128 bits and aborts when one does not fit.  Here the TEST-ONLY emulator is built that way and the kernel
tests that reach every product of the hot path (full proofs on both curves, both groups, skewed scalars: heavy buckets,
doublings, cancellations) run on it in a child process — an abort fails the test."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(ROOT, "zokrates_amd", "csrc")


def test_kernels_keep_every_column_inside_64_bits(tmp_path):
    d = os.path.join(HERE, "_emu", "ovf")
    lib = os.path.join(d, "libzkhip_emu.so")
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if os.path.isfile(os.path.join(CSRC, f))]
    if not os.path.exists(lib) or any(os.path.getmtime(s) > os.path.getmtime(lib) for s in srcs):
        os.makedirs(d, exist_ok=True)
        script = open(os.path.join(HERE, "_emu", "build_emu.sh")).read().replace('SRC="$HERE/../../zokrates_amd/csrc"', 'SRC="%s"' % CSRC)
        with open(os.path.join(d, "build_emu.sh"), "w") as f:
            f.write(script)
        os.chmod(os.path.join(d, "build_emu.sh"), 0o755)
        subprocess.check_call([os.path.join(d, "build_emu.sh")], env=dict(os.environ, EMU_FLAGS="-DZK_CHECK_OVERFLOW"))
    # (test_bound_key.py: the two transforms over G1 points of zkhip_pk_bind_r1cs — negated XYZZ points, the two-bit ladder, general
    # additions of stored sums — run their products through the same check)
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(HERE, "test_emu_kernels.py"), os.path.join(HERE, "test_bound_key.py"), "-x", "-q", "-k",
                        "test_prove_matches_oracle or test_msm_skewed_scalars or test_ntt_three_passes or test_bound_key_proves_the_same_bytes or "
                        "test_bound_key_with_heavy_columns or test_bound_key_over_two_and_three", "-p", "no:cacheprovider"],
                       env=dict(os.environ, ZKHIP_EMU_LIBRARY=lib), capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "overflows 64 bits" not in p.stderr + p.stdout, (p.stdout[-2000:], p.stderr[-2000:])
