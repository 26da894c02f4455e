"""Reproduces the round-1 GPUTEST failure: two zkhip_prove_gm17_partial calls on the SAME shard, n = 300.
Raw XYZZ records differ between calls on a real GPU (within-bucket order is decided by atomics); canonical records do not.
    [ZKHIP_PKG=zokrates_amd_v1] python tools/repro_partial_records.py      (zokrates_amd_v1 = the round-1 package + library)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import importlib
native = importlib.import_module(os.environ.get("ZKHIP_PKG", "zokrates_amd") + ".native")   # ZKHIP_PKG=zokrates_amd_v1: the round-1 build
from oracle import cpu, gm17
from oracle.fields import BN254
from test_gm17 import circuit, csr_of, le

ctx = native.Context(0)
print(ctx.describe(), "library:", native.DEFAULT_LIB)
curve = BN254
cs, z = circuit(curve, 300, 61, extra_public=1)
tox = gm17.Toxic.from_seed(curve)
mats = [csr_of(cs.A), csr_of(cs.B), csr_of(cs.C)]
dcs = native.ConstraintSystem(ctx, curve.curve_id, cs.n, cs.l, cs.w, mats)
oc = cpu.Circuit.from_csr(curve.curve_id, cs.n, cs.l, cs.w, mats)
raw = cpu.Gm17ProvingKey.setup(oc, cpu.gm17_toxic_bytes(tox)).serialize()
shard = native.ProvingKey(ctx, curve.curve_id, raw, rank=1, world=4, scheme="gm17")
recs = [native.prove_gm17_partial(ctx, shard, dcs, le(z), 5, 6, 7).tobytes() for _ in range(8)]
distinct = len(set(recs))
print("8 calls on one shard -> %d distinct partial records (%d bytes each)" % (distinct, len(recs[0])))
for k in range(5):
    span = slice(k * len(recs[0]) // 6, (k + 1) * len(recs[0]) // 6)
print("REPRO_RESULT distinct=%d" % distinct)
