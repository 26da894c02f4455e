#!/usr/bin/env python3
"""Where the ~0.2-0.3 s of a one-proof process's HIP start go: runtime initialisation (first HIP call), context creation (streams,
events), the first kernel launch (code object load).  One fresh process per line.  python tools/hip_init_probe.py [n]"""
import os
import subprocess
import sys

CHILD = r'''
import ctypes, os, sys, time
sys.path.insert(0, %r)
t0 = time.perf_counter()
from zokrates_amd import native
import numpy as np
t1 = time.perf_counter()
lib = native.default_library()
n = lib.device_count()
t2 = time.perf_counter()
ctx = native.Context(0)
t3 = time.perf_counter()
a = np.zeros(32, dtype=np.uint8); a[0] = 3
out = ctx.field_op(0, 0, "mul", a, a)
t4 = time.perf_counter()
print("import %%6.1f ms | first HIP call (runtime init) %%6.1f ms | zkhip_ctx_create %%6.1f ms | first kernel (code object load) %%6.1f ms | queues=%%s" %% (
    1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3), os.environ.get("GPU_MAX_HW_QUEUES")), flush=True)
os._exit(0)
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for q in ("4", "16"):
        env = dict(os.environ, GPU_MAX_HW_QUEUES=q)
        p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=300)
        print((p.stdout.strip() or p.stderr[-300:]))
