#!/usr/bin/env python3
"""Compile one HIP translation unit with -Rpass-analysis=kernel-resource-usage and print a per-kernel table
(VGPRs, AGPRs, SGPRs, scratch bytes, occupancy, LDS).  Usage: tools/kres.py zokrates_amd/csrc/curve_bn254.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/tmp/kres.o",
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+: (.*?) \[-Rpass", line) or re.search(r"remark: (.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = re.sub(r"zk::", "", name)
    name = re.sub(r"\(.*", "", name)
    if flt and flt not in name:
        continue
    print("%-70s VGPR %-4s AGPR %-3s SGPR %-4s scratch %-6s occ %-3s LDS %s" % (
        name[:70], r.get("VGPRs"), r.get("AGPRs"), r.get("SGPRs"), r.get("ScratchSize [bytes/lane]"), r.get("Occupancy [waves/SIMD]"),
        r.get("LDS Size [bytes/block]")))
