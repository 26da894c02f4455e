#!/usr/bin/env python3
"""Copies the counter files of a GPU session (gpurun_out/<tag>/pmc_traffic.json, pmc_valu.json: written on the GPU box by
tools/pmc_traffic.py / pmc_valu.py with the fingerprint of the sources they ran on) into profiles/, after checking that the
fingerprint IS this tree's, and adds the commit they belong to.  usage: tools/adopt_evidence.py <tag> | --rekey"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zokrates_amd.build import csrc_hash  # noqa: E402

now = csrc_hash()
if sys.argv[1] == "--rekey":
    # the fingerprint's DEFINITION changed (zokrates_amd/build.py: the host-only readers no longer count); a file stamped under the
    # old definition keeps its meaning only if this tree, hashed the old way, is still what it was taken from
    old = csrc_hash(with_host_only=True)
    for name in ("pmc_traffic.json", "pmc_valu.json"):
        path = os.path.join(ROOT, "profiles", name)
        doc = json.load(open(path))
        if doc.get("csrc_hash") == now:
            print(name, "already carries", now)
            continue
        if doc.get("csrc_hash") != old:
            sys.exit("%s was taken from sources %s; this tree is %s under the old definition: not re-keyed" % (name, doc.get("csrc_hash"), old))
        doc["rekeyed_from"] = {"csrc_hash": old, "why": "fingerprint definition: csrc/ingest.hip, ingest.h, emu.h (no kernels, no launches) left out; "
                               "same kernel sources, checked by hashing this tree under the old definition"}
        doc["csrc_hash"] = now
        json.dump(doc, open(path, "w"), indent=1)
        print("profiles/" + name, "re-keyed", old, "->", now)
    sys.exit(0)
tag = sys.argv[1]
head = subprocess.run(["git", "rev-parse", "HEAD"], cwd=ROOT, capture_output=True, text=True).stdout.strip()
dirty = bool(subprocess.run(["git", "status", "--porcelain", "--", "zokrates_amd/csrc", "include"], cwd=ROOT, capture_output=True, text=True).stdout.strip())
for name in ("pmc_traffic.json", "pmc_valu.json"):
    src = os.path.join(ROOT, "gpurun_out", tag, name)
    if not os.path.exists(src):
        print(name, "absent in", tag)
        continue
    doc = json.load(open(src))
    if doc.get("csrc_hash") != now:
        sys.exit("%s was taken from sources %s, this tree is %s: not adopted" % (name, doc.get("csrc_hash"), now))
    doc["git_head"] = head + (" + uncommitted changes to csrc/" if dirty else "")
    doc["session"] = tag
    json.dump(doc, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print("profiles/" + name, "<-", tag, "csrc_hash", now, "head", doc["git_head"])
