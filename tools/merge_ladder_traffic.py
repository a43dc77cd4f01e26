#!/usr/bin/env python3
"""Merge gpurun_out/ladder_<tag>/ladder_traffic.json (tools/profile_ladder.sh) into profiles/pmc_traffic.json, stamping the
commit the counters were taken at (the GPU box has no .git)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
commit = sys.argv[2] if len(sys.argv) > 2 else subprocess.check_output(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"]).decode().strip()
dst = os.path.join(ROOT, "profiles", "pmc_traffic.json")
cur = json.load(open(dst))
new = json.load(open(src))
for k, v in new.items():
    if not v.get("commit"):
        v["commit"] = commit
    cur[k] = v
json.dump(cur, open(dst, "w"), indent=1)
print("merged", sorted(new))
