#!/bin/bash
# same-box A/B of environment switches on the phases of a config-5 run: usage  bash tools/ab_phase.sh "ENV=1" "" "ENV=1" ""
for spec in "$@"; do
  env $spec python bench.py --no-cpu --no-side --no-ladder --no-solve --steps 30 --warmup 5 > /tmp/p.json 2>/dev/null
  python - "$spec" <<'PY'
import json, sys
d = json.load(open("/tmp/p.json")); sp = d.get("spmv_only") or {}
print("%-28s it/s %6.0f  power %.4f s  scaling %.4f s  spmv AT %.4f A %.4f ms" % (sys.argv[1] or "(default)", d["value"], d["phases_s"]["power_iteration"], d["phases_s"]["scaling"], sp.get("AT_y_ms", 0), sp.get("A_xhat_ms", 0)), flush=True)
PY
done
