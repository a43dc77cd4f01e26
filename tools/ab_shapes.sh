#!/bin/bash
# same-box A/B of kernel-selection switches on the shard-shaped workloads: usage  bash tools/ab_shapes.sh "w1 w2" "ENV1=.. ENV2=.." ...
wls=$1; shift
for w in $wls; do
  for v in "$@"; do
    env $v timeout -k 10 200 python bench.py --no-cpu --no-side --no-solve --steps 60 --warmup 10 --workload $w > /tmp/ab.json 2>/dev/null || { echo "$w [$v] FAILED"; continue; }
    python - "$w" "$v" <<'PY'
import json, sys
d=json.load(open("/tmp/ab.json")); r=d["roofline"]; sp=d.get("spmv_only") or {}
print("%-14s %-34s it/s %7.0f  x %.4f  y %.4f ms  spmv AT %.4f A %.4f" % (sys.argv[1], sys.argv[2], d["value"], r["avg_launch_ms"], r["yhalf_avg_launch_ms"], sp.get("AT_y_ms", 0), sp.get("A_xhat_ms", 0)))
PY
  done
done
