#!/bin/bash
# same-box A/B over (library variant, environment) pairs: usage  bash tools/ab_env.sh "w1 w2" "name|ENV1=.. ENV2=.." ...   (name: base or a variant)
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
wls=$1; shift
for w in $wls; do
  for spec in "$@"; do
    v=${spec%%|*}; envs=${spec#*|}; [ "$envs" = "$spec" ] && envs=""
    if [ "$v" = base ]; then lib=lib/libhprlp.so; else lib=lib/variants/libhprlp_$v.so; fi
    env HPRLP_LIB=$PWD/$lib $envs timeout -k 10 300 python bench.py --no-cpu --no-side --no-solve --steps 100 --warmup 20 --workload $w > /tmp/ab.json 2>/tmp/ab.err || { echo "$w [$spec] FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$w" "$spec" <<'PY'
import json, sys
d=json.load(open("/tmp/ab.json")); r=d["roofline"]; sp=d.get("spmv_only") or {}
print("%-12s %-52s it/s %7.0f  x %.4f (%.3f)  y %.4f ms  spmv AT %.4f A %.4f" % (sys.argv[1], sys.argv[2], d["value"], r["avg_launch_ms"], r["frac"], r["yhalf_avg_launch_ms"], sp.get("AT_y_ms", 0), sp.get("A_xhat_ms", 0)), flush=True)
PY
  done
done
