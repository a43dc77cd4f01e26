#!/bin/bash
# same-box A/B of developer variants of the library (make variant NAME=..): usage  bash tools/ab_libs.sh "w1 w2" base name1 name2 ...
# ("base" = lib/libhprlp.so); prints it/s and the half-step times per workload and variant
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
wls=$1; shift
for w in $wls; do
  for v in "$@"; do
    if [ "$v" = base ]; then lib=lib/libhprlp.so; else lib=lib/variants/libhprlp_$v.so; fi
    HPRLP_LIB=$PWD/$lib timeout -k 10 300 python bench.py --no-cpu --no-side --no-solve --steps 60 --warmup 10 --workload $w > /tmp/ab.json 2>/tmp/ab.err || { echo "$w [$v] FAILED"; tail -3 /tmp/ab.err; continue; }
    python - "$w" "$v" <<'PY'
import json, sys
d=json.load(open("/tmp/ab.json")); r=d["roofline"]; sp=d.get("spmv_only") or {}
print("%-12s %-16s it/s %7.0f  x %.4f  y %.4f ms  spmv AT %.4f A %.4f" % (sys.argv[1], sys.argv[2], d["value"], r["avg_launch_ms"], r["yhalf_avg_launch_ms"], sp.get("AT_y_ms", 0), sp.get("A_xhat_ms", 0)), flush=True)
PY
  done
done
