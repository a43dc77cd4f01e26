#!/bin/bash
# kernel times of the batched half-step kernels for developer variants of the library: usage  bash tools/c4_variants.sh "ENV=.. ENV=.." ...
export TMPDIR=/tmp
i=0
for v in "$@"; do
  i=$((i+1)); out=/tmp/c4v_$i; rm -rf $out
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/c4_run.py 400 > /dev/null 2> /tmp/c4v_err_$i.txt || { echo "[$v] FAILED"; tail -3 /tmp/c4v_err_$i.txt; continue; }
  echo "== $v"
  python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:4]:
    print("   %-60s calls %5s avg %8.1f us" % (r["Name"].replace("void hprlp::", "").replace("(anonymous namespace)::", "")[:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
