#!/bin/bash
# rocprofv3 kernel statistics of one pattern of tools/ab_pb_rows.py under forced forms:  tools/ktrace_forms.sh PATTERN HEIGHTS FORM [FORM..]
export TMPDIR=/tmp
pat=$1; rows=$2; shift 2
for f in "$@"; do
  out=$PWD/gpurun_out/prof_$f
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -- python3 tools/ab_pb_rows.py $pat $rows $f > gpurun_out/prof_$f.txt 2> /dev/null
  cp "$out"/*/*kernel_stats.csv gpurun_out/prof_$f.csv; rm -rf "$out"
  echo "== $f"; cat gpurun_out/prof_$f.txt
  python3 - gpurun_out/prof_$f.csv <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:12]: print("  %-120s %6s total %10.1f us avg %8.1f"%(r["Name"][:120].replace("void hprlp::","").replace("hprlp::",""), r["Calls"], float(r["TotalDurationNs"])/1e3, float(r["AverageNs"])/1e3))
P
done
