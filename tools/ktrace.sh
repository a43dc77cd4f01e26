#!/bin/bash
# rocprofv3 kernel-trace statistics of a short bench run: gpurun_out/ktrace_TAG.csv (top kernels by total time)
tag=${1:-run}; shift
out=$PWD/gpurun_out/ktrace_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=${@:---no-cpu --no-side --no-solve --steps 50 --warmup 10}
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py $args > "$out/bench_line.json" 2> /dev/null
cp "$out"/trace/*/*kernel_stats.csv "$PWD/gpurun_out/ktrace_$tag.csv"
rm -rf "$out/trace"
python3 - "$PWD/gpurun_out/ktrace_$tag.csv" <<'PY'
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:14]:
    print("%-90s calls %5s avg %10.1f us" % (r["Name"].replace("void hprlp::", "").replace("hprlp::", "")[:90], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
