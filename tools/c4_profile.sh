#!/bin/bash
# kernel trace + FETCH/WRITE counters of the batched half-step kernels (config 4): gpurun_out/c4_TAG/
tag=${1:-run}
out=$PWD/gpurun_out/c4_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 tools/c4_run.py 600 > /dev/null 2> "$out/run.err"
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d "$out/$c" -- python3 tools/c4_run.py 100 > /dev/null 2>&1
done
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
for r in list(csv.DictReader(open(f"{out}/kernel_stats.csv")))[:6]:
    print("%-70s calls %6s avg %9.1f us" % (r["Name"].replace("void hprlp::", "")[:70], r["Calls"], float(r["AverageNs"]) / 1e3))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{ctr}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "kb_half" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(ctr, k.replace("void hprlp::", ""), "launches", len(v), "mean KB %.0f" % (sum(v) / len(v)))
PY
tail -2 "$out/run.err"
rm -rf "$out/trace" "$out/FETCH_SIZE" "$out/WRITE_SIZE"
