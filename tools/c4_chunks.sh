#!/bin/bash
# Config 4 per chunk width of the batched panels (HPRLP_BATCH_CHUNK): kernel times of the half-step kernels (rocprofv3 kernel trace)
# and their HBM traffic (FETCH_SIZE / WRITE_SIZE, separate --pmc passes).  usage (inside one gpurun call):  bash tools/c4_chunks.sh "64 8"
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
export TMPDIR=/tmp
for c in $1; do
  out=/tmp/c4c_$c; rm -rf $out
  HPRLP_BATCH_CHUNK=$c timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -- python3 tools/c4_run.py 400 > /dev/null 2> $out.err || { echo "[$c] kernel trace FAILED"; tail -3 $out.err; continue; }
  for ctr in FETCH_SIZE WRITE_SIZE; do
    HPRLP_BATCH_CHUNK=$c timeout -k 10 200 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/$ctr -- python3 tools/c4_run.py 100 > /dev/null 2> $out.err || { echo "[$c] $ctr FAILED"; tail -3 $out.err; }
  done
  echo "== chunk $c"
  python3 - $out <<'PY'
import collections, csv, glob, sys
out = sys.argv[1]
f = glob.glob(out + "/kt/**/*kernel_stats.csv", recursive=True)[0]
def short(n): return n.replace("void hprlp::", "").replace("(anonymous namespace)::", "").split("(")[0][:48]
for r in list(csv.DictReader(open(f)))[:3]:
    print("   %-48s calls %5s avg %8.1f us" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    for g in glob.glob(out + "/" + ctr + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(g)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    if "kb_half" not in k or ", true" in k.split("<")[1][5:]:
        continue
    fe = sum(acc[k]["FETCH_SIZE"]) / max(1, len(acc[k]["FETCH_SIZE"])); wr = sum(acc[k]["WRITE_SIZE"]) / max(1, len(acc[k]["WRITE_SIZE"]))
    print("   %-48s FETCH %9.0f KB  WRITE %9.0f KB  -> HBM bytes (2*F + W)*1024 = %.1f MB" % (k, fe, wr, (2 * fe + wr) * 1024 / 1e6))
PY
done
