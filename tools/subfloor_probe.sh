#!/bin/bash
# Where do the half-steps of the small ladder points spend their time?  rocprofv3 kernel trace (GPU-side duration of every
# kernel) against the HIP-event windows bench.py reports (launch gaps included), plus wave and LDS counters in separate passes.
#   bash tools/subfloor_probe.sh "family_nug_like band_2e6"   ->  gpurun_out/subfloor_<point>.txt
export TMPDIR=/tmp
points=${1:-family_nug_like}
for p in $points; do
  out=$PWD/gpurun_out/subfloor_$p
  rm -rf "$out"; mkdir -p "$out"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --ladder-point $p --steps 200 --warmup 20 > "$out/line.json" 2> /dev/null
  i=0
  for ctrs in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_RD" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" "FETCH_SIZE WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/p$i" -- python3 bench.py --ladder-point $p --steps 20 --warmup 5 > /dev/null 2> "$out/p$i.err" || echo "pass $i failed" >> "$out/failed.txt"
  done
  python3 - "$out" "$p" > "$PWD/gpurun_out/subfloor_$p.txt" <<'PY'
import csv, glob, json, collections, sys
out, p = sys.argv[1], sys.argv[2]
line = json.load(open(f"{out}/line.json"))[p]
print(f"# {p}: {line['m']} x {line['n']}, {line['nnz']} entries; {line['kernels'][:160]}")
print(f"# HIP-event windows (bench.py, mode 1): x-half {1e3*line['xhalf_ms']:.2f} us, y-half {1e3*line['yhalf_ms']:.2f} us; graph replay: {1e6/line['iterations_per_s']:.2f} us per iteration")
for f in glob.glob(f"{out}/trace/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:8]:
        print("kernel %-86s calls %5s avg %8.2f us" % (r["Name"].replace("void hprlp::", "").replace("hprlp::", "")[:86], r["Calls"], float(r["AverageNs"]) / 1e3))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void hprlp::", "").replace("hprlp::", "")
        if "XEpi<false" in k or "YEpi<false" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    print("counters", k[:70], " ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())))
PY
  cat "$PWD/gpurun_out/subfloor_$p.txt"
done
