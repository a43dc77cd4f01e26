#!/bin/bash
# Memory-path counters (L1 <-> L2 latency, TLB, L2 -> fabric queues, TA stalls) of the tiled kernels for one library variant, in
# separate rocprofv3 --pmc passes (the TA_* set is split over two passes: in one, rocprofv3 aborted with "Request exceeds the
# capabilities of the hardware to collect" -- round 3, gpurun_out/pmcmem_*/p7.err).  usage (inside one gpurun call, repo root):  bash tools/pmc_mem.sh NAME [workload]   (NAME: base | variant)
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
name=${1:-base}; wl=${2:-c5}
if [ "$name" = base ]; then lib=$PWD/lib/libhprlp.so; else lib=$PWD/lib/variants/libhprlp_$name.so; fi
out=$PWD/gpurun_out/pmcmem_$name
mkdir -p "$out"
export TMPDIR=/tmp HPRLP_LIB=$lib
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/p$i" -- python3 bench.py --no-cpu --no-side --no-solve --steps 8 --warmup 2 --workload $wl > /dev/null 2> "$out/p$i.err" || echo "pass $i ($ctrs) failed" >> "$out/failed.txt"
done <<'LIST'
TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TCP_LATENCY_sum TCP_TOTAL_ACCESSES_sum TCP_GATE_EN1_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum
TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum
TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_TAG_STALL_sum TCC_CYCLE_sum
TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum
TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE
TCP_RFIFO_STALL_CYCLES_sum TCP_LFIFO_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_BUSY_avr
LIST
python3 - "$out" "$name" <<'PY'
import csv, glob, collections, sys
out, name = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void hprlp::", "").replace("hprlp::", "")
        if "tiled_fused" in k or "tiled_part" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/summary.csv", "w") as g:
    g.write("variant,kernel,counter,launches,mean\n")
    for k in sorted(acc):
        if "XEpi<false" not in k and "YEpi<false" not in k and "tiled_part" not in k:
            continue
        for c in sorted(acc[k]):
            v = acc[k][c]
            g.write('%s,"%s",%s,%d,%.4g\n' % (name, k, c, len(v), sum(v) / len(v)))
print(open(f"{out}/summary.csv").read())
PY
rm -rf "$out"/p[0-9]*/
