#!/bin/bash
# Collects SQ / LDS / cache counters of the config-5 kernels in separate rocprofv3 --pmc passes (per-kernel means).
# usage (inside one gpurun call, from the repo root):  bash tools/pmc_passes.sh TAG [bench args...]  ->  gpurun_out/pmc_TAG/
tag=${1:-run}; shift
out=$PWD/gpurun_out/pmc_$tag
mkdir -p "$out"
export TMPDIR=/tmp
args=${@:---no-cpu --no-side --no-solve --steps 10 --warmup 3}
i=0
while read -r ctrs; do
  [ -z "$ctrs" ] && continue
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$out/p$i" -- python3 bench.py $args > /dev/null 2> "$out/p$i.err" || echo "pass $i ($ctrs) failed" >> "$out/failed.txt"
done <<'LIST'
SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS
SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_SMEM SQ_LDS_UNALIGNED_STALL SQ_LDS_ADDR_CONFLICT SQ_INSTS_WAVE32_LDS
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum
TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
GRBM_GUI_ACTIVE GRBM_COUNT
LIST
python3 - "$out" <<'PY'
import csv, glob, collections, sys
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{out}/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void hprlp::", "").replace("hprlp::", "")
        if "tiled_fused" in k or "far_products" in k or "spmv_fused" in k or k.startswith("kb_") or "k_mid" in k:
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(f"{out}/summary.csv", "w") as g:
    g.write("kernel,counter,launches,mean\n")
    for k in sorted(acc):
        for c in sorted(acc[k]):
            v = acc[k][c]
            g.write('"%s",%s,%d,%.1f\n' % (k, c, len(v), sum(v) / len(v)))
print(open(f"{out}/summary.csv").read()[:6000])
PY
rm -rf "$out"/p[0-9]*/
