#!/bin/bash
# HBM traffic of the normal half-steps of every ladder point (bench.py: LADDER_POINTS), from FETCH_SIZE / WRITE_SIZE collected
# in SEPARATE rocprofv3 --pmc passes (gpurun refuses counter passes combined with other trace domains; the guide's HBM section:
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 on gfx950, calibrated in profiles/r01_pmc_summary.md).
# usage (repo root, inside one gpurun call):  bash tools/profile_ladder.sh r04 [point ...]  ->  gpurun_out/ladder_r04/
#   ladder_traffic.json : entries "ladder:<point>" to merge into profiles/pmc_traffic.json (tools/merge_ladder_traffic.py)
#   <point>_kernel_stats.csv : rocprofv3 --kernel-trace --stats of the same command
set -e
tag=${1:-run}
shift || true
points=${@:-band_2e6 band_2e7 band_6e7 block_angular_2e7 unstructured_4e7 expander_7e6}
out=$PWD/gpurun_out/ladder_$tag
mkdir -p "$out"
export TMPDIR=/tmp
for p in $points; do
  echo "[profile_ladder] $p" >&2
  rocprofv3 --kernel-trace --stats --output-format csv -d "$out/$p/trace" -- python3 bench.py --ladder-point $p --steps 50 --warmup 10 > "$out/${p}_record.json" 2> "$out/${p}_stderr.txt"
  cp "$out/$p"/trace/*/*kernel_stats.csv "$out/${p}_kernel_stats.csv"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/$p/fetch" -- python3 bench.py --ladder-point $p --steps 20 --warmup 5 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/$p/write" -- python3 bench.py --ladder-point $p --steps 20 --warmup 5 > /dev/null 2>&1
  python3 tools/ladder_traffic.py "$out" "$p"
  rm -rf "$out/$p"
done
python3 - "$out" <<'PY'
import glob, json, sys
out = sys.argv[1]
res = {}
for f in sorted(glob.glob(f"{out}/*_traffic.json")):
    res.update(json.load(open(f)))
json.dump(res, open(f"{out}/ladder_traffic.json", "w"), indent=1)
print(json.dumps({k: {q: v[q] for q in ("xhalf_hbm_bytes_per_launch", "yhalf_hbm_bytes_per_launch", "xhalf_ratio", "yhalf_ratio")} for k, v in res.items()}, indent=1))
PY
