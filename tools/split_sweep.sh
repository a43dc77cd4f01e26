export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ktrace.sh c5e --no-cpu --no-side --no-solve --steps 50 --warmup 10 --workload c5_eighth 2>&1 | head -12
for sp in 1 2 3 6 8; do
  HPRLP_TILE_SPLIT=$sp timeout -k 10 200 python bench.py --no-cpu --no-side --no-solve --steps 100 --workload c5_eighth > gpurun_out/r02_c5e_split$sp.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_c5e_split$sp.json")); r=d["roofline"]
print("split $sp: x %.4f y %.4f" % (r["avg_launch_ms"], r["yhalf_avg_launch_ms"]), "spmv A %.4f" % d["spmv_only"]["A_xhat_ms"])
PY
done
