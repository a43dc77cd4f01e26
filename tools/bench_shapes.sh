for w in c5_eighth c5_quarter c5_shard_like c5_small; do
  timeout -k 10 200 python bench.py --no-cpu --no-side --no-solve --steps 100 --workload $w > gpurun_out/r02_shape_$w.json 2> gpurun_out/r02_shape_$w.err || echo "FAILED $w"
  python - <<PY
import json
d=json.load(open("gpurun_out/r02_shape_$w.json"))
r=d["roofline"]
print("$w", "it/s %.0f" % d["value"], "x %.4f ms y %.4f ms frac %.3f" % (r["avg_launch_ms"], r["yhalf_avg_launch_ms"], r["frac"]), r["kernel"][:30], d.get("spmv_only"))
PY
done
