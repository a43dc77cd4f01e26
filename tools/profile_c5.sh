#!/bin/bash
# Collects, on the GPU box, what profiles/ holds for config 5: the rocprofv3 kernel-trace statistics of the bench command and the
# FETCH_SIZE / WRITE_SIZE counters in separate passes (gpurun refuses counter passes combined with other trace domains).
# usage (from the repo root, inside one gpurun call):  bash tools/profile_c5.sh v7   ->  gpurun_out/prof_v7/
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
set -e
tag=${1:-run}
out=$PWD/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 bench.py --no-cpu --no-side --no-solve --steps 100 --warmup 20 > "$out/bench_line.json" 2> /dev/null
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$out/fetch" -- python3 bench.py --no-cpu --no-side --no-solve --steps 20 --warmup 5 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$out/write" -- python3 bench.py --no-cpu --no-side --no-solve --steps 20 --warmup 5 > /dev/null 2>&1
cp "$out"/trace/*/*kernel_stats.csv "$out/kernel_stats.csv"
python3 - "$out" <<'PY'
import csv, glob, collections, sys, json
out = sys.argv[1]
res = {}
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    acc = collections.defaultdict(list)
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    with open(f"{out}/pmc_{name}.csv", "w") as g:
        g.write("Kernel_Name,Launches,Mean_%s_KB\n" % ctr)
        for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1])):
            g.write('"%s",%d,%.1f\n' % (k, len(v), sum(v) / len(v)))
            if "k_tiled_fused" in k or "k_spmv_fused" in k:
                res.setdefault(k.split("(")[0].replace("void hprlp::", ""), {})[ctr] = sum(v) / len(v)
for k, d in res.items():
    if "FETCH_SIZE" in d and "WRITE_SIZE" in d:
        d["bytes_per_launch"] = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024
# the remainder pre-pass (k_far_products) belongs to every tiled launch: a half-step = pre-pass + fused kernel
far = {}
for name, ctr in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
    vals = []
    for f in glob.glob(f"{out}/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr and "k_far_products" in r["Kernel_Name"]:
                vals.append(float(r["Counter_Value"]))
    if vals:
        far[ctr] = sum(vals) / len(vals)
if "FETCH_SIZE" in far and "WRITE_SIZE" in far:
    far["bytes_per_launch"] = (2 * far["FETCH_SIZE"] + far["WRITE_SIZE"]) * 1024
    res["k_far_products"] = far
# with the producer-side hand-off (kernels.h: FarPush; default, HPRLP_NO_FAR_PUSH=1 disables) the normal half-steps run WITHOUT
# a pre-pass of their own: the k_far_products launches left belong to the residual / power-iteration SpMVs
import os
handoff = os.environ.get("HPRLP_NO_FAR_PUSH", "0") != "1"
calls = {}
def half(tag):
    k = [v for n, v in res.items() if "k_tiled_fused" in n and tag in n and "bytes_per_launch" in v]
    return (k[0]["bytes_per_launch"] + (0.0 if handoff else far.get("bytes_per_launch", 0.0))) if k else None
res["_half_steps"] = {"xhalf_hbm_bytes_per_launch": half("XEpi<false"), "yhalf_hbm_bytes_per_launch": half("YEpi<false"),
                      "handoff": handoff, "launches": dict(calls), "note": "fused kernel (+ remainder pre-pass unless handed over by the producing half-step); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (profiles/r01_pmc_summary.md: calibration)"}
json.dump(res, open(f"{out}/pmc_traffic_per_kernel.json", "w"), indent=1)
print(json.dumps(res, indent=1))
# the entry bench.py reads (profiles/pmc_traffic.json: copy gpurun_out/prof_<tag>/pmc_traffic_entry.json into it under the
# workload's name): which kernel the counters were taken on, at which commit, when
import datetime, subprocess
xk = [n for n in res if "k_tiled_fused" in n and "XEpi<false" in n]
hs = res["_half_steps"]
key = None
if xk:
    # k_tiled_fused<XEpi<false, true>, REP, PUSH, NARROW>: the key names PUSH and NARROW (REP depends on the matrix only)
    args = [a.strip() for a in xk[0].rstrip(">").split(">,")[-1].split(",")]
    key = ("k_tiled_fused<XEpi<false, true>, *, %s, %s>" % (args[1], args[2] if len(args) > 2 else "false")) if "XEpi<false, true>" in xk[0] and len(args) >= 2 else xk[0]
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:
    commit = os.environ.get("HPRLP_COMMIT")  # the GPU box has no .git: pass HPRLP_COMMIT=$(git rev-parse --short HEAD) into the call
entry = {"xhalf_hbm_bytes_per_launch": hs["xhalf_hbm_bytes_per_launch"], "yhalf_hbm_bytes_per_launch": hs["yhalf_hbm_bytes_per_launch"],
         "kernel_key": key, "kernel_name": xk[0] if xk else None, "commit": commit, "date": datetime.date.today().isoformat(),
         "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; factor 2 calibrated on k_stream<1>/k_vec_scale (profiles/r01_pmc_summary.md)",
         "collected_by": "tools/profile_c5.sh"}
json.dump(entry, open(f"{out}/pmc_traffic_entry.json", "w"), indent=1)
PY
rm -rf "$out/trace" "$out/fetch" "$out/write"
