#!/bin/bash
# Timeline of ONE whole solve() of a bench workload (default config 5) to 1e-4: every kernel in launch order from a
# rocprofv3 --kernel-trace, consecutive launches of one kernel folded into a line (count, busy time, idle gaps before / inside),
# beside the library's own phase table (HPRLP_TIMING=1).  usage (repo root, inside one gpurun call):
#   bash tools/solve_timeline.sh TAG [workload]   ->  gpurun_out/timeline_TAG.txt
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
tag=${1:-run}; wl=${2:-c5}
out=$PWD/gpurun_out/timeline_$tag
mkdir -p "$out"
export TMPDIR=/tmp HPRLP_TIMING=1 HPRLP_SOLVE_REPS=2   # the SECOND solve is the one listed (warm process)
rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -- python3 tools/solve_c5.py $wl > /dev/null 2> "$out/stderr.txt"
python3 - "$out" > "$PWD/gpurun_out/timeline_$tag.txt" <<'PY'
import csv, glob, sys
out = sys.argv[1]
f = max(glob.glob(f"{out}/trace/**/*kernel_trace.csv", recursive=True), key=lambda p: __import__("os").path.getsize(p))
rows = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))))
starts = [i for i, r in enumerate(rows) if "k_check_columns" in r[2]]
rows = rows[starts[-1] - 1 if starts and starts[-1] > 0 else 0:]
t0 = rows[0][0]
def short(k):
    return k.replace("void hprlp::", "").replace("hprlp::", "").replace("(anonymous namespace)::", "").split("(")[0][:100]
groups = []
for s, e, k in rows:
    k = short(k)
    if groups and groups[-1]["k"] == k:
        g = groups[-1]
        g["inner_gap"] += s - g["end"]; g["busy"] += e - s; g["end"] = e; g["n"] += 1
    else:
        groups.append({"k": k, "start": s, "end": e, "busy": e - s, "n": 1, "gap_before": s - (groups[-1]["end"] if groups else s), "inner_gap": 0})
print("# start_ms  calls  busy_ms  gap_before_ms  gaps_inside_ms  kernel")
# fold the iteration loops (alternating kernels) by printing only groups over 0.3 ms busy or 0.3 ms gap individually, the rest summed
small_busy = small_gap = 0; small_n = 0
for g in groups:
    if g["busy"] > 1e6 or g["gap_before"] > 1e6 or g["inner_gap"] > 1e6:
        if small_n:
            print("          ...  %5d  %8.3f  %8.3f   (short launches folded)" % (small_n, small_busy / 1e6, small_gap / 1e6)); small_busy = small_gap = small_n = 0
        print("%9.3f  %5d  %8.3f  %8.3f  %8.3f  %s" % ((g["start"] - t0) / 1e6, g["n"], g["busy"] / 1e6, g["gap_before"] / 1e6, g["inner_gap"] / 1e6, g["k"]))
    else:
        small_n += g["n"]; small_busy += g["busy"]; small_gap += g["gap_before"] + g["inner_gap"]
if small_n:
    print("          ...  %5d  %8.3f  %8.3f   (short launches folded)" % (small_n, small_busy / 1e6, small_gap / 1e6))
print("# total span %.3f ms, busy %.3f ms" % ((rows[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in rows) / 1e6))
print("# --- library phase lines (stderr) ---")
for l in open(f"{out}/stderr.txt"):
    if "hprlp" in l.lower() or "solve_c5" in l:
        print("# " + l.rstrip()[:300])
PY
rm -rf "$out/trace"
wc -l "$PWD/gpurun_out/timeline_$tag.txt"
