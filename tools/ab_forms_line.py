"""One line per record of `bench.py --ladder-point` (stdin) for tools/ab_forms.sh."""
import json
import sys

label = {"1": "stream", "0": "chosen"}.get(sys.argv[1], sys.argv[1])
for k, v in json.load(sys.stdin).items():
    a = v["kernels"].split("; A^T:")[0]
    print("%-24s %-7s x %.4f ms (%.3f)  y %.4f ms (%.3f)  %6.0f it/s | %s" % (
        k, label, v["xhalf_ms"], v["xhalf_frac_of_8000"], v["yhalf_ms"], v["yhalf_frac_of_8000"], v["iterations_per_s"],
        a[:34] + " .. " + a[-44:]))
