#!/usr/bin/env python3
"""Per-half-step HBM bytes of one ladder point from two rocprofv3 --pmc passes (tools/profile_ladder.sh).

A normal half-step is one to three dispatches: [k_far_products] k_tiled_fused<XEpi<false..>> / k_spmv_fused<XEpi<false..>>
[k_long_finish<XEpi..>], or the piece form's k_tiled_part + k_tiled_finish<XEpi<false..>>.  The helper kernels carry no
epilogue name, so the dispatches are walked in order: bytes of helper kernels are held until the next epilogue-bearing kernel
and added to ITS half-step; k_long_finish follows its fused kernel and is added to the half-step just closed.  Only the NORMAL
variants count (XEpi<false / YEpi<false: reference src/cuda_kernels/HPR_cuda_kernels.cu:229-247, 274-295).
bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE counts 64-byte units as 32; profiles/r01_pmc_summary.md)."""
import collections
import csv
import datetime
import glob
import json
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import subprocess
import sys

out, point = sys.argv[1], sys.argv[2]
HELPERS = ("k_far_products", "k_tiled_part")


def per_half(ctr_dir, ctr):
    rows = []
    for f in glob.glob(f"{out}/{point}/{ctr_dir}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == ctr:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], float(r["Counter_Value"])))
    rows.sort()
    acc = {"x": [], "y": []}
    names = {"x": collections.Counter(), "y": collections.Counter()}
    pending, pend_names, last = 0.0, [], None
    for _, name, val in rows:
        short = name.split("(")[0].replace("void hprlp::", "").replace("hprlp::", "")
        if any(h in name for h in HELPERS):
            pending += val
            pend_names.append(short)
            continue
        half = "x" if "XEpi<false" in name else "y" if "YEpi<false" in name else None
        if half and "k_long_finish" in name and last == half and acc[half]:
            acc[half][-1] += val
            names[half][short] += 1
        elif half:
            acc[half].append(val + pending)
            for q in pend_names + [short]:
                names[half][q] += 1
            last = half
        else:
            last = None
        pending, pend_names = 0.0, []
    return {h: (sum(v) / len(v) if v else None) for h, v in acc.items()}, {h: dict(c) for h, c in names.items()}, {h: len(v) for h, v in acc.items()}


fetch, names, cnt = per_half("fetch", "FETCH_SIZE")
write, _, _ = per_half("write", "WRITE_SIZE")
rec = {}
try:
    rec = json.load(open(f"{out}/{point}_record.json"))[point]
except Exception as e:  # noqa: BLE001
    rec = {"error": str(e)}
try:
    commit = subprocess.check_output(["git", "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL).decode().strip()
except Exception:  # noqa: BLE001
    commit = os.environ.get("HPRLP_COMMIT")  # the GPU box has no .git
ent = {"kernels": rec.get("kernels"), "commit": commit, "date": datetime.date.today().isoformat(), "collected_by": "tools/profile_ladder.sh",
       "correction": "bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (profiles/r01_pmc_summary.md)", "half_steps_counted": cnt, "dispatches_per_half_step": names}
for h, key in (("x", "xhalf"), ("y", "yhalf")):
    if fetch[h] is not None and write[h] is not None:
        ent[f"{key}_fetch_size_kb"] = fetch[h]
        ent[f"{key}_write_size_kb"] = write[h]
        ent[f"{key}_hbm_bytes_per_launch"] = (2 * fetch[h] + write[h]) * 1024
        alg = rec.get(f"{key}_algorithmic_bytes")
        if alg:
            ent[f"{key}_ratio"] = ent[f"{key}_hbm_bytes_per_launch"] / alg
    else:
        ent[f"{key}_hbm_bytes_per_launch"] = None
        ent[f"{key}_ratio"] = None
for k in ("m", "n", "nnz", "xhalf_ms", "yhalf_ms", "xhalf_frac_of_8000", "yhalf_frac_of_8000"):
    ent["record_" + k] = rec.get(k)
json.dump({"ladder:" + point: ent}, open(f"{out}/{point}_traffic.json", "w"), indent=1)
