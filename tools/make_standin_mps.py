#!/usr/bin/env python3
"""Writes the shape-matched stand-ins of BASELINE.json's configs 1-3 plus two ladder-class LPs as .mps(.gz) files into a directory,
so that tools/run_mps_dir.py can be shown end to end on the GPU box (real Netlib / Mittelmann files are not available offline):
    python tools/make_standin_mps.py DIR  &&  python tools/run_mps_dir.py DIR --out table.md
The writer is the test suite's (tests/test_mps.py: write_mps); the product only reads."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_mps import write_mps  # noqa: E402
from conftest import lpgen  # noqa: E402

out = sys.argv[1]
os.makedirs(out, exist_ok=True)
shutil.copy(os.path.join(ROOT, "tests", "data", "lp_small.mps"), os.path.join(out, "c1_model.mps"))  # content of the reference's data/model.mps
write_mps(os.path.join(out, "c2_25fv47_like.mps"), lpgen.c2_25fv47_like())
write_mps(os.path.join(out, "c3_pds20_like.mps.gz"), lpgen.c3_pds20_like())
write_mps(os.path.join(out, "banded_100k_2e6nnz.mps.gz"), lpgen.banded_lp(100_000, 100_000, 20, 1_000, 5))
write_mps(os.path.join(out, "block_angular_100x500x1000.mps.gz"), lpgen.block_angular_lp(100, 500, 1000, 12, 20, 20, 600, 9))
for name, make in lpgen.FAMILIES_SMALL.items():   # the four Mittelmann-family constructions at test size (round 4)
    write_mps(os.path.join(out, f"family_{name}.mps"), make())
print("wrote", sorted(os.listdir(out)))
