"""Imports bench.py's generator helpers for the tests without running the benchmark."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_argv = sys.argv
sys.argv = ["bench.py"]
try:
    _spec = importlib.util.spec_from_file_location("hprlp_bench", os.path.join(ROOT, "bench.py"))
    _b = importlib.util.module_from_spec(_spec)
    _spec.loader.exec_module(_b)
finally:
    sys.argv = _argv
gen_banded = _b.gen_banded
banded_lp = _b.banded_lp
bytes_per_iteration = _b.bytes_per_iteration
