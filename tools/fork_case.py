"""A fuzz case whose iteration count differs from the oracle's: the same LP with the single-launch power iteration off, with the
single-workgroup kernels off (regular kernels), and at a higher iteration limit -- does the difference follow a kernel, or is it a
fork at a thresholded restart decision that every summation order takes differently?
usage: python tools/fork_case.py m n nnz seed [max_iter]"""
import os

os.environ.setdefault("HPRLP_TEST_HOOKS", "1")  # the HPRLP_* switches used here are test hooks (csrc/env.h)
import sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import fuzz_parity as F  # noqa: E402

m, n, nnz, seed = (int(v) for v in sys.argv[1:5])
max_iter = int(sys.argv[5]) if len(sys.argv) > 5 else 200000
os.dup2(2, 1)
for env in ({}, {"HPRLP_NO_SMALL_POWER": "1"}, {"HPRLP_NO_SMALL": "1"}, {"HPRLP_NO_SMALL": "1", "HPRLP_NO_GRAPH": "1"}):
    r = F.one(m, n, nnz, seed, 1e-6, env, max_iter)
    print("FORK", (m, n, nnz, seed), env, r["status"], "iterations (gpu, oracle)", r["iters"], "rel %.2e dobj %.2e" % (r["rel"], r["dobj"]), file=sys.stderr, flush=True)
