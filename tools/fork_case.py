import os, sys
sys.path.insert(0, "tests")
import fuzz_parity as F
for env in ({}, {"HPRLP_NO_SMALL_POWER": "1"}, {"HPRLP_NO_SMALL": "1"}):
    r = F.one(326, 1416, 1015, 15031, 1e-6, env)
    print(env, r["status"], r["iters"], "%.2e %.2e" % (r["rel"], r["dobj"]), flush=True)
for env in ({}, {"HPRLP_NO_SMALL_POWER": "1"}):
    r = F.one(1238, 1736, 6920, 12022, 1e-6, env)
    print(env, r["status"], r["iters"], "%.2e %.2e" % (r["rel"], r["dobj"]), flush=True)
