#!/bin/bash
# The two 36-case randomised parity sweeps (tests/fuzz_parity.py, seeds 31001 and 41001) with the logs of EVERY case compared
# (FUZZ_TRACE_ALL=1: the fork rule's growth figures also for the runs that keep the oracle's count).  On a GPU box:
#   bash tools/fuzz_sweeps.sh   ->  gpurun_out/r05_fuzz_parity_seed<seed>.txt
mkdir -p gpurun_out
for seed in 31001 41001; do
  FUZZ_TRACE_ALL=1 python tests/fuzz_parity.py $seed 2> gpurun_out/r05_fuzz_parity_seed$seed.txt
  echo "seed $seed: exit $?" >> gpurun_out/r05_fuzz_parity_seed$seed.txt
  tail -2 gpurun_out/r05_fuzz_parity_seed$seed.txt
done
