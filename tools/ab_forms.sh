#!/bin/bash
# Is the kernel form the library chooses the fastest one it has?  Every given ladder point (bench.py: LADDER_POINTS / FAMILY_POINTS /
# banded_<rows>_<per row>_<band>) as chosen and with the tiled forms switched off (HPRLP_NO_TILED=1: stream kernel); half-step
# times and iterations/s.   usage (inside one gpurun call): bash tools/ab_forms.sh "band_2e7 family_cont_like ..."
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
points=${1:-band_2e7}
for p in $points; do
  for e in 0 1; do
    HPRLP_NO_TILED=$e timeout -k 10 240 python bench.py --ladder-point $p --steps 50 --warmup 10 2>/dev/null | python3 tools/ab_forms_line.py $e
  done
done
