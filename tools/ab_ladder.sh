#!/bin/bash
# A/B of library variants (lib/variants/libhprlp_<name>.so, `make variant`) on ladder points: half-step times per variant.
# usage: bash tools/ab_ladder.sh "default norem nopush" "unstructured_4e7 band_2e7"
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
libs=${1:-default}
points=${2:-unstructured_4e7}
for p in $points; do
  for l in $libs; do
    if [ "$l" = default ]; then unset HPRLP_LIB; else export HPRLP_LIB=$PWD/lib/variants/libhprlp_$l.so; fi
    timeout -k 10 180 python bench.py --ladder-point $p --steps 50 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
for k,v in d.items(): print('%-20s %-10s x %.4f ms (%.3f)  y %.4f ms (%.3f)  finite %s' % (k, '$l', v['xhalf_ms'], v['xhalf_frac_of_8000'], v['yhalf_ms'], v['yhalf_frac_of_8000'], v['finite']))
"
  done
done
