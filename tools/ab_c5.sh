#!/bin/bash
# same-box A/B of library variants on a bench workload: tools/ab_c5.sh "default head default head" [workload]
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
libs=${1:-default}
wl=${2:-c5}
for l in $libs; do
  if [ "$l" = default ]; then unset HPRLP_LIB; else export HPRLP_LIB=$PWD/lib/variants/libhprlp_$l.so; fi
  timeout -k 10 200 python bench.py --workload $wl --no-cpu --no-side --no-solve --steps 100 --warmup 20 2>/dev/null | python3 -c "
import json,sys
d=json.load(sys.stdin)
print('%-8s %-10s %.1f it/s  x %.4f ms (%.3f)  y %.4f ms' % ('$wl', '$l', d['value'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['yhalf_avg_launch_ms']))
"
done
