#!/bin/bash
# Layered tile lists (tiled.h: kTileLayers): tile width with layers -- the width Solver::choose_sb_rows picks against the other
# one forced -- on banded ladder points.   bash tools/ab_layers.sh "banded_1000000_20_2000 ..."
export HPRLP_TEST_HOOKS=1  # the switches below are test hooks (csrc/env.h)
points=${1:-banded_1000000_20_2000}
for p in $points; do
  for e in "" "HPRLP_TILE_COLS=2048" "HPRLP_TILE_COLS=1024"; do
    env $e timeout -k 10 240 python bench.py --ladder-point $p --steps 50 --warmup 10 2>/dev/null | python3 tools/ab_forms_line.py "${e:-chosen}"
  done
done
