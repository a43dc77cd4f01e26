#!/bin/bash
# same-box A/B of library builds on patterns whose rows run the stream kernel's vector mode (rows of 65..4096 entries):
#   make variant NAME=head   (from the tree to compare against);  tools/ab_tail.sh lib/variants/libhprlp_head.so lib/libhprlp.so
P="dense_blocks_tridiag_64,wide_30kx3M_200,assignment_1500,transportation_1000x3000,column_degree_3_wide,facility_location_2000x1000"
for lib in "$@"; do
  echo "== $lib"
  HPRLP_LIB=$PWD/$lib python tools/ab_pb_rows.py $P 512 stream 2>/dev/null | cut -c1-75
done
