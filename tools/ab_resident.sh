#!/bin/bash
# same-box A/B of library builds on the on-chip resident kernel (gx1-sized piece): tools/ab_lib_*.so against the product
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for round in 1 2 3; do
  for lib in "" $(ls tools/ab_lib_*.so 2>/dev/null); do
    echo -n "round $round lib=${lib:-product}: "
    EVP_TIMING_LIB=$lib timeout 300 python tools/piece_timing.py 320 384 960 7 2>&1 | tail -1
  done
done
