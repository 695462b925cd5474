#!/bin/bash
# widths 9..16: wide kernels (default) against the thread-per-point tile kernels (NUFFT_HIP_NO_WIDE=1)
for t in type_1 type_2; do
  for args in "1e7 1024,1024 $t 1e-9 c128" "1e7 1024,1024 $t 1e-12 c128" "1e7 128,128,128 $t 1e-9 c128" "1e7 128,128,128 $t 1e-12 c128"; do
    echo "--- wide"; python tools/time_case3.py $args
    echo "--- generic"; NUFFT_HIP_NO_WIDE=1 python tools/time_case3.py $args
  done
done
