#!/bin/bash
for t in type_1 type_2; do
  for args in "1e7 1048576 $t 1e-6 c64" "1e7 4096 $t 1e-6 c64" "1e7 1048576 $t 1e-9 c128" "1e5 4096 $t 1e-9 c128"; do
    echo "--- line"; python tools/time_case3.py $args
    echo "--- generic"; NUFFT_HIP_NO_LINE=1 python tools/time_case3.py $args
  done
done
