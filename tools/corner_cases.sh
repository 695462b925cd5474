#!/bin/bash
# The corners outside the specialised float w <= 8 kernels: 1-D, and w = 9..16 (fp64, tol < 1e-7)
for t in type_1 type_2; do
  python tools/time_case3.py 1e7 1048576 $t 1e-6 c64
  python tools/time_case3.py 1e7 4096 $t 1e-6 c64
  python tools/time_case3.py 1e7 1048576 $t 1e-9 c128
  python tools/time_case3.py 1e7 4096 $t 1e-9 c128
  python tools/time_case3.py 1e7 1024,1024 $t 1e-6 c128
  python tools/time_case3.py 1e7 1024,1024 $t 1e-9 c128
  python tools/time_case3.py 1e7 1024,1024 $t 1e-12 c128
  python tools/time_case3.py 1e7 128,128,128 $t 1e-6 c128
  python tools/time_case3.py 1e7 128,128,128 $t 1e-9 c128
  python tools/time_case3.py 1e7 128,128,128 $t 1e-12 c128
done
