#!/bin/bash
# small problems (the reference benchmark's sizes): sort launch shape and tile choices
run() { echo "--- $1"; env $1 python tools/time_case3.py 2e5 256,256 type_1 1e-6; env $1 python tools/time_case3.py 2e5 256,256 type_2 1e-6; env $1 python tools/time_case3.py 8e5 128,128,128 type_1 1e-6; env $1 python tools/time_case3.py 8e5 128,128,128 type_2 1e-6; }
run "X=1"
run "NUFFT_HIP_SORT_MINPB=2048"
run "NUFFT_HIP_SORT_MINPB=1024"
run "NUFFT_HIP_SORT_MINPB=1024 NUFFT_HIP_STAGED_SCATTER=0"
run "NUFFT_HIP_STAGED_SCATTER=0"
run "NUFFT_HIP_BIG_T2_ALWAYS=1"
