#!/bin/bash
run() { echo "--- $1"; for M in 5e5 1e6 2e6 4e6; do env $1 python tools/time_case3.py $M 512,512 type_1 1e-6; done; env $1 python tools/time_case3.py 1e6 512,512 type_2 1e-6; }
run "X=1"
run "NUFFT_HIP_SORT_MINPB=1024"
run "NUFFT_HIP_SORT_MINPB=1024 NUFFT_HIP_STAGED_SCATTER=0"
run "NUFFT_HIP_SORT_MINPB=2048"
