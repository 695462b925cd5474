#!/bin/bash
# Launch shape of spread_2d_w8_group_kernel at config 2: waves per workgroup (NUFFT_GROUP_NW) x points staged per wave
# (NUFFT_GROUP_STAGE) -- LDS per workgroup 37 KB + NW x (3.4 KB at 32 staged points, 1.7 KB at 16): 12 x 32 (product) = 78 KB, two
# workgroups per CU; 8 x 16 = 51 KB, three. Spread stage by HIP events, each shape twice. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/group_shape_experiment.txt
: > $OUT
for shape in "12 32" "8 16" "8 32" "12 16" "6 16" "10 16"; do
  set -- $shape
  bash tools/variant_build.sh gshape nufft_kernels.hip "-DNUFFT_GROUP_NW=$1 -DNUFFT_GROUP_STAGE=$2" > /dev/null 2>&1 || { echo "build $shape failed" | tee -a $OUT; continue; }
  for rep in 1 2; do
    echo "NW=$1 STAGE=$2: $(NUFFT_PKG=/tmp/variants/gshape python tools/stage_times.py type_1 1024,1024 1e7 1e-6 "" --one-call 2>&1 | tail -1)" | tee -a $OUT
  done
done
