#!/bin/bash
# What the pieces of spread_2d_w8_group_kernel cost: builds with pieces left out (NUFFT_GROUP_EXP bits: 1 no LDS atomics,
# 2 no staging reads, 4 no kernel evaluation / staging writes), spread stage of config 2 under each. The results of
# such a build are wrong by construction: timing only. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/group_loop_experiment.txt
for v in ${VARIANTS:-0 1 2 4 3 7}; do
  bash tools/variant_build.sh gexp$v nufft_kernels.hip "-DNUFFT_GROUP_EXP=$v" > /dev/null 2>&1 || { echo "build $v failed" | tee -a $OUT; continue; }
  echo "EXP=$v: $(NUFFT_PKG=/tmp/variants/gexp$v python tools/stage_times.py type_1 1024,1024 1e7 1e-6 "" --one-call 2>&1 | tail -1)" | tee -a $OUT
done
