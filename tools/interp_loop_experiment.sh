#!/bin/bash
# What the pieces of interp_point_kernel cost: builds with pieces left out (NUFFT_INTERP_EXP bits: 1 no LDS cell reads, 2 no
# result stores, 4 no tile load, 8 results stored in sorted order instead of through the point index); interp stage of
# config 3 (2-D) and of the 3-D type-2 cases. Such builds give wrong results: timing only. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/interp_loop_experiment.txt
: > $OUT
for v in ${VARIANTS:-0 1 2 4 8 3 7}; do
  bash tools/variant_build.sh iexp$v nufft_kernels.hip "-DNUFFT_INTERP_EXP=$v" > /dev/null 2>&1 || { echo "build $v failed" | tee -a $OUT; continue; }
  echo "EXP=$v: $(NUFFT_PKG=/tmp/variants/iexp$v python tools/stage_times.py type_2 1024,1024 1e7 1e-6 2>&1 | tail -1)" | tee -a $OUT
  echo "EXP=$v: $(NUFFT_PKG=/tmp/variants/iexp$v python tools/stage_times.py type_2 256,256,256 1e8 1e-4 2>&1 | tail -1)" | tee -a $OUT
  echo "EXP=$v: $(NUFFT_PKG=/tmp/variants/iexp$v python tools/stage_times.py type_2 256,256,256 3e7 1e-6 2>&1 | tail -1)" | tee -a $OUT
done
