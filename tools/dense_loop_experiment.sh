#!/bin/bash
# What the pieces of the grouped spread_dense3_kernel cost at config 4: builds with pieces of dense3_accumulate left out
# (NUFFT_DENSE_EXP bits: 1 no LDS atomics, 2 no staging reads, 4 no kernel evaluation / staging writes); spread stage
# of config 4 (256^3 modes, M = 1e8, tol 1e-4). Results of such builds are wrong by construction. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
OUT=gpurun_out/r05/dense_loop_experiment.txt
: > $OUT
for v in ${VARIANTS:-0 1 2 4 3 7}; do
  bash tools/variant_build.sh dexp$v nufft_dense3.hip "-DNUFFT_DENSE_EXP=$v" > /dev/null 2>&1 || { echo "build $v failed" | tee -a $OUT; continue; }
  echo "EXP=$v: $(NUFFT_PKG=/tmp/variants/dexp$v python tools/stage_times.py type_1 256,256,256 1e8 1e-4 "" --one-call 2>&1 | tail -1)" | tee -a $OUT
done
