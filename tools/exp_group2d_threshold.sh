#!/bin/bash
# The density from which the 2-D float spreader sorts by cell and accumulates runs (wave8_use_group: 0.5 points per fine cell, r01)
# re-checked on the final kernels: spread stage, GROUP_OFF / GROUP_ON, one-call and two-call entries. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
O=gpurun_out/r05/group2d_threshold.txt
: > $O
for M in 5e5 1e6 1.5e6 2e6 3e6 4e6; do
  for t in GROUP_OFF GROUP_ON; do
    echo "$t one-call: $(python tools/stage_times.py type_1 1024,1024 $M 1e-6 $t --one-call 2>&1 | tail -1)" | tee -a $O
    echo "$t two-call: $(python tools/stage_times.py type_1 1024,1024 $M 1e-6 $t 2>&1 | tail -1)" | tee -a $O
  done
done
