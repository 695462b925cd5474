#!/bin/bash
# The randomised -m gpu tests under further seeds (r06: smooth fine grids on the mixed-radix passes, complex128 / w = 9..16
# over stacks, the two-level sort with partial super-tiles ride along in the geometry sweeps). Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06
export NUFFT_TEST_BIGM=${NUFFT_TEST_BIGM:-1}
SEEDS="${SEEDS:-801 802 803 804 805 806 807 808 809 810}" bash tools/soak.sh 2>&1 | tee gpurun_out/r06/soak_r06.txt
