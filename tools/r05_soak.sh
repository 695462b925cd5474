#!/bin/bash
# The randomised -m gpu tests under further seeds (r05: stacks of tiles forced on over random grids / densities / cuts),
# with the larger point sets. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export NUFFT_TEST_BIGM=${NUFFT_TEST_BIGM:-1}
SEEDS="${SEEDS:-501 502 503 504 505 506 507 508 509 510 511 512}" bash tools/soak.sh 2>&1 | tee gpurun_out/r05/soak_r05.txt
