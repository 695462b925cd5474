#!/bin/bash
# The randomised -m gpu tests under further seeds (the r04 kernels: 3-D float w = 7 / 8 fixed point with bounds, the
# persistent fp64 fallback, strength statistics), with the larger point sets. Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NUFFT_TEST_BIGM=${NUFFT_TEST_BIGM:-1}
SEEDS="${SEEDS:-301 302 303 304 305 306 307 308 309 310 311 312 313 314 315 316}" bash tools/soak.sh 2>&1 | tee gpurun_out/soak_r04.txt
