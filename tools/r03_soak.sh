#!/bin/bash
# The three randomised -m gpu tests under further seeds (SEEDS), with the larger point sets (NUFFT_TEST_BIGM). Run through gpurun.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export NUFFT_TEST_BIGM=${NUFFT_TEST_BIGM:-1}
SEEDS="${SEEDS:-201 202 203 204 205 206 207 208 209 210 211 212 213 214 215 216 217 218 219 220 221 222 223 224}" bash tools/soak.sh 2>&1 | tee gpurun_out/soak_r03b.txt
