#!/bin/bash
cd $GRAFT_REPO_ROOT
(cd tools/ubench && ./mfma_spread_bench) 2>&1 | tee gpurun_out/mfma_spread_ubench.txt
