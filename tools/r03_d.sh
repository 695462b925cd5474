#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03d; mkdir -p $O
(cd tools/ubench && ./lds_pattern_bench) > $O/lds_pattern_ubench.txt 2>&1; cat $O/lds_pattern_ubench.txt
timeout 1200 python -m pytest tests -m gpu -x -q -k "3d or fixed_point or crowded or geometry_sweep or fine_grid" > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
python3 tools/bench_configs.py 4 4s 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
bash tools/pmc_kernels.sh cfg4d "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/pmc_cfg4_dense.txt 2>&1
grep -A3 "dense3" $O/pmc_cfg4_dense.txt
