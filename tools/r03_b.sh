#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03b; mkdir -p $O
(cd tools/ubench && ./lds_pattern_bench) > $O/lds_pattern_ubench.txt 2>&1; cat $O/lds_pattern_ubench.txt
timeout 900 python -m pytest tests -m gpu -x -q -k "3d or config4 or fixed_point or crowded or geometry_sweep or stages or fine_grid" > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python3 tools/bench_configs.py 4 4s 3d5 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
cat gpurun_out/full_size_parity.txt
