#!/bin/bash
# r03, first GPU call: the GPU suite, the headline bench, config-4 counters and kernel stats.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03a; mkdir -p $O
rm -f gpurun_out/full_size_parity.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.txt 2>&1; echo "pytest rc $?" >> $O/pytest.txt
tail -25 $O/pytest.txt
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json
bash tools/pmc_kernels.sh cfg4 "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/pmc_cfg4.txt 2>&1
cat $O/pmc_cfg4.txt
python3 tools/bench_configs.py 2 3 4 5 5op 2>&1 | grep -v amdgpu > $O/configs.txt; cat $O/configs.txt
