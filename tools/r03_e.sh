#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03e; mkdir -p $O
rm -f gpurun_out/full_size_parity.txt
timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.txt 2>&1; tail -14 $O/pytest.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 3000 $O/bench.json; tail -3 $O/bench.err
