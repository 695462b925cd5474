#!/bin/bash
# Regenerates the raw material of profiles/ on the GPU box (run through gpurun):
# bench line, rocprofv3 kernel stats of the bench command, the two separate PMC passes.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err
rm -rf gpurun_out/prof_bench2 gpurun_out/pmc2_fetch gpurun_out/pmc2_write
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench2 -o runc --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/prof_bench2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc2_fetch -o runc --output-format csv -- python3 tools/profile_run.py --steps 3 > gpurun_out/pmc2_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc2_write -o runc --output-format csv -- python3 tools/profile_run.py --steps 3 > gpurun_out/pmc2_write.log 2>&1
find gpurun_out/prof_bench2 gpurun_out/pmc2_fetch gpurun_out/pmc2_write -name "*.csv" | head -20
cat gpurun_out/bench_r01.json
