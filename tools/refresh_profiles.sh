#!/bin/bash
# Regenerates the raw material of profiles/ on the GPU box (run through gpurun):
# bench line, rocprofv3 kernel stats of the bench command, the two separate PMC passes
# (never combined with other trace domains), then the summaries.   usage: refresh_profiles.sh r02
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python3 bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err
rm -rf gpurun_out/prof_bench2 gpurun_out/pmc2_fetch gpurun_out/pmc2_write
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench2 -o runc --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras > gpurun_out/prof_bench2.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d gpurun_out/pmc2_fetch -o runc --output-format csv -- python3 tools/profile_run.py --steps 3 --one-call > gpurun_out/pmc2_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d gpurun_out/pmc2_write -o runc --output-format csv -- python3 tools/profile_run.py --steps 3 --one-call > gpurun_out/pmc2_write.log 2>&1
python3 tools/summarize_profiles.py $TAG
cp gpurun_out/bench_$TAG.json profiles/${TAG}_bench.json
mkdir -p gpurun_out/profiles_$TAG && cp profiles/${TAG}_* profiles/pmc_spread_traffic.json gpurun_out/profiles_$TAG/
cat gpurun_out/bench_$TAG.json | cut -c1-600
