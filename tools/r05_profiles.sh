#!/bin/bash
# Regenerates the r05 artefacts under profiles/ in one gpurun call (results land in gpurun_out/profiles_r05/, to be
# copied into profiles/): GPU suite first (the numbers belong to a green tree), the bench line, rocprofv3 kernel stats
# of the bench command at its default step counts, counter summaries (one --pmc pass per counter set, --kernel-trace
# only) of configs 2, 3, 4, the 3-D default-tolerance case per subproblem (M = 3e7) and over stacks (M = 1e7),
# HIP-event stage times of configs 1-5, per-kernel averages of configs 3, 4 and the 3-D default-tolerance case, the 3-D
# tolerance sweep with whole-output errors, the reference harness's eight cases, the clustered / radial tables.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r05; rm -rf $O; mkdir -p $O
rm -f gpurun_out/full_size_parity.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > $O/r05_gpu_suite.txt 2>&1; tail -12 $O/r05_gpu_suite.txt
cp gpurun_out/full_size_parity.txt $O/r05_full_size_parity.txt 2>/dev/null
timeout 1200 python3 bench.py > $O/r05_bench.json 2> $O/bench.err; cut -c1-400 $O/r05_bench.json
rm -rf gpurun_out/prof_bench5
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench5 -o runc --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/prof_bench5.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras   (MI355X, r05)"; python3 tools/kstats.py gpurun_out/prof_bench5 14; tail -c 3000 gpurun_out/prof_bench5.log | grep -o '"kernel_ms": [0-9.]*' | sed 's/^/# same run, HIP events in bench.py: /'; } > $O/r05_bench_kernel_stats.txt
cat $O/r05_bench_kernel_stats.txt
bash tools/pmc_kernels.sh cfg2 "--type type_1 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" > $O/r05_pmc_cfg2.txt 2>&1
grep -A2 "spread_2d_w8_group" $O/r05_pmc_cfg2.txt | cut -c1-260
bash tools/pmc_kernels.sh cfg3 "--type type_2 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" > $O/r05_pmc_cfg3.txt 2>&1
grep -A2 "interp_point" $O/r05_pmc_cfg3.txt | cut -c1-260
bash tools/pmc_kernels.sh cfg4 "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" > $O/r05_pmc_cfg4.txt 2>&1
grep -A2 "spread_dense3" $O/r05_pmc_cfg4.txt | cut -c1-260
bash tools/pmc_kernels.sh w8r05 "--type type_1 --grid 256,256,256 --M 3e7 --tol 1e-6 --one-call" > $O/r05_pmc_w8_3d.txt 2>&1
grep -A2 "spread_patch3\|spread_stack3" $O/r05_pmc_w8_3d.txt | cut -c1-260
bash tools/pmc_kernels.sh w8stack "--type type_1 --grid 256,256,256 --M 1e7 --tol 1e-6 --one-call" > $O/r05_pmc_w8_3d_stacks.txt 2>&1
grep -A2 "spread_stack3" $O/r05_pmc_w8_3d_stacks.txt | cut -c1-260
bash tools/pmc_kernels.sh t2w8 "--type type_2 --grid 256,256,256 --M 1e7 --tol 1e-6 --one-call" > $O/r05_pmc_3d_type2.txt 2>&1
grep -A2 "interp_point" $O/r05_pmc_3d_type2.txt | cut -c1-260
python3 tools/bench_configs.py 2 3 4 4t2 5 5s 5op 1 2>&1 | grep -v amdgpu > $O/r05_configs.txt; cat $O/r05_configs.txt
for cfg in "type_2 1024,1024 1e7 1e-6 cfg3" "type_1 256,256,256 1e8 1e-4 cfg4" "type_1 256,256,256 3e7 1e-6 w8_3d" "type_1 256,256,256 1e7 1e-6 w8_3d_stacks" "type_2 256,256,256 1e7 1e-6 t2_3d"; do
  set -- $cfg
  rm -rf gpurun_out/prof5_$5
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof5_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 3 --one-call > gpurun_out/prof5_$5.log 2>&1
  echo "== $5: $1 grid $2 M=$3 tol=$4 (rocprofv3 --kernel-trace --stats, 3 x nufft_hip_execute_with_points)" >> $O/r05_configs_kernel_stats.txt
  python3 tools/kstats.py gpurun_out/prof5_$5 12 | grep -v "at::native" >> $O/r05_configs_kernel_stats.txt
done
cat $O/r05_configs_kernel_stats.txt
{ python3 tools/tol_sweep_3d.py; python3 tools/tol_sweep_3d.py --M 100000000 --tols 1e-5,1e-6; python3 tools/tol_sweep_3d.py --M 10000000 --tols 1e-4,1e-5,1e-6; python3 tools/tol_sweep_3d.py --grid 128 --M 800000 --tols 1e-5,1e-6 --steps 10; } 2>&1 | grep -v amdgpu > $O/r05_3d_tol_sweep.txt; cat $O/r05_3d_tol_sweep.txt
python3 tools/bench_reference_cases.py 2>&1 | grep -v "amdgpu\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/r05_reference_benchmark_cases.txt; cat $O/r05_reference_benchmark_cases.txt
