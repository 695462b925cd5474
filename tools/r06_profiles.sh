#!/bin/bash
# Regenerates the r06 artefacts under profiles/ in one gpurun call (results land in gpurun_out/profiles_r06/, to be
# copied into profiles/): GPU suite first (the numbers belong to a green tree), the bench line, rocprofv3 kernel stats
# of the bench command at its default step counts, counter summaries (one --pmc pass per counter set, --kernel-trace
# only) of configs 2, 3, 4, one config-5 item, the 3-D default-tolerance case per subproblem (M = 3e7) and over stacks
# (M = 1e7), 3-D type 2, the mixed-radix FFT passes (240^3 modes), the complex128 3-D spreader and interpolation over stacks; HIP-event
# stage times of configs 1-5; per-kernel averages of configs 3, 4, the 3-D cases, a non-power-of-two grid and complex128;
# the non-power-of-two and complex128 tables; the reference harness's eight cases.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/profiles_r06; rm -rf $O; mkdir -p $O
rm -f gpurun_out/full_size_parity.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=6 > $O/r06_gpu_suite.txt 2>&1; tail -12 $O/r06_gpu_suite.txt
cp gpurun_out/full_size_parity.txt $O/r06_full_size_parity.txt 2>/dev/null
timeout 1200 python3 bench.py > $O/r06_bench.json 2> $O/bench.err; cut -c1-400 $O/r06_bench.json
rm -rf gpurun_out/prof_bench6
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_bench6 -o runc --output-format csv -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > gpurun_out/prof_bench6.log 2>&1
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras   (MI355X, r06)"; python3 tools/kstats.py gpurun_out/prof_bench6 14; tail -c 3000 gpurun_out/prof_bench6.log | grep -o '"kernel_ms": [0-9.]*' | sed 's/^/# same run, HIP events in bench.py: /'; } > $O/r06_bench_kernel_stats.txt
cat $O/r06_bench_kernel_stats.txt
pmc() { bash tools/pmc_kernels.sh $1 "$2" > $O/r06_pmc_$1.txt 2>&1; grep -A2 "$3" $O/r06_pmc_$1.txt | cut -c1-260; }
pmc cfg2 "--type type_1 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" "spread_2d_w8_group"
pmc cfg3 "--type type_2 --grid 1024,1024 --M 1e7 --tol 1e-6 --one-call" "interp_point"
pmc cfg4 "--type type_1 --grid 256,256,256 --M 1e8 --tol 1e-4 --one-call" "spread_dense3"
pmc cfg5_item "--type type_1 --grid 512,512 --M 1e6 --tol 1e-6 --one-call" "spread_2d_w8_group"
pmc w8_3d "--type type_1 --grid 256,256,256 --M 3e7 --tol 1e-6 --one-call" "spread_patch3\|spread_stack3"
pmc w8_3d_stacks "--type type_1 --grid 256,256,256 --M 1e7 --tol 1e-6 --one-call" "spread_stack3"
pmc 3d_type2 "--type type_2 --grid 256,256,256 --M 1e7 --tol 1e-6 --one-call" "interp_point"
pmc mixfft_240 "--type type_1 --grid 240,240,240 --M 1e7 --tol 1e-6 --one-call" "fft_mixed"
pmc c128_3d_stacks "--type type_1 --grid 256,256,256 --M 1e7 --tol 1e-6 --double" "spread_wave3_stack"
pmc c128_3d_type2_stacks "--type type_2 --grid 256,256,256 --M 1e7 --tol 1e-6 --double" "interp_point"
python3 tools/bench_configs.py 2 3 4 4t2 5 5s 5op 1 2>&1 | grep -v amdgpu > $O/r06_configs.txt; cat $O/r06_configs.txt
for cfg in "type_2 1024,1024 1e7 1e-6 cfg3 --one-call" "type_1 256,256,256 1e8 1e-4 cfg4 --one-call" "type_1 256,256,256 3e7 1e-6 w8_3d --one-call" "type_1 256,256,256 1e7 1e-6 w8_3d_stacks --one-call" "type_2 256,256,256 1e7 1e-6 t2_3d --one-call" "type_1 240,240,240 1e7 1e-6 nonpow2_240 --one-call" "type_2 240,240,240 1e7 1e-6 nonpow2_240_t2 --one-call" "type_1 960,960 1e7 1e-6 nonpow2_960 --one-call" "type_1 256,256,256 1e7 1e-6 c128_3d --double" "type_1 256,256,256 1e7 1e-9 c128_3d_tol1e-9 --double" "type_2 256,256,256 1e7 1e-6 c128_3d_type2 --double" "type_2 256,256,256 1e7 1e-9 c128_3d_type2_tol1e-9 --double" "type_1 384,384,384 1e8 1e-4 big_384 --one-call"; do
  set -- $cfg
  rm -rf gpurun_out/prof6_$5
  timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof6_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 3 $6 > gpurun_out/prof6_$5.log 2>&1
  echo "== $5: $1 grid $2 M=$3 tol=$4 $6 (rocprofv3 --kernel-trace --stats, 3 steps)" >> $O/r06_configs_kernel_stats.txt
  python3 tools/kstats.py gpurun_out/prof6_$5 12 | grep -v "at::native" >> $O/r06_configs_kernel_stats.txt
done
cat $O/r06_configs_kernel_stats.txt
python3 tools/bench_nonpow2.py --rocfft 2>&1 | grep -v amdgpu > $O/r06_nonpow2.txt; cut -c1-200 $O/r06_nonpow2.txt
python3 tools/bench_c128.py 2d 3d 2>&1 | grep -v amdgpu > $O/r06_c128.txt; cut -c1-200 $O/r06_c128.txt
python3 tools/bench_big3d.py 2>&1 | grep -v amdgpu > $O/r06_big3d.txt; cat $O/r06_big3d.txt
python3 tools/bench_reference_cases.py 2>&1 | grep -v "amdgpu\|RCCL\|HIP version\|ROCm version\|Hostname\|Librccl" > $O/r06_reference_benchmark_cases.txt; cat $O/r06_reference_benchmark_cases.txt
