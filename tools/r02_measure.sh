#!/bin/bash
# r02 measurements that feed profiles/ (run through gpurun)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O
python3 tools/sparse_crossover.py > $O/r02_sparse_crossover.txt 2>&1
python3 tools/bench_configs.py 2 3 4 4t2 5 5s 5op 1 2d 3d6 2>&1 | grep -v amdgpu > $O/r02_configs.txt
make -C tools/ubench -s all
(cd tools/ubench && ./scatter_write_bench) > $O/r02_scatter_write_ubench.txt 2>&1
(cd tools/ubench && ./lds_atomic_bench) > $O/r02_lds_atomic_ubench.txt 2>&1
bash tools/sweep_group.sh > $O/r02_cfg5_group_sweep.txt 2>&1
for cfg in "type_2 1024,1024 1e7 1e-6 cfg3" "type_1 256,256,256 1e8 1e-4 cfg4" "type_2 256,256,256 1e8 1e-4 cfg4t2"; do
  set -- $cfg
  rm -rf $O/prof_$5
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 3 --one-call > $O/prof_$5.log 2>&1
  echo "== $5: $1 grid $2 M=$3 tol=$4 (rocprofv3 --kernel-trace --stats, 3 transforms)" >> $O/r02_configs_kernel_stats.txt
  python3 tools/kstats.py $O/prof_$5 10 >> $O/r02_configs_kernel_stats.txt
done
cat $O/r02_sparse_crossover.txt $O/r02_configs.txt
