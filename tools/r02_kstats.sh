#!/bin/bash
# rocprofv3 per-kernel averages of configs 3, 4 and 4-as-type-2 (the kernel-stats part of tools/r02_measure.sh)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02m; mkdir -p $O; rm -f $O/r02_configs_kernel_stats.txt
for cfg in "type_2 1024,1024 1e7 1e-6 cfg3" "type_1 256,256,256 1e8 1e-4 cfg4" "type_2 256,256,256 1e8 1e-4 cfg4t2"; do
  set -- $cfg
  rm -rf $O/prof_$5
  timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$5 -o run --output-format csv -- python3 tools/profile_run.py --type $1 --grid $2 --M $3 --tol $4 --steps 3 --one-call > $O/prof_$5.log 2>&1
  echo "== $5: $1 grid $2 M=$3 tol=$4 (rocprofv3 --kernel-trace --stats, 3 transforms)" >> $O/r02_configs_kernel_stats.txt
  python3 tools/kstats.py $O/prof_$5 10 >> $O/r02_configs_kernel_stats.txt
done
cat $O/r02_configs_kernel_stats.txt | grep -v "at::native" | head -30
