#!/bin/bash
# A/B on the GPU box: sort kernels compiled for one or two 1024-thread workgroups per CU,
# two-call and one-call (fused) paths of config 2; rocprofv3 per-kernel averages.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02b; mkdir -p $O
for occ in 4 8; do
  touch tensorflow-nufft_amd/csrc/nufft_kernels.hip
  make -C tensorflow-nufft_amd/csrc EXTRA=-DNUFFT_SORT_MIN_WAVES=$occ > $O/build_$occ.log 2>&1
  for mode in two one; do
    flag=""; [ $mode = one ] && flag="--one-call"
    rm -rf $O/prof_${occ}_$mode
    timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_${occ}_$mode -o run --output-format csv -- python3 tools/profile_run.py --steps 12 $flag > $O/prof_${occ}_$mode.log 2>&1
    echo "== min waves $occ, $mode-call"; python3 tools/kstats.py $O/prof_${occ}_$mode 9
  done
done
touch tensorflow-nufft_amd/csrc/nufft_kernels.hip
make -C tensorflow-nufft_amd/csrc > /dev/null 2>&1
