#!/bin/bash
# A/B on the GPU box of the grouped 2-D spread kernel (config 2, one-call path).
# rocprofv3 kernel averages; every variant in the same box / run (devices differ by 5-10 %).
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r02d; mkdir -p $O
run() {  # tag, build flags, shapes
  touch tensorflow-nufft_amd/csrc/nufft_kernels.hip
  make -C tensorflow-nufft_amd/csrc EXTRA="$2" > $O/build_$1.log 2>&1 || { echo "build failed $1"; tail -5 $O/build_$1.log; return; }
  for shape in $3; do
    rm -rf $O/prof_$1_$shape
    NUFFT_HIP_W8_SHAPE=$shape timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$1_$shape -o run --output-format csv -- python3 tools/profile_run.py --steps 12 --one-call > $O/prof_$1_$shape.log 2>&1
    echo "== $1 ($2) shape $shape"; python3 tools/kstats.py $O/prof_$1_$shape 4 | grep -E "spread|scatter|hist"
  done
}
# (the r02 variants -- balanced shares, 16-point staging, readlane strengths -- were built with -D macros
# that have since been removed from the kernel; see DESIGN.md section 4 for their numbers)
run base "" "12x64 8x64 16x64"
run occ4 "-DNUFFT_SORT_MIN_WAVES=4" "12x64"
touch tensorflow-nufft_amd/csrc/nufft_kernels.hip
make -C tensorflow-nufft_amd/csrc > /dev/null 2>&1
