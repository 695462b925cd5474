#!/bin/bash
# Builds a copy of the library whose fixed-point bound limit is switched off (every subproblem of an uncrowded tile
# stays on spread_patch3_kernel whatever its bound) and runs tools/fx_error_vs_crest.py on it and on the product.
# The experimental library is built HERE (container) before gpurun; on the GPU box only the python part runs.
set -e
cd "$(dirname "$0")/.."
if [ "$1" = build ]; then
  mkdir -p tools/_exp
  C=tensorflow-nufft_amd/csrc
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -Wall -Wno-unused-result --offload-arch=gfx950 -munsafe-fp-atomics \
      -DNUFFT_EXPERIMENT_BUILD -DNUFFT_FX_BOUND_LIMIT=1e9 -x hip -c $C/nufft_plan.cpp -o tools/_exp/nufft_plan_nolimit.o
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -DNUFFT_EXPERIMENT_BUILD -DNUFFT_FX_BOUND_LIMIT=1e9 -DNUFFT_SOURCE_DIGEST=experiment -x c++ -c $C/nufft_build_info.cpp -o /tmp/nufft_build_info_exp.o || exit 1   # (the variant says what it is: nufft_hip_build_info)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nufft_build_info_exp.o -o tools/_exp/libnufft_hip_nolimit.so $C/_obj/nufft_kernels.o $C/_obj/nufft_dense3.o \
      $C/_obj/nufft_wide.o $C/_obj/nufft_line.o $C/_obj/nufft_fft.o tools/_exp/nufft_plan_nolimit.o $C/_obj/nufft_op.o \
      -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
  exit 0
fi
mkdir -p gpurun_out/fx
python tools/fx_error_vs_crest.py --lib tools/_exp/libnufft_hip_nolimit.so 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fx/nolimit.txt
python tools/fx_error_vs_crest.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/fx/product.txt
