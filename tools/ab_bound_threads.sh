#!/bin/bash
# bound3_kernel with 256 / 384 / 512 threads per workgroup (builds scratch copies of the library). Run through gpurun.
cd $GRAFT_REPO_ROOT
C=tensorflow-nufft_amd/csrc
FL="-O3 -std=c++17 -fPIC -Iinclude -I$C --offload-arch=gfx950 -munsafe-fp-atomics"
for nt in 256 384 512; do
  S=/tmp/bound_$nt; rm -rf $S; mkdir -p $S/pkg
  cp -r tensorflow-nufft_amd/tensorflow_nufft $S/pkg/
  /opt/rocm/bin/hipcc $FL -DNUFFT_EXPERIMENT_BUILD -DNUFFT_BOUND_THREADS=$nt -c $C/nufft_dense3.hip -o $S/d.o || exit 1
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -DNUFFT_EXPERIMENT_BUILD -DNUFFT_BOUND_THREADS=$nt -DNUFFT_SOURCE_DIGEST=experiment -x c++ -c $C/nufft_build_info.cpp -o /tmp/nufft_build_info_exp.o || exit 1   # (the variant says what it is: nufft_hip_build_info)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nufft_build_info_exp.o -o $S/pkg/tensorflow_nufft/libnufft_hip.so $C/_obj/nufft_kernels.o \
    $S/d.o $C/_obj/nufft_wide.o $C/_obj/nufft_line.o $C/_obj/nufft_fft.o $C/_obj/nufft_plan.o $C/_obj/nufft_op.o \
    -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
  python3 - $S/pkg $nt <<'PY'
import sys
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import tensorflow_nufft as tfft
g = torch.Generator(device='cuda').manual_seed(1)
for n, M in ((256, 30_000_000), (128, 800_000)):
  pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
  plan = tfft.Plan('type_1', [n, n, n], 'forward', tol=1e-6)
  for _ in range(3): plan.set_points(pts)
  plan.set_timing(True); plan.get_timing()
  for _ in range(5): plan.set_points(pts)
  tm = plan.get_timing()
  b = plan.sub_bounds()
  print(f'bound3_kernel with {sys.argv[2]} threads, {n}^3 modes, M = {M:.0e}: {tm["sort_cell"][0] / tm["sort_cell"][1] * 1e3:.0f} us (bound mean {np.abs(b[b != 0]).mean():.2f})')
  plan.close()
PY
done
