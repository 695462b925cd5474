#!/bin/bash
# Where does a workgroup of spread_patch3_kernel (3-D float, w = 8 / 7) spend its time? Same scheme as
# tools/phase_log_experiment.sh: a copy of the library built with -DNUFFT_HIP_PHASE_LOG. Run through gpurun.
# usage: tools/phase_log_patch3.sh [M] [modes]
cd $GRAFT_REPO_ROOT
S=/tmp/phaselog3; rm -rf $S; mkdir -p $S/obj $S/pkg
cp -r tensorflow-nufft_amd/tensorflow_nufft $S/pkg/
C=tensorflow-nufft_amd/csrc
FL="-O3 -std=c++17 -fPIC -Iinclude -I$C --offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $FL -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -c $C/nufft_dense3.hip -o $S/obj/d.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -DNUFFT_SOURCE_DIGEST=experiment -x c++ -c $C/nufft_build_info.cpp -o /tmp/nufft_build_info_exp.o || exit 1   # (the variant says what it is: nufft_hip_build_info)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nufft_build_info_exp.o -o $S/pkg/tensorflow_nufft/libnufft_hip.so $C/_obj/nufft_kernels.o \
  $S/obj/d.o $C/_obj/nufft_wide.o $C/_obj/nufft_line.o $C/_obj/nufft_fft.o $C/_obj/nufft_plan.o $C/_obj/nufft_op.o \
  -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
python3 - $S/pkg ${1:-30000000} ${2:-256} <<'PY'
import sys, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
M, n = int(sys.argv[2]), int(sys.argv[3])
names = ['locate subproblem + bound', '-', 'zero plane (+ strength pass)', 'accumulate (main loop)', 'write-out']
g = torch.Generator(device='cuda').manual_seed(1)
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
for tol in (1e-6, 1e-5):
  plan = tfft.Plan('type_1', [n, n, n], 'forward', tol=tol)
  plan.set_points(pts)
  for _ in range(2): plan.execute(c)
  torch.cuda.synchronize()
  ntile = 65536
  buf = (ctypes.c_uint64 * (ntile * 8))()
  assert lib.nufft_hip_debug_phase_log3(buf, ntile * 8) == 0
  t = np.frombuffer(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)[:, :6]
  d = np.diff(t, axis=1)
  ok = (d >= 0).all(axis=1) & (t[:, 0] > 0)
  d = d[ok]; tot = t[ok, 5] - t[ok, 0]
  live = int((plan.sub_bounds() != 0).sum())
  print(f'tol {tol:g} (w = {plan.info().kernel_width}), {n}^3 modes, M = {M:.3g}: {ok.sum()} logged workgroups ({M / max(live, 1):.0f} points each), whole workgroup {tot.mean():.0f} ticks')
  for nm, col in zip(names, d.T):
    print(f'   {nm:32s} {col.mean():9.0f} ticks  {100 * col.mean() / tot.mean():5.1f} %')
  plan.close()
PY
