#!/bin/bash
# Where does a workgroup of spread_2d_w8_group_kernel spend its time? Builds a copy of the library with
# -DNUFFT_HIP_PHASE_LOG (thread 0 of every workgroup stores the shader clock at the phase boundaries) into a scratch
# package and prints mean cycles per phase at config 2 and at config 5's item size. Run through gpurun.
cd $GRAFT_REPO_ROOT
S=/tmp/phaselog; rm -rf $S; mkdir -p $S/obj $S/pkg
cp -r tensorflow-nufft_amd/tensorflow_nufft $S/pkg/
C=tensorflow-nufft_amd/csrc
FL="-O3 -std=c++17 -fPIC -Iinclude -I$C --offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $FL -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -c $C/nufft_kernels.hip -o $S/obj/k.o || exit 1
/opt/rocm/bin/hipcc $FL -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -c $C/nufft_dense3.hip -o $S/obj/d.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -DNUFFT_SOURCE_DIGEST=experiment -x c++ -c $C/nufft_build_info.cpp -o /tmp/nufft_build_info_exp.o || exit 1   # (the variant says what it is: nufft_hip_build_info)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nufft_build_info_exp.o -o $S/pkg/tensorflow_nufft/libnufft_hip.so $S/obj/k.o \
  $S/obj/d.o $C/_obj/nufft_wide.o $C/_obj/nufft_line.o $C/_obj/nufft_fft.o $C/_obj/nufft_plan.o $C/_obj/nufft_op.o \
  -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
python3 - $S/pkg <<'PY'
import sys, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
names = ['locate subproblem', 'zero planes + counters', 'load keys + LDS histogram', 'scan', 'permutation', 'accumulate (main loop)', 'write-out']
for label, N, M in (('config 2: 1024^2, M = 1e7 (2441 points per tile)', 1024, 10_000_000), ('config 5 item: 512^2, M = 1e6 (977 points per tile)', 512, 1_000_000)):
  g = torch.Generator(device='cuda').manual_seed(2)
  pts = (torch.rand((M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
  plan = tfft.Plan('type_1', [N, N], 'forward', tol=1e-6)
  for _ in range(3): plan.execute_with_points(pts, c)
  torch.cuda.synchronize()
  ntile = (2 * N // 32) ** 2
  buf = (ctypes.c_uint64 * (ntile * 8))()
  assert lib.nufft_hip_debug_phase_log(buf, ntile * 8) == 0
  t = np.frombuffer(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)
  d = np.diff(t, axis=1)
  ok = (d >= 0).all(axis=1) & (t[:, 0] > 0)
  d = d[ok]
  tot = (t[ok, 7] - t[ok, 0])
  print(f'{label}: {ok.sum()} workgroups, whole workgroup {tot.mean():.0f} ticks of s_memtime (first start to last end: {(t[ok, 7].max() - t[ok, 0].min())} ticks)')
  for n, col in zip(names, d.T):
    print(f'   {n:28s} {col.mean():9.0f} ticks  {100 * col.mean() / tot.mean():5.1f} %')
  plan.close()
# config 4: spread_dense3_kernel
names = ['locate subproblem', '(in-LDS sort: GROUP only)', 'zero plane + sum of |c| pass', 'accumulate (main loop)', 'write-out']
g = torch.Generator(device='cuda').manual_seed(4)
M = 100_000_000
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
plan = tfft.Plan('type_1', [256, 256, 256], 'forward', tol=1e-4)
for _ in range(2): plan.execute_with_points(pts, c)
torch.cuda.synchronize()
ntile = 65536
buf = (ctypes.c_uint64 * (ntile * 8))()
assert lib.nufft_hip_debug_phase_log3(buf, ntile * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(ntile, 8).astype(np.int64)[:, :6]
d = np.diff(t, axis=1)
ok = (d >= 0).all(axis=1) & (t[:, 0] > 0)
d = d[ok]; tot = t[ok, 5] - t[ok, 0]
print(f'config 4: 256^3, M = 1e8 (1526 points per tile), fused records: {ok.sum()} workgroups, whole workgroup {tot.mean():.0f} ticks')
for n, col in zip(names, d.T):
  print(f'   {n:32s} {col.mean():9.0f} ticks  {100 * col.mean() / tot.mean():5.1f} %')
PY
