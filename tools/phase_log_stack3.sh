#!/bin/bash
# Where does a workgroup of spread_stack3_kernel (3-D float, w = 8) spend its time? A copy of the library built with
# -DNUFFT_HIP_PHASE_LOG (EXTRA flags after the sizes). Run through gpurun.
# usage: tools/phase_log_stack3.sh [M] [modes] [len] [cap]
cd $GRAFT_REPO_ROOT
S=/tmp/phaselog3s; rm -rf $S; mkdir -p $S/obj $S/pkg
cp -r tensorflow-nufft_amd/tensorflow_nufft $S/pkg/
C=tensorflow-nufft_amd/csrc
FL="-O3 -std=c++17 -fPIC -Iinclude -I$C --offload-arch=gfx950 -munsafe-fp-atomics"
/opt/rocm/bin/hipcc $FL -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -c $C/nufft_dense3.hip -o $S/obj/d.o || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -Iinclude -I$C -DNUFFT_EXPERIMENT_BUILD -DNUFFT_HIP_PHASE_LOG -DNUFFT_SOURCE_DIGEST=experiment -x c++ -c $C/nufft_build_info.cpp -o /tmp/nufft_build_info_exp.o || exit 1   # (the variant says what it is: nufft_hip_build_info)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC /tmp/nufft_build_info_exp.o -o $S/pkg/tensorflow_nufft/libnufft_hip.so $C/_obj/nufft_kernels.o \
  $S/obj/d.o $C/_obj/nufft_wide.o $C/_obj/nufft_line.o $C/_obj/nufft_fft.o $C/_obj/nufft_plan.o $C/_obj/nufft_op.o \
  -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib || exit 1
python3 - $S/pkg ${1:-30000000} ${2:-256} ${3:-0} ${4:-0} <<'PY'
import sys, ctypes
sys.path.insert(0, sys.argv[1])
import numpy as np, torch
import tensorflow_nufft as tfft
from tensorflow_nufft import _lib
lib = ctypes.CDLL(_lib.LIB_PATH)
M, n, ln, cp = int(float(sys.argv[2])), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
names = ['setup (lookup, step, zero)', 'accumulate (wave 0)', 'wait for the other waves', 'write-out + move', 'barrier behind it']
g = torch.Generator(device='cuda').manual_seed(1)
pts = (torch.rand((M, 3), generator=g, device='cuda') * 2 - 1) * np.pi
c = torch.complex(torch.rand(M, generator=g, device='cuda') - .5, torch.rand(M, generator=g, device='cuda') - .5)
plan = tfft.Plan('type_1', [n, n, n], 'forward', tol=1e-6, tuning=_lib.TUNE['STACK_ON'])
if ln or cp: plan.stack_params(ln, cp)
plan.set_points(pts)
for _ in range(2): plan.execute(c)
torch.cuda.synchronize()
nst = plan.stacks().shape[0]
nlog = min(nst, 65536)
buf = (ctypes.c_uint64 * (nlog * 8))()
assert lib.nufft_hip_debug_phase_log3(buf, nlog * 8) == 0
t = np.frombuffer(buf, dtype=np.uint64).reshape(nlog, 8).astype(np.int64)
ok = (t[:, 0] > 0) & (t[:, 6] > t[:, 0])
t = t[ok]
tot = (t[:, 6] - t[:, 0]).mean()
print(f'{n}^3 modes, M = {M:.3g}, {nst} stacks, {ok.sum()} logged, tiles per stack {t[:, 7].mean():.1f}: whole workgroup {tot:.0f} ticks, per tile {tot / t[:, 7].mean():.0f}')
for k, nm in enumerate(names):
  v = t[:, 1 + k].mean()
  print(f'   {nm:32s} {v:9.0f} ticks  {100 * v / tot:5.1f} %   per tile {v / t[:, 7].mean():7.0f}')
plan.close()
PY
