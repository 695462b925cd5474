"""Times BASELINE configs 2-5 (GPU) with per-stage HIP-event timing."""
import os, sys, time, json
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), '..'))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tensorflow-nufft_amd'))
import numpy as np, torch
import tensorflow_nufft as tfft

def rnd_c(shape, g):
  return torch.complex(torch.rand(shape, generator=g, device='cuda') - .5, torch.rand(shape, generator=g, device='cuda') - .5)

def run(name, ttype, grid, M, tol, ntransf=1, per_item_points=False, steps=10, **kw):
  g = torch.Generator(device='cuda').manual_seed(3)
  rank = len(grid)
  npts = ntransf if per_item_points else 1
  pts = (torch.rand((npts, M, rank), generator=g, device='cuda') * 2 - 1) * np.pi
  T = 1 if per_item_points else ntransf
  lead = [T] if T > 1 else []
  plan = tfft.Plan(ttype, grid, 'forward', num_transforms=T, tol=tol, **kw)
  srcs = [rnd_c(lead + ([M] if ttype == 'type_1' else grid), g) for _ in range(npts)]
  def step():
    for i in range(npts):
      plan.set_points(pts[i]); plan.execute(srcs[i])
  step(); step()
  plan.set_timing(True); plan.get_timing()
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(steps): step()
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
  tm = plan.get_timing()
  i = plan.info()
  tot_pts = M * ntransf
  print(f'{name}: {dt*1e3:.3f} ms/step  {tot_pts/dt/1e6:.1f} Mpts/s  w={i.kernel_width} method={i.spread_method} tile={list(i.tile_dims)} batch={i.batch_size}')
  print('    stage us/call:', ' '.join(f'{k}={v[0]/max(v[1],1)*1e3:.0f}(x{v[1]//steps})' for k, v in tm.items() if v[1]))
  plan.close(); del pts, srcs; torch.cuda.empty_cache()

S = int(os.environ.get('BENCH_S', '0'))
which = sys.argv[1:] or ['2', '3', '4', '5', '5s']
if '2' in which: run('cfg2 2D t1 1024^2 M=1e7', 'type_1', [1024, 1024], 10_000_000, 1e-6)
if '3' in which: run('cfg3 2D t2 1024^2 M=1e7', 'type_2', [1024, 1024], 10_000_000, 1e-6)
if '4' in which: run('cfg4 3D t1 256^3 M=1e8 tol1e-4', 'type_1', [256, 256, 256], 100_000_000, 1e-4, steps=3, max_subproblem_size=S)
if '4s' in which: run('cfg4-small 3D t1 256^3 M=1e7 tol1e-4', 'type_1', [256, 256, 256], 10_000_000, 1e-4, steps=3, max_subproblem_size=S)
if '5' in which: run('cfg5 per-item pts: 32 x (2D t1 512^2 M=1e6)', 'type_1', [512, 512], 1_000_000, 1e-6, ntransf=32, per_item_points=True, steps=3)
if '5s' in which: run('cfg5 shared pts: 32 transforms (2D t1 512^2 M=1e6)', 'type_1', [512, 512], 1_000_000, 1e-6, ntransf=32, steps=3)
if '4t2' in which: run('cfg4-type2 3D t2 256^3 M=1e8 tol1e-4', 'type_2', [256, 256, 256], 100_000_000, 1e-4, steps=3)
if '2d' in which: run('2D t1 1024^2 M=1e7 f64 tol1e-9', 'type_1', [1024, 1024], 10_000_000, 1e-9, dtype=torch.complex128, steps=3)
if '2d6' in which: run('2D t1 1024^2 M=1e7 f64 tol1e-6', 'type_1', [1024, 1024], 10_000_000, 1e-6, dtype=torch.complex128, steps=3)
for t in ('1e-5', '1e-4', '1e-3', '1e-2'):
  if ('2t' + t) in which: run(f'2D t1 1024^2 M=1e7 f32 tol{t}', 'type_1', [1024, 1024], 10_000_000, float(t), steps=5)
  if ('3t' + t) in which: run(f'2D t2 1024^2 M=1e7 f32 tol{t}', 'type_2', [1024, 1024], 10_000_000, float(t), steps=5)
if '5op' in which:
  g = torch.Generator(device='cuda').manual_seed(5)
  B, M = 32, 1_000_000
  pts = (torch.rand((B, M, 2), generator=g, device='cuda') * 2 - 1) * np.pi
  c = rnd_c([B, M], g)
  opts = tfft.Options()   # BENCH_OP_GROUP / BENCH_OP_LANES: options.op_group / op_lanes (tools/sweep_group.sh)
  opts._internal = {'op_group': int(os.environ.get('BENCH_OP_GROUP', '0')), 'op_lanes': int(os.environ.get('BENCH_OP_LANES', '0'))}
  for _ in range(3): out = tfft.nufft(c, pts, grid_shape=[512, 512], transform_type='type_1', options=opts)
  torch.cuda.synchronize(); t0 = time.perf_counter()
  for _ in range(5): out = tfft.nufft(c, pts, grid_shape=[512, 512], transform_type='type_1', options=opts)
  torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
  print(f'cfg5 via tfft.nufft (op level, per-item points, 32 items): {dt*1e3:.3f} ms  {B*M/dt/1e6:.1f} Mpts/s')
if '1dbig' in which:
  run('1D t1 N=2^20 M=1e7 f32', 'type_1', [1 << 20], 10_000_000, 1e-6, steps=5)
  run('1D t2 N=2^20 M=1e7 f32', 'type_2', [1 << 20], 10_000_000, 1e-6, steps=5)
if '1' in which: run('cfg1 1D t1 N=4096 M=1e5 f64', 'type_1', [4096], 100_000, 1e-6, dtype=torch.complex128)
if '4t2s' in which: run('3D t2 256^3 M=1e7 tol1e-4 (sparse)', 'type_2', [256, 256, 256], 10_000_000, 1e-4, steps=3)
if '4t2d' in which: run('3D t2 128^3 M=3e7 tol1e-4 (dense)', 'type_2', [128, 128, 128], 30_000_000, 1e-4, steps=3)
if '3d6' in which: run('3D t1 256^3 M=3e7 tol1e-6 f32', 'type_1', [256, 256, 256], 30_000_000, 1e-6, steps=3)
if '3d6t2' in which: run('3D t2 256^3 M=3e7 tol1e-6 f32', 'type_2', [256, 256, 256], 30_000_000, 1e-6, steps=3)
if '3d5' in which: run('3D t1 256^3 M=3e7 tol1e-5 f32', 'type_1', [256, 256, 256], 30_000_000, 1e-5, steps=3)
if '3d5d' in which: run('3D t1 256^3 M=3e7 tol1e-5 f32, fp64 LDS planes', 'type_1', [256, 256, 256], 30_000_000, 1e-5, steps=3, lds_accumulate=1)
if '3dd' in which: run('3D t1 128^3 M=2e7 tol1e-4 f64', 'type_1', [128, 128, 128], 20_000_000, 1e-4, dtype=torch.complex128, steps=3)
