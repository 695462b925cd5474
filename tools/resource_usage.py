#!/usr/bin/env python3
"""Registers / scratch / occupancy of the kernels in nufft_kernels.hip, from
hipcc -Rpass-analysis=kernel-resource-usage (no GPU needed).
    python tools/resource_usage.py [--file nufft_fft.hip] [substring ...]"""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'tensorflow-nufft_amd', 'csrc')
fname = 'nufft_kernels.hip'
if len(sys.argv) > 2 and sys.argv[1] == '--file':
  fname = sys.argv[2]
  del sys.argv[1:3]
r = subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'), '-I' + src,
                    '--offload-arch=gfx950', '-munsafe-fp-atomics', '-Rpass-analysis=kernel-resource-usage',
                    '-c', os.path.join(src, fname), '-o', '/tmp/nufft_kernels_ru.o'],
                   capture_output=True, text=True)
blocks = re.split(r'remark: [^\n]*Function Name: ', r.stderr)
names = [b.split('\n')[0].strip() for b in blocks[1:]]
dem = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
print(f'{"VGPR":>5} {"AGPR":>5} {"SGPR":>5} {"scratch":>8} {"occ":>4} {"LDS":>7}  kernel')
for b, n in zip(blocks[1:], dem):
  if sys.argv[1:] and not any(k in n for k in sys.argv[1:]):
    continue
  def g(k):
    m = re.search(k + r': (\d+)', b)
    return int(m.group(1)) if m else -1
  print(f'{g("VGPRs"):5d} {g("AGPRs"):5d} {g("SGPRs"):5d} {g("ScratchSize .bytes/lane."):8d} '
        f'{g("Occupancy .waves/SIMD."):4d} {g("LDS Size .bytes/block."):7d}  {n[:140]}')
