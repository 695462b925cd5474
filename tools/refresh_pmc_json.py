"""Rebuilds profiles/pmc_spread_traffic.json and profiles/pmc_traffic_configs.json (what bench.py quotes as `traffic`)
from the counter summaries of one round: python3 tools/refresh_pmc_json.py r06"""
import json, os, re, sys
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
P = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'profiles')

def read(name, kernel_re):
  path = os.path.join(P, f'{tag}_pmc_{name}.txt')
  lines = open(path).read().split('\n')
  for i, l in enumerate(lines):
    if l.startswith('## ') and re.search(kernel_re, l):
      m = re.search(r'FETCH_SIZE (\d+) KiB\s+WRITE_SIZE (\d+) KiB', lines[i + 1])
      kern = re.sub(r'\(Geom.*', '', l[3:]).strip()
      f, w = int(m.group(1)), int(m.group(2))
      return {'kernel': 'nufft_hip::' + kern, 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
              'traffic_bytes_corrected': (2 * f + w) * 1024, 'traffic_bytes_raw': (f + w) * 1024,
              'source': f'profiles/{tag}_pmc_{name}.txt'}
  raise SystemExit(f'{path}: no kernel matching {kernel_re}')

d = read('cfg2', 'spread_2d_w8_group')
json.dump({'kernel': 'nufft_hip::spread_2d_w8_group_kernel', 'config': '2D t1 1024^2 M=1e7 tol=1e-6 fp32',
           'FETCH_SIZE_KiB': d['FETCH_SIZE_KiB'], 'WRITE_SIZE_KiB': d['WRITE_SIZE_KiB'],
           'traffic_bytes_corrected': d['traffic_bytes_corrected'], 'traffic_bytes_raw': d['traffic_bytes_raw'],
           'points': 10000000, 'source': d['source']}, open(os.path.join(P, 'pmc_spread_traffic.json'), 'w'), indent=1)
out = {}
out['config4_3d_type1_256_M1e8_tol1e-4'] = read('cfg4', 'spread_dense3_kernel')
e = out['config4_3d_type1_256_M1e8_tol1e-4']
e['traffic_bytes_calibrated'] = e['traffic_bytes_raw'] + 800_000_000
e['note'] = 'calibrated = FETCH + WRITE + half of the 1.6 GB of streamed records: gathers are counted exactly (profiles/r04_fetch_calibration.txt)'
out['3d_type1_256_M3e7_tol1e-6'] = read('w8_3d', 'spread_patch3_kernel')
out['3d_type1_256_M3e7_tol1e-6']['traffic_bytes_calibrated'] = out['3d_type1_256_M3e7_tol1e-6']['traffic_bytes_raw'] + 240_000_000
out['config3_2d_type2_1024_M1e7'] = read('cfg3', 'interp_point_kernel')
out['config5_item'] = read('cfg5_item', 'spread_2d_w8_group')
out['config5_item']['note'] = 'one 512^2, M = 1e6 item through the one-call entry (the fused records are a wide streaming read: FETCH doubled); algorithmic 24.4 MB'
out['3d_type1_256_M1e7_tol1e-6_stacks'] = read('w8_3d_stacks', 'spread_stack3_kernel')
out['3d_type1_256_M1e7_tol1e-6_stacks']['note'] = 'stacks of tiles: WRITE_SIZE = 2.2 x the 1.07 GB fine grid (per subproblem: 3.9 x)'
out['3d_type2_256_M1e7_tol1e-6'] = read('3d_type2', 'interp_point_kernel')
out['nonpow2_3d_type1_240_M1e7_tol1e-6'] = read('mixfft_240', 'spread_stack3_kernel')
out['nonpow2_3d_type1_240_M1e7_tol1e-6']['fft_mixed_kernel_per_pass'] = read('mixfft_240', 'fft_mixed_kernel')
out['nonpow2_3d_type1_240_M1e7_tol1e-6']['note'] = 'traffic = the dominant (spread) kernel; fft_mixed_kernel_per_pass = average of the three passes of the 480^3 fine grid (x 3 = 3.2 GB: what the passes move algorithmically)'
out['c128_3d_type1_256_M1e7_tol1e-6'] = read('c128_3d_stacks', 'spread_wave3_stack_kernel')
out['c128_3d_type1_256_M1e7_tol1e-6']['note'] = 'complex128 over stacks: WRITE_SIZE = 2.3 x the 2.15 GB fine grid (per subproblem: 5.7 x)'
out['c128_3d_type2_256_M1e7_tol1e-6'] = read('c128_3d_type2_stacks', 'interp_point_kernel')
out['c128_3d_type2_256_M1e7_tol1e-6']['note'] = 'complex128 interpolation over pipelined stacks'
json.dump(out, open(os.path.join(P, 'pmc_traffic_configs.json'), 'w'), indent=1)
print('ok', list(out))
