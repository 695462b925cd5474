"""Turns the rocprofv3 outputs under gpurun_out/ into the committed summaries in
profiles/ (kernel stats + PMC traffic of the spread kernel)."""
import collections, csv, glob, json, re, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r01'
stats = sorted(glob.glob('gpurun_out/prof_bench2/**/*kernel_stats.csv', recursive=True), key=lambda f: __import__('os').path.getmtime(f))[-1]
rows = list(csv.DictReader(open(stats)))
lines = [f'# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras   (MI355X, {tag})',
         '# kernel | calls | avg_us | total_ms | pct']
for r in rows[:14]:
  lines.append(f"{r['Name'][:90]} | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['TotalDurationNs'])/1e6:.3f} | {r['Percentage']}")
open(f'profiles/{tag}_bench_kernel_stats.txt', 'w').write('\n'.join(lines) + '\n')
print('\n'.join(lines[:9]))

def counters(d):
  f = sorted(glob.glob(f'gpurun_out/{d}/**/*counter_collection.csv', recursive=True), key=lambda f: __import__('os').path.getmtime(f))[-1]
  agg = collections.defaultdict(lambda: collections.defaultdict(list))
  for r in csv.DictReader(open(f)):
    name = re.split(r'[(<]', r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', ''))[0]
    agg[name][r['Counter_Name']].append(float(r['Counter_Value']))
  return agg
fe, wr = counters('pmc2_fetch'), counters('pmc2_write')
out = {}
txt = ['# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 tools/profile_run.py --steps 3 --one-call',
       '# config 2 (2D t1 1024^2, M=1e7). KiB per dispatch. Corrected traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 per',
       '# MI355X_MICROARCH.md (FETCH_SIZE counts 128-B requests as 64 B on gfx950; the hist kernel below, which reads',
       '# exactly 80 MB of points, calibrates it: FETCH_SIZE = 40 MB).']
for k in fe:
  if 'nufft_hip' in k or 'fft_rtc' in k or 'transpose' in k:
    f = sum(fe[k]['FETCH_SIZE']) / len(fe[k]['FETCH_SIZE'])
    w = sum(wr[k]['WRITE_SIZE']) / len(wr[k]['WRITE_SIZE']) if k in wr else 0.0
    txt.append(f'{k[:48]:48s} FETCH_SIZE={f:10.0f} KiB  WRITE_SIZE={w:10.0f} KiB  corrected_bytes={(2*f+w)*1024:.4g}')
    if 'spread' in k:
      out = {'kernel': k, 'config': '2D t1 1024^2 M=1e7 tol=1e-6 fp32', 'FETCH_SIZE_KiB': f, 'WRITE_SIZE_KiB': w,
             'traffic_bytes_corrected': (2 * f + w) * 1024, 'traffic_bytes_raw': (f + w) * 1024, 'points': 10000000,
             'source': f'profiles/{tag}_pmc_traffic.txt'}
open(f'profiles/{tag}_pmc_traffic.txt', 'w').write('\n'.join(txt) + '\n')
json.dump(out, open('profiles/pmc_spread_traffic.json', 'w'), indent=1)
print('\n'.join(txt[4:]))
