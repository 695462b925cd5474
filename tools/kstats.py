#!/usr/bin/env python3
"""Prints the per-kernel averages of a rocprofv3 --kernel-trace --stats output directory."""
import csv, glob, os, sys
d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_stats.csv'), recursive=True), key=os.path.getmtime)[-1]
tot = 0.0
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 12]:
  n = r['Name'].replace('void nufft_hip::(anonymous namespace)::', '')
  print(f"{float(r['AverageNs'])/1e3:9.1f} us x {int(r['Calls']):4d}  {r['Percentage']:>6}%  {n[:100]}")
