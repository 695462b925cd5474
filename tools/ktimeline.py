#!/usr/bin/env python3
"""Timeline of the last N kernel dispatches of a rocprofv3 --kernel-trace directory: start offset,
duration and the gap to the previous kernel's end (us):  python tools/ktimeline.py DIR [N]"""
import csv, glob, os, sys
d = sys.argv[1]; n = int(sys.argv[2]) if len(sys.argv) > 2 else 24
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))[-n:]
t0 = int(rows[0]['Start_Timestamp']); prev = None
for r in rows:
  s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
  gap = (s - prev) / 1e3 if prev else 0.0
  name = r['Kernel_Name'].replace('void nufft_hip::(anonymous namespace)::', '').replace('nufft_hip::(anonymous namespace)::', '')
  print(f"{(s - t0) / 1e3:9.1f}  dur {(e - s) / 1e3:7.1f}  gap {gap:6.1f}  {name[:70]}")
  prev = e
