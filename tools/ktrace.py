#!/usr/bin/env python3
"""Per-dispatch durations (us) of kernels whose name contains a substring, from a rocprofv3
--kernel-trace output directory:  python tools/ktrace.py DIR SUBSTR"""
import csv, glob, os, sys
d, sub = sys.argv[1], sys.argv[2]
f = sorted(glob.glob(os.path.join(d, '**', '*kernel_trace.csv'), recursive=True), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(f)):
  if sub in r['Kernel_Name']:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    print(f"{dur:8.1f} us  grid {r.get('Grid_Size_X','?')}x{r.get('Grid_Size_Y','?')} wg {r.get('Workgroup_Size_X','?')} lds {r.get('LDS_Block_Size','?')} vgpr {r.get('VGPR_Count','?')} sgpr {r.get('SGPR_Count','?')} scratch {r.get('Scratch_Size', r.get('Private_Segment_Size','?'))}")
