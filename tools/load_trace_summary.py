"""Kernel counts of a `rocprofv3 --kernel-trace` CSV by name class: python tools/load_trace_summary.py DIR"""
import csv
import glob
import os
import sys
from collections import Counter

c = Counter()
for f in glob.glob(os.path.join(sys.argv[1], '**', '*kernel_trace.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        c[r['Kernel_Name'].split('(')[0][:70]] += 1
pack = sum(v for k, v in c.items() if 'pack' in k)
tensile = sum(v for k, v in c.items() if k.startswith('Cijk'))
print(f'total launches {sum(c.values())}; pack kernels {pack}; Tensile (Cijk_*) kernels {tensile}')
for k, v in c.most_common(25):
    print(f'  {v:6d}  {k}')
