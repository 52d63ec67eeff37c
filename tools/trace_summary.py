"""Summarise a rocprofv3 kernel trace CSV by (kernel, grid): count and mean duration in us."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 else ''
d = collections.defaultdict(list)
for r in rows:
    if pat in r['Kernel_Name']:
        key = (r['Kernel_Name'][:48], int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1), r['Grid_Size_Y'])
        d[key].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
for k, v in sorted(d.items()):
    print(f'{k[0]:50s} wg={k[1]:6d} y={k[2]:3s} n={len(v):5d} mean={sum(v) / len(v) / 1000:9.2f} us  min={min(v) / 1000:9.2f}')
