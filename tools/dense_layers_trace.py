"""Developer tool: per-layer launch durations of the 16-bit dense kernel from a rocprofv3 kernel trace of a C3 step (layer = launch index mod 8).
   rocprofv3 --kernel-trace --output-format csv -d DIR -o t -- python3 bench.py --workload c3 --precision f16 --steps 1 --warmup 1 --no-...
   python tools/dense_layers_trace.py DIR/t_kernel_trace.csv"""
import collections
import csv
import sys

rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'dense_h16' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
by = collections.defaultdict(list)
gaps = []
for i, r in enumerate(rows):
    d = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
    by[(i % 8, int(r['Grid_Size_X']) // max(int(r['Workgroup_Size_X']), 1))].append(d)
    if i and i % 8:
        gaps.append(int(r['Start_Timestamp']) - int(rows[i - 1]['End_Timestamp']))
print(f'{len(rows)} launches; gap between consecutive dense launches inside a block: mean {sum(gaps) / max(len(gaps), 1) / 1e3:.2f} us, min {min(gaps) / 1e3:.2f}')
tot = 0
for (layer, wgs), v in sorted(by.items()):
    tot += sum(v)
    print(f'layer {layer} ({2 * (layer + 1):2d} stages / item)  workgroups {wgs:4d}  n={len(v):4d}  mean {sum(v) / len(v) / 1e3:8.2f} us  min {min(v) / 1e3:8.2f}')
print(f'total {tot / 1e6:.2f} ms')
