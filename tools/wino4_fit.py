"""Developer tool: per-launch durations of the F(4x4) dense-layer kernel from a rocprofv3 kernel trace, fitted as a + b * groups
(one 192x192 tile = 72 workgroups = one per CU: the fit reads workgroup time directly).
    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/w4 -o w4 -- python3 $R/bench.py --workload c3tile --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc
    python tools/wino4_fit.py /tmp/w4"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'dense_wino4' in r['Kernel_Name']:
            rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']) - int(r['Start_Timestamp']), r['Kernel_Name'][:40]))
rows.sort()
d = np.array([r[1] for r in rows], float)
n = len(d) // 128 * 128
d = d[-n:].reshape(-1, 16, 8)              # [pass][block][layer]
per_layer = np.median(d.reshape(-1, 8), axis=0)
G = np.arange(1, 9)
b, a = np.polyfit(G, per_layer, 1)
print(rows[-1][2], 'launches', len(rows))
print('median ns by layer l = 0..7:', ' '.join(f'{v:.0f}' for v in per_layer))
print(f'fit: {a:.0f} ns fixed + {b:.0f} ns per 64-channel group = {b / 8:.0f} ns per 8-channel step = {b / 8 * 2.4:.0f} cycles at 2.4 GHz')
