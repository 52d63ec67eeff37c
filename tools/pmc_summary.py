"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; units KiB per dispatch).
gfx950 correction (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request for wide
coalesced reads, so the read side is doubled; WRITE_SIZE is taken as is (uncalibrated).
   python tools/pmc_summary.py FETCH_counter_collection.csv WRITE_counter_collection.csv
`hbm_bytes_per_wave` = bytes / (Grid_Size / 64): the kernels that run one wavefront per unit (K4 local attention: a query; K1 gather rows:
a (query, sample) row) read their per-unit traffic off it whatever the sizes of the individual launches were."""
import collections
import csv
import json
import sys


def load(path, counter):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter:
            d[r['Kernel_Name'].split('(')[0]].append((float(r['Counter_Value']), int(r['Grid_Size'])))
    return d


fetch, write = load(sys.argv[1], 'FETCH_SIZE'), load(sys.argv[2], 'WRITE_SIZE')
out = {}
for k in sorted(fetch, key=lambda k: -sum(v for v, _ in fetch[k])):
    if 'ciaosr' not in k:
        continue
    n = len(fetch[k])
    f = sum(v for v, _ in fetch[k]) / n * 1024 * 2.0          # bytes per launch, gfx950 x2 correction
    wl = write.get(k, [(0.0, 64)])
    w = sum(v for v, _ in wl) / max(len(wl), 1) * 1024
    waves = sum(g for _, g in fetch[k]) / n / 64.0
    out[k] = dict(launches=n, fetch_bytes_per_launch=round(f), write_bytes_per_launch=round(w),
                  hbm_bytes_per_launch=round(f + w), waves_per_launch=round(waves, 1), hbm_bytes_per_wave=round((f + w) / max(waves, 1.0), 1))
print(json.dumps(out, indent=1))
