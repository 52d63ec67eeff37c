"""gpurun_out/prof_<tag>/staged_<route>/{stats,fetch,write} (tools/profile_staged.sh) -> the K1 / K4 evidence files for profiles/:
   <out>/c3tile_staged_kernel_stats.csv          rocprofv3 --stats rows of the staged kernels, one section per route (Name, Calls, AverageNs ...)
   <out>/c3tile_staged_pmc_hbm_traffic.json      per kernel: FETCH_SIZE x2 (gfx950) + WRITE_SIZE bytes per launch and per wavefront
   <out>/c3tile_staged_rooflines.txt             the table: kernel, route, average launch (rocprof), algorithmic bytes, GB/s, fraction of 8 TB/s,
                                                 counter bytes / algorithmic bytes
   python tools/staged_summary.py gpurun_out/prof_r6 profiles_staging_dir"""
import csv
import glob
import json
import os
import subprocess
import sys

src, out = sys.argv[1], sys.argv[2]
os.makedirs(out, exist_ok=True)
HERE = os.path.dirname(os.path.abspath(__file__))
ROUTES = ('as-written', 'staged', 'as-written-bf16', 'as-written-f16')
KEEP = ('local_attention', 'gather_rows', 'head_rows', 'gemm_f32_kernel', 'gemm_h16', 'cast_rows', 'decode_residual', 'patch_rows')
Q_TILE, CHUNK = 768 * 768, 30000
LAUNCHES = -(-Q_TILE // CHUNK)
# algorithmic HBM bytes per query (SURVEY 8(d), C = 64, x4)
ALGO = {'local_attention_kernel<4>': 22064.0, 'gather_rows_kernel': 21936.0, 'local_attention_h16_kernel': 11056.0,
        'head_rows_kernel': 4 * 2 * 256 * 4.0 * 2,          # hoisted K1: reads 2 table rows + writes 2 hidden rows of 256 fp32 per (query, sample)
        'head_rows_query_kernel': 4 * 2 * 256 * 4.0 * 2}    # ... one wave per query (round 6)

stats_rows, pmc, table = [], {}, []
for route in ROUTES:
    d = os.path.join(src, 'staged_' + route)
    f = glob.glob(os.path.join(d, 'stats', '**', '*kernel_stats.csv'), recursive=True)
    rows = list(csv.DictReader(open(f[0]))) if f else []
    for r in rows:
        if any(k in r['Name'] for k in KEEP):
            stats_rows.append(dict(route=route, **r))
    fe = glob.glob(os.path.join(d, 'fetch', '**', '*counter_collection.csv'), recursive=True)
    wr = glob.glob(os.path.join(d, 'write', '**', '*counter_collection.csv'), recursive=True)
    pm = {}
    if fe and wr:
        pm = json.loads(subprocess.run([sys.executable, os.path.join(HERE, 'pmc_summary.py'), fe[0], wr[0]], capture_output=True, text=True, check=True).stdout)
        for k, v in pm.items():
            if any(s in k for s in KEEP):
                pmc[f'{k} [{route}]'] = v
    for kname, algo in ALGO.items():
        hit = [r for r in rows if kname in r['Name']]
        if not hit:
            continue
        avg_ns, calls = float(hit[0]['AverageNs']), int(hit[0]['Calls'])
        cnt = [v for k, v in pm.items() if kname in k]
        waves_per_q = 4 if kname in ('gather_rows_kernel', 'head_rows_kernel') else 1      # head_rows_query_kernel: one wave per query
        # queries of an average launch: from the counter pass's grid sizes (one wavefront per query / per (query, sample) row) -- the as-written
        # route launches per eval_bsize chunk (20 per tile: 19 x 30 000 + 19 824), the C library's staged route per 65 536 queries (9 per tile)
        q_launch = cnt[0]['waves_per_launch'] / waves_per_q if cnt else Q_TILE / LAUNCHES
        per_launch = algo * q_launch
        gbs = per_launch / avg_ns
        cb = cnt[0]['hbm_bytes_per_wave'] * waves_per_q if cnt else None
        table.append((kname, route, calls, avg_ns / 1e3, per_launch, gbs, gbs / 8000.0, cb, cb / algo if cb else None))

with open(os.path.join(out, 'c3tile_staged_kernel_stats.csv'), 'w', newline='') as f:
    if stats_rows:
        w = csv.DictWriter(f, fieldnames=list(stats_rows[0].keys()))
        w.writeheader()
        w.writerows(stats_rows)
json.dump(pmc, open(os.path.join(out, 'c3tile_staged_pmc_hbm_traffic.json'), 'w'), indent=1)
with open(os.path.join(out, 'c3tile_staged_rooflines.txt'), 'w') as f:
    f.write('K1 / K4 of the staged routes on ONE C3 tile (192x192 LR -> 768x768, 589 824 queries, 20 eval_bsize chunks): rocprofv3 --kernel-trace --stats average\n'
            'launch duration against the ALGORITHMIC bytes of an average launch (SURVEY 8(d) bytes per query x the queries of an average launch, from the grid sizes), peak 8000 GB/s; counter bytes =\n'
            'FETCH_SIZE x 2 (gfx950 correction) + WRITE_SIZE from separate --pmc passes, per query (one wavefront per query in K4, per (query, sample) in K1).\n\n')
    f.write(f'{"kernel":32s} {"route":16s} {"calls":>6s} {"avg us":>9s} {"alg. MB/launch":>15s} {"GB/s":>8s} {"frac of HBM":>12s} {"counter B/query":>16s} {"counter/alg.":>13s}\n')
    for k, route, calls, us, bl, gbs, frac, cb, ratio in table:
        f.write(f'{k:32s} {route:16s} {calls:6d} {us:9.1f} {bl / 1e6:15.1f} {gbs:8.1f} {frac:12.3f} ' +
                (f'{cb:16.0f} {ratio:13.2f}' if cb else f'{"-":>16s} {"-":>13s}') + '\n')
print(open(os.path.join(out, 'c3tile_staged_rooflines.txt')).read())
