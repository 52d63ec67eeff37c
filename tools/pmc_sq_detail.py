"""Per-kernel SQ stall / instruction-mix summary from the two --pmc passes of tools/pmc_sq.sh.
   python tools/pmc_sq_detail.py gpurun_out/pmc_bf16 [name-filter]"""
import collections, csv, glob, sys
root = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ''
acc = collections.defaultdict(lambda: collections.defaultdict(list))
grid = {}
for d in ('sq1', 'sq2', 'fetch'):
    for f in glob.glob(f'{root}/{d}/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('ciaosr::', '')
            if 'ciaosr' not in r['Kernel_Name'] or pat not in k:
                continue
            acc[k][r['Counter_Name']].append(float(r['Counter_Value']))
            grid[k] = (int(r['Grid_Size']) // int(r['Workgroup_Size']), int(r['Workgroup_Size']), r['VGPR_Count'], r['Accum_VGPR_Count'], r['LDS_Block_Size'], r['Scratch_Size'])
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
    m = {c: sum(x) / len(x) for c, x in v.items()}
    if 'SQ_WAVE_CYCLES' not in m:
        continue
    cyc = m['GRBM_GUI_ACTIVE'] / 8.0
    wg, wgs, vg, ag, lds, scr = grid[k]
    waves = wg * wgs / 64
    wc = m['SQ_WAVE_CYCLES']
    print(f'{k[:44]:44s} n={len(v["GRBM_GUI_ACTIVE"]):3d} wg={wg:6d} vgpr={vg}+{ag} lds={lds} scratch={scr} cycles={cyc:9.0f} ({cyc / 2.4e3:7.1f} us @2.4GHz)')
    print(f'    mfma_busy {m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (cyc * 1024):.3f}   wave-cycle split: wait_any {m.get("SQ_WAIT_ANY", 0) / wc:.2f} '
          f'wait_inst {m.get("SQ_WAIT_INST_ANY", 0) / wc:.2f} active {m.get("SQ_ACTIVE_INST_ANY", 0) / wc:.2f} (valu {m.get("SQ_ACTIVE_INST_VALU", 0) / wc:.2f} lds {m.get("SQ_ACTIVE_INST_LDS", 0) / wc:.2f})')
    if 'SQ_INSTS_VALU' in m:
        print(f'    per wave: valu {m["SQ_INSTS_VALU"] / waves:8.0f} lds {m["SQ_INSTS_LDS"] / waves:7.0f} vmem_rd {m["SQ_INSTS_VMEM_RD"] / waves:6.0f} salu {m["SQ_INSTS_SALU"] / waves:7.0f}   '
              f'lds conflict {m.get("SQ_LDS_BANK_CONFLICT", 0) / max(m.get("SQ_LDS_IDX_ACTIVE", 1), 1):.2f} of lds cycles; wave life {wc * 4 / waves:9.0f} cyc')
    if 'FETCH_SIZE' in m:
        print(f'    FETCH_SIZE x2 = {m["FETCH_SIZE"] * 2 / 1024:.1f} MB per launch (KB units, gfx950 x2 correction)')
