"""Per-kernel MFMA-busy fraction from a rocprofv3 --pmc pass with GRBM_GUI_ACTIVE, SQ_BUSY_CYCLES,
SQ_VALU_MFMA_BUSY_CYCLES, SQ_WAVE_CYCLES (own run, --kernel-trace only).
Units on gfx950 (checked against head_kv_fused, whose MFMA count is known exactly: 40 108 032 v_mfma_f32_32x32x2_f32
x 64 cycles = the counter value to the cycle): SQ_VALU_MFMA_BUSY_CYCLES = sum over all SIMDs of MFMA-pipe busy cycles;
GRBM_GUI_ACTIVE = kernel cycles summed over the 8 XCDs.  mfma_busy = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs)."""
import collections
import csv
import json
import sys

d = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    d[r['Kernel_Name'].split('(')[0]][r['Counter_Name']].append(float(r['Counter_Value']))
out = {}
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1].get('GRBM_GUI_ACTIVE', [0]))):
    if 'ciaosr' not in k or 'GRBM_GUI_ACTIVE' not in v:
        continue
    mean = {c: sum(x) / len(x) for c, x in v.items()}
    cyc = mean['GRBM_GUI_ACTIVE'] / 8.0
    out[k] = dict(launches=len(v['GRBM_GUI_ACTIVE']), kernel_cycles=round(cyc),
                  mfma_busy_cycles_all_simds=round(mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)),
                  mfma_busy=round(mean.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(cyc * 1024.0, 1.0), 4),
                  wave_quad_cycles=round(mean.get('SQ_WAVE_CYCLES', 0.0)))
print(json.dumps(out, indent=1))
