"""Developer tool: registers, scratch and spills of every kernel of the library (hipcc -Rpass-analysis=kernel-resource-usage).
   python tools/spill_report.py [--all]      (default: only kernels that spill or use scratch)"""
import os
import re
import subprocess
import sys

HERE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'ciaosr_amd', 'csrc')
FLAGS = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-fno-slp-vectorize', '-Rpass-analysis=kernel-resource-usage']


def report(src, extra=()):
    out = subprocess.run(['/opt/rocm/bin/hipcc', *FLAGS, *extra, '-c', src, '-o', '/dev/null'], cwd=HERE, capture_output=True, text=True).stderr
    cur, rows = None, []
    for line in out.splitlines():
        m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|VGPRs Spill|SGPRs Spill|LDS Size \[bytes/block\]):\s*(\S+)', line)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == 'Function Name':
            cur = {'name': subprocess.run(['c++filt', v], capture_output=True, text=True).stdout.strip()[:70]}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(' [')[0]] = int(v)
    return rows


show_all = '--all' in sys.argv
srcs = sorted(f for f in os.listdir(HERE) if f.endswith('.hip'))
for f in srcs:
    variants = [('', ())] if not f.endswith('_h16.hip') else [('bf16', ('-DCIAOSR_F16=0',)), ('f16', ('-DCIAOSR_F16=1',))]
    for tag, extra in variants:
        for r in report(f, extra):
            if show_all or r.get('VGPRs Spill', 0) or r.get('SGPRs Spill', 0) or r.get('ScratchSize', 0):
                print(f"{f:22s} {tag:4s} {r['name']:70s} VGPR {r.get('VGPRs', 0):3d} AGPR {r.get('AGPRs', 0):3d} scratch {r.get('ScratchSize', 0):4d} B "
                      f"spills {r.get('VGPRs Spill', 0):3d} V / {r.get('SGPRs Spill', 0):2d} S")
