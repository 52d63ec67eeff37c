"""Developer probe: cycle stamps (wave 0) of the LAST dense_h16_wide_kernel launch of a bench run in f16 mode (layer 7 of the last block).
   Needs `make -C ciaosr_amd/csrc probe`:
   CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_probe.so python tools/dense_wide_probe.py [bench.py arguments, default --workload c3]"""
import contextlib
import ctypes as C
import io
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ciaosr_amd import _lib  # noqa: E402

sys.argv = ['bench.py'] + (sys.argv[1:] or ['--workload', 'c3']) + ['--precision', 'f16', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-live-pmc', '--no-extras', '--no-rccl-probe']
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
lib = _lib.load()
buf = (C.c_ulonglong * (256 * 8))()
lib.ciaosr_debug_probe_dw_read.restype = C.c_int
assert lib.ciaosr_debug_probe_dw_read(buf, 256 * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(256, 8).astype(np.int64)
a = a[a[:, 4] > 0]
print(f'{len(a)} workgroups; ticks of s_memtime (100 MHz: 1 tick = 10 ns ~ 20-24 shader cycles); per workgroup of the last launch:')
names = ['compute loops (sum over stages)', 's_waitcnt vmcnt(0) at stage end', 's_barrier at stage end', 'epilogues', 'lifetime', 'stages', 'items', 'prologue']
for i, nm in enumerate(names):
    x = a[:, i]
    print(f'  {nm:36s} avg {x.mean():9.1f}  min {x.min():7d}  max {x.max():7d}')
st = a[:, 5].mean()
print(f'  per stage: compute {a[:, 0].mean() / st:.1f} ticks, vmcnt {a[:, 1].mean() / st:.1f}, barrier {a[:, 2].mean() / st:.1f}; MFMA issue floor per stage and SIMD: 4608 cycles = ~210 ticks at 2.2 GHz')
