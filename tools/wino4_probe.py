"""Developer probe: phase stamps of the LAST dense_wino4_f32_kernel launch of a bench run (layer 7 of the last block: 64 steps).
   Needs `make -C ciaosr_amd/csrc probe`:
   CIAOSR_HIP_LIB=ciaosr_amd/csrc/libciaosr_hip_probe.so python tools/wino4_probe.py [bench.py arguments, default --workload c3tile]"""
import contextlib
import ctypes as C
import io
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from ciaosr_amd import _lib  # noqa: E402

sys.argv = ['bench.py'] + (sys.argv[1:] or ['--workload', 'c3tile']) + ['--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--no-live-pmc', '--no-extras']
with contextlib.redirect_stdout(io.StringIO()):
    bench.main()
lib = _lib.load()
buf = (C.c_ulonglong * (1024 * 8))()
lib.ciaosr_debug_probe_w4_read.restype = C.c_int
assert lib.ciaosr_debug_probe_w4_read(buf, 1024 * 8) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(1024, 8).astype(np.int64)
a = a[a[:, 0] > 0]
t0 = a[:, 0].min()
print(f'{len(a)} workgroups of image 0; ticks of s_memtime (100 MHz: 10 ns = 24 shader cycles)')
print(f'  start spread {a[:, 0].max() - t0}; end (stores acknowledged) spread {a[:, 7].max() - a[:, 7].min()}; launch span {a[:, 7].max() - t0} ticks')
for nm, x in (('prologue (first patches + weights landed)', a[:, 1] - a[:, 0]), ('main loop', a[:, 2] - a[:, 1]),
              ('  of which T phases to the barrier', a[:, 5]), ('  of which T phases through the barrier', a[:, 6]),
              ('output transform + exchange', a[:, 3] - a[:, 2]), ('bias / ReLU / stores issued', a[:, 4] - a[:, 3]), ('stores acknowledged', a[:, 7] - a[:, 4]),
              ('lifetime', a[:, 7] - a[:, 0])):
    print(f'  {nm:45s} avg {x.mean():9.1f}  min {x.min():7d}  max {x.max():7d}')
