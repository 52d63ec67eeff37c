"""The RCCL branches of the tile exchange, executed on the ONE GPU of the box: a fresh child (started before anything here touches
the GPU API in it; nothing is exec'ed from an initialised process) creates a 1-rank 'nccl' group and runs tests/rccl_child.py.
What a 1-rank communicator cannot show is the multi-peer xGMI transfer itself; everything on this side of it runs here:
`init_process_group('nccl', device_id=...)` under `rccl_env_defaults()`, the counted all-reduce, grouped isend / irecv with DEVICE
tensors, `wait()` as a stream dependency (no host synchronisation between the producer kernels, the copy and the consumer), and
the product's `clip_test_distributed` in loopback mode -- bitwise `restore()`.  Reference being replaced: tools/test.py:82-86,
:124-146; tools/dist_test.sh:8-10."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _child_env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT', 'CIAOSR_DIST_BACKEND')}
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        env['CIAOSR_CHILD_PORT'] = str(s.getsockname()[1])
    env.update(OMP_NUM_THREADS='4', **kw)
    return env


def test_one_rank_rccl_group_handoff_and_loopback_tiles_are_bitwise():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'tests', 'rccl_child.py')], capture_output=True, text=True, timeout=900,
                         cwd=ROOT, env=_child_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    ok = [l for l in out.stdout.splitlines() if l.startswith('RCCL1_OK')]
    assert len(ok) == 1, out.stdout[-1500:]
    d = json.loads(ok[0][len('RCCL1_OK'):])
    assert d['probe']['ranks'] == 1 and d['probe']['backend'] == 'nccl' and d['probe']['p2p_channels'] == '1'
    assert d['probe']['ipc_mode_legacy'] == '0' and abs(d['probe']['tile_mb'] - 7.08) < 0.01
    assert d['checked'] == ['loopback/fp32/batch8/share1', 'loopback/f16/batch2/share1', 'loopback/f16/batch8/share0']


def test_bench_n1_line_counts_its_ranks_over_rccl():
    """`python bench.py` at N = 1 reports `rccl_ranks` as counted by a communicator (probe child), not as a constant."""
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--workload', 'c2', '--steps', '3', '--warmup', '1', '--no-cpu-baseline',
           '--no-extras', '--no-live-pmc']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=_child_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    d = json.loads([l for l in out.stdout.strip().splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['rccl_ranks'] == 1
    assert d['rccl_probe']['ok'] and d['rccl_probe']['warm']['tile_bitwise'] and d['rccl_probe']['backend'] == 'nccl'
