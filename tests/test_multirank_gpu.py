"""The N > 1 PRODUCT path on the GPU box: fresh child ranks (torch.distributed.run) sharing the one GPU, exchange over gloo
(CIAOSR_DIST_BACKEND=gloo).  Covers BASELINE config C4's code -- `clip_test_distributed` with the HIP tile function, its
batched-encoder variant, `predict_query_sharded`, `bench.py --gpus N` started from a plain shell, and
`tools/test.py --launcher pytorch` -- which the CPU gloo tests (tests/test_tile_shard.py) only drive with an oracle tile
function.  Reference being replaced: tools/dist_test.sh:8-10, tools/test.py:124-146, restorers/ciaosr.py:233-254."""
import json
import os
import re
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def gpu():
    if not torch.cuda.is_available():
        pytest.skip('needs the MI355X')
    from ciaosr_amd import _lib
    _lib.load()


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_PORT')}
    env.update(dict(CIAOSR_DIST_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0', OMP_NUM_THREADS='4'), **kw)
    return env


@pytest.mark.parametrize('world', [2, 3])
def test_child_ranks_tile_and_query_sharding_are_bitwise_the_single_process_result(gpu, world):
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={world}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'multirank_child.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    ok = [l for l in out.stdout.splitlines() if l.startswith('MULTIRANK_OK')]
    assert len(ok) == 1 and ok[0].split()[1] == str(world), out.stdout[-1500:]
    for case in ('tiles/fp32/batch8', 'tiles/fp32/batch2', 'tiles/f16/batch8', 'tiles/f16/batch1', 'tiles/gather_to_all',
                 'queries/fp32', 'queries/f16'):
        assert case in ok[0], ok[0]


def test_c4_at_its_real_shape_eight_ranks_117_tiles(gpu):
    """BASELINE config C4's shape on the one GPU (exchange over gloo): 8 fresh ranks x the 117 tiles of the LR 1356x2040 image,
    f16 and fp32 -- rank 0's image bitwise == restore(); 117 = 14 full rounds of 8 + a ragged round of 5 that rank 0 sits out."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=8', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'tests', 'multirank_child.py')]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, cwd=ROOT,
                         env=_env(CIAOSR_CHILD_CASE='c4', OMP_NUM_THREADS='2'))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    ok = [l for l in out.stdout.splitlines() if l.startswith('MULTIRANK_OK')]
    assert len(ok) == 1 and ok[0].split()[1] == '8', out.stdout[-1500:]
    assert 'c4/f16' in ok[0] and 'c4/fp32' in ok[0] and 'tiles_per_rank 14,15,15,15,15,15,14,14' in ok[0], ok[0]


def test_bench_eight_ranks_c3_from_a_plain_shell(gpu):
    """`python bench.py --gpus 8 --workload c3 --precision f16 --steps 1` from a plain shell: the C4 command line, rehearsed on one
    GPU over gloo; the line carries the N > 1 diagnostics."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8', '--workload', 'c3', '--precision', 'f16', '--steps', '1',
           '--warmup', '1']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=2400, cwd=ROOT, env=_env(OMP_NUM_THREADS='2'))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['rccl_ranks'] == 8 and d['scaling'] == 'strong'
    assert d['tiles_per_rank'] == [14, 15, 15, 15, 15, 15, 14, 14] and len(d['rank_ms_per_step']) == 8
    assert d['p2p_channels'] == 1 and d['step_deadline_s'] > 0 and d['rank0_share']['value'] == 1.0
    assert 0 <= d['exposed_tail_ms'] < d['ms_per_step']
    assert d['config']['tiles'] == 117 and d['config']['parallelism'] == 'tile-shard x8'


def test_bench_starts_its_own_ranks_from_a_plain_shell(gpu):
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how the driver starts N = 1): bench.py spawns its ranks,
    asserts the sharded image bitwise against restore() before timing, and rank 0 prints the one JSON line."""
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--workload', 'c3s', '--steps', '2', '--warmup', '1']
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, cwd=ROOT, env=_env())
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{')]
    assert len(lines) == 1, out.stdout[-1500:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['rccl_ranks'] == 2 and d['scaling'] == 'strong' and d['steps'] == 2
    assert d['tiles_per_rank'] == [3, 3] and len(d['rank_ms_per_step']) == 2 and d['p2p_channels'] == 1
    assert 0 <= d['exposed_tail_ms'] < d['ms_per_step']
    assert abs(d['value'] - 1356 * 2040 / 1e6 / (d['ms_per_step'] * 1e-3)) < 1e-2 * d['value']
    assert d['config']['parallelism'] == 'tile-shard x2' and d['roofline']['kernel']


def test_tools_test_launcher_pytorch_from_a_plain_shell(gpu, tmp_path):
    """tools/dist_test.sh CONFIG CKPT 2 (= tools/test.py --launcher pytorch --gpus 2) on a tiled config prints the same Eval-PSNR /
    Eval-SSIM as the single-process run, and on a whole-image config (images sharded) too."""
    from ciaosr_amd.imageio import imwrite
    from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
    from ciaosr_amd import metrics
    import ciaosr_amd
    from ciaosr_amd.config import Config
    (tmp_path / 'lq').mkdir(); (tmp_path / 'gt').mkdir()
    for i, (h, w) in enumerate([(40, 56), (64, 48), (36, 36)]):
        lq, gt = synthetic_pair(h, w, 4, seed=200 + i)
        imwrite(metrics.tensor2img(lq), str(tmp_path / 'lq' / f'img{i}.png'))
        imwrite(metrics.tensor2img(gt), str(tmp_path / 'gt' / f'img{i}.png'))
    body = (
        "from mmedited.models.restorers.ciaosr import CiaoSR\n"
        "from mmedited.models.backbones.sr_backbones.ciaosr_net import LocalImplicitSREDSR\n"
        "mk = lambda i, o: dict(type='MLPRefiner', in_dim=i, out_dim=o, hidden_list=[256, 256, 256, 256])\n"
        "model = dict(type=CiaoSR, generator=dict(type=LocalImplicitSREDSR, encoder=dict(type='EDSR', in_channels=3,"
        " out_channels=3, mid_channels=64, num_blocks=4), imnet_q=mk(4, 3), imnet_k=mk(64, 64), imnet_v=mk(64, 64),"
        " feat_unfold=True, eval_bsize=30000), rgb_mean=(0.4488, 0.4371, 0.4040), rgb_std=(1., 1., 1.),"
        " pixel_loss=dict(type='L1Loss', loss_weight=1.0, reduction='mean'))\n"
        "test_cfg = dict(metrics=['PSNR', 'SSIM'], crop_border=4, scale=4, {tile}convert_to='y')\n"
        f"data = dict(test=dict(type='SRFolderDataset', lq_folder={str(tmp_path / 'lq')!r}, gt_folder={str(tmp_path / 'gt')!r},"
        " scale=4, filename_tmpl='{{}}'))\n"
        "dist_params = dict(backend='nccl')\n")
    for name, tile in (('tiled', 'tile=32, tile_overlap=8, '), ('whole', '')):
        cfg_path = tmp_path / f'cfg_{name}.py'
        cfg_path.write_text(body.format(tile=tile))
        cfg = Config.fromfile(str(cfg_path))
        model = ciaosr_amd.build_model(cfg.model, train_cfg=None, test_cfg=cfg.test_cfg)
        seeded_init_(model, seed=8, gain=1.25, head_gain=2.0)
        torch.save({'state_dict': model.state_dict()}, tmp_path / 'ck.pth')
        evals = []
        for cmd in ([sys.executable, os.path.join(ROOT, 'tools', 'test.py'), str(cfg_path), str(tmp_path / 'ck.pth')],
                    ['bash', os.path.join(ROOT, 'tools', 'dist_test.sh'), str(cfg_path), str(tmp_path / 'ck.pth'), '2']):
            out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())
            assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
            evals.append(dict(re.findall(r'Eval-(\w+): ([0-9.eE+-]+)', out.stdout)))
        assert set(evals[0]) == {'PSNR', 'SSIM'} and evals[0] == evals[1], (name, evals)
