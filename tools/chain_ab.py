"""Developer A/B timing of head_kv_chain variants on ONE box: every library given on the command line is loaded in a fresh child process
(CIAOSR_HIP_LIB; or `default`, or `ENV:NAME=VALUE` = the default library under one more environment variable), which times one kernel (--tag prefix, default the kv kernel) of one 192x192 tile (HIP-event profile, mean of N launches); the list is walked ROUNDS times so
that clock drift shows up as spread between rounds instead of as a difference between variants.
   python tools/chain_ab.py [--mode f16] [--rounds 3] libA.so libB.so ..."""
import argparse
import os
import subprocess
import sys

CHILD = r'''
import os, sys, torch
sys.path.insert(0, os.getcwd())
from bench import rdn_ciaosr
from ciaosr_amd import hip_ops
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair
mode = sys.argv[1]; n = int(sys.argv[2])
dev = torch.device('cuda')
model = rdn_ciaosr(dict(scale=4, tile=192, tile_overlap=32)); seeded_init_(model, 0); model = model.to(dev)
lq = synthetic_pair(192, 192, 4)[0].to(dev)
opt = hip_ops.Options(mode)
for _ in range(3): model.restore(lq, options=opt)
torch.cuda.synchronize()
with hip_ops.profile():
    for _ in range(n): model.restore(lq, options=opt)
    torch.cuda.synchronize()
pr = hip_ops.profile.results()
k = [x for x in pr if x.startswith(os.environ.get('AB_TAG', 'head_kv_chain'))]
tot = sum(v['total_ms'] for v in pr.values())
print('RESULT', k[0] if k else 'none', pr[k[0]]['avg_ms'] if k else 0.0, tot / n)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--mode', default='f16')
    ap.add_argument('--rounds', type=int, default=3)
    ap.add_argument('--n', type=int, default=20)
    ap.add_argument('--tag', default='head_kv_chain', help='prefix of the profile tag to report')
    ap.add_argument('libs', nargs='+')
    a = ap.parse_args()
    os.environ['AB_TAG'] = a.tag
    res = {lib: [] for lib in a.libs}
    for r in range(a.rounds):
        for lib in a.libs:
            if lib.startswith('ENV:'):            # default library, one more environment variable (developer switches read by the library)
                key, _, val = lib[4:].partition('=')
                env = dict(os.environ, **{key: val})
            elif lib == 'default':
                env = dict(os.environ)
            else:
                env = dict(os.environ, CIAOSR_HIP_LIB=os.path.abspath(lib))
            out = subprocess.run([sys.executable, '-c', CHILD, a.mode, str(a.n)], env=env, capture_output=True, text=True)
            line = [ln for ln in out.stdout.splitlines() if ln.startswith('RESULT')]
            if not line:
                print(lib, 'FAILED', out.stderr[-400:])
                continue
            _, tag, kv, tile = line[0].split()
            res[lib].append((float(kv), float(tile)))
            print(f'round {r} {os.path.basename(lib):40s} {tag:28s} kv {float(kv):.4f} ms   tile kernels {float(tile):.3f} ms', flush=True)
    for lib, v in res.items():
        if v:
            print(f'{os.path.basename(lib):40s} kv mean {sum(x[0] for x in v) / len(v):.4f} min {min(x[0] for x in v):.4f}   tile {sum(x[1] for x in v) / len(v):.3f}')


if __name__ == '__main__':
    main()
