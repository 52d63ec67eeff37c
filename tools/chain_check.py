"""Developer check of the weights-stationary 16-bit head kernel (csrc/head_chain_h16.hip) against the 128-row kernels it replaces:
same model, same input, `head_route=HEAD_NO_CHAIN` vs default, every 16-bit weight form; odd sizes exercise the ragged block edges of
the grid traversal, a coordinate tensor of its own the index-order traversal.  Prints max |delta| of the RGB output (both routes round
the same products; fp32 summation orders differ) and the per-kernel times.   python tools/chain_check.py [192]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import rdn_ciaosr                              # noqa: E402
from ciaosr_amd import _lib, hip_ops                      # noqa: E402
from ciaosr_amd.coords import make_cell, make_coord       # noqa: E402
from ciaosr_amd.init_utils import seeded_init_, synthetic_pair   # noqa: E402


def main():
    size = int(sys.argv[1]) if len(sys.argv) > 1 else 192
    dev = torch.device('cuda', 0)
    model = rdn_ciaosr(dict(scale=4, tile=None))
    seeded_init_(model, seed=0, gain=1.0)
    model = model.to(dev)
    for (h, w, scale) in ((size, size, 4), (45, 51, 3.3), (48, 48, 4)):
        lq = synthetic_pair(h, w, 4)[0].to(dev)
        ht, wt = round(h * scale), round(w * scale)
        own = (make_coord((ht, wt)).unsqueeze(0).to(dev), make_cell((ht, wt)).unsqueeze(0).to(dev))     # no grid hint: index order
        hc, hl = hip_ops.make_coord_cell(ht, wt, dev)                                                     # hinted: a make_coord grid
        hc, hl = hc.unsqueeze(0), hl.unsqueeze(0)
        assert torch.equal(hc, own[0]) and torch.equal(hl, own[1])
        model.test_cfg['scale'] = scale
        for prec in ('f16', 'f16-pairs', 'bf16'):
            base = hip_ops.Options(prec)
            old = base.replace(head_route=_lib.HEAD_NO_CHAIN)
            ref32 = model.restore(lq, hc, hl, options=hip_ops.Options('fp32'))
            a = model.restore(lq, hc, hl, options=old)
            with hip_ops.profile():
                b = model.restore(lq, hc, hl, options=base)
                torch.cuda.synchronize()
            pr = hip_ops.profile.results()
            c = model.restore(lq, own[0], own[1], options=base)
            same = all(torch.equal(b, model.restore(lq, hc, hl, options=base)) for _ in range(4))     # run-to-run bitwise
            with hip_ops.profile():
                model.restore(lq, hc, hl, options=old)
                torch.cuda.synchronize()
            po = hip_ops.profile.results()
            dec = ' '.join('%s %.3f' % (k, pr[k]['total_ms']) for k in pr if k.startswith('head_decode'))
            tag = [k for k in pr if k.startswith('head_kv_chain')]
            told = [k for k in po if k.startswith('head_kv_fused')]
            print(f'{h}x{w} x{scale} {prec:10s} chain-old {float((a - b).abs().max()):.3e}  hinted-unhinted {float((b - c).abs().max()):.3e}  '
                  f'rerun {"bitwise" if same else "DIFFERS"}  chain-fp32 {float((b - ref32).abs().max()):.3e}  old-fp32 {float((a - ref32).abs().max()):.3e}  '
                  f'{tag[0] if tag else "NO CHAIN KERNEL"} {pr[tag[0]]["total_ms"] if tag else 0:.3f} ms '
                  f'(fallback {sum(pr[k]["total_ms"] for k in pr if k.startswith("head_kv_fused")):.3f})  old {sum(po[k]["total_ms"] for k in told):.3f} ms  '
                  f'decode {dec}  '
                  f'old {sum(po[k]["total_ms"] for k in po if k.startswith("head_decode")):.3f} ms', flush=True)


if __name__ == '__main__':
    main()
