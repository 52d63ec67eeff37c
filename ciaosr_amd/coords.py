"""Coordinate / cell construction for implicit-function super-resolution.

`make_coord` restates mmedit 0.11.0 `mmedit.datasets.pipelines.utils.make_coord`
(external to /root/reference; call sites: ciaosr_net.py:148, ciaosr.py:240,
generate_assistant.py:70).  The fp32 evaluation order is part of the contract
(SURVEY Appendix A.1): seq[i] = fp32(v0 + r) + fp32(2r) * fp32(i).
"""
import torch


def make_coord(shape, ranges=None, flatten=True):
    """Grid-centre coordinates in [-1, 1] for an image of `shape` (H, W).

    Returns [H*W, 2] (flatten) or [H, W, 2]; last dim is (y, x).
    """
    seqs = []
    for i, n in enumerate(shape):
        v0, v1 = (-1, 1) if ranges is None else ranges[i]
        r = (v1 - v0) / (2 * n)
        seqs.append(v0 + r + (2 * r) * torch.arange(n).float())
    grid = torch.stack(torch.meshgrid(*seqs, indexing='ij'), dim=-1)
    return grid.view(-1, grid.shape[-1]) if flatten else grid


def make_cell(target_hw, n_query=None):
    """Per-query cell size (2/Ht, 2/Wt) as the test pipeline / clip_test build it
    (ciaosr.py:241-243, generate_assistant.py:88-90)."""
    ht, wt = target_hw
    n = ht * wt if n_query is None else n_query
    cell = torch.ones(n, 2)
    cell[:, 0] *= 2 / ht
    cell[:, 1] *= 2 / wt
    return cell
