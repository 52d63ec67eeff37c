"""mmcv-style checkpoint loading (`load_checkpoint`, tools/test.py:115-118): accepts a bare
state_dict or {'state_dict': ...}, strips a leading 'module.', honours `revise_keys`."""
import re

import torch


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, revise_keys=((r'^module\.', ''),)):
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    for pat, rep in revise_keys:
        sd = {re.sub(pat, rep, k): v for k, v in sd.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if strict and (missing or unexpected):
        raise RuntimeError(f'checkpoint mismatch: missing {missing}, unexpected {unexpected}')
    return ckpt
