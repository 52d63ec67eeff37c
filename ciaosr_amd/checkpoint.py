"""mmcv-style checkpoint loading (`load_checkpoint`, tools/test.py:115-118): accepts a bare
state_dict or {'state_dict': ...}, strips a leading 'module.', honours `revise_keys`."""
import re

import torch


def load_checkpoint(model, filename, map_location='cpu', strict=False, logger=None, revise_keys=((r'^module\.', ''),)):
    """Like mmcv.runner.load_checkpoint: every missing / unexpected key is reported (through `logger.warning`, or a
    `warnings.warn` when no logger is given), `strict=True` raises on any mismatch, and a checkpoint of which NOT ONE key
    matches the model (wrong prefix, wrong file) always raises -- evaluating a random-init model silently is never right."""
    import warnings
    ckpt = torch.load(filename, map_location=map_location, weights_only=False)
    sd = ckpt['state_dict'] if isinstance(ckpt, dict) and 'state_dict' in ckpt else ckpt
    for pat, rep in revise_keys:
        sd = {re.sub(pat, rep, k): v for k, v in sd.items()}
    own = set(model.state_dict().keys())
    if own and not (own & set(sd.keys())):
        raise RuntimeError(f'{filename}: none of its {len(sd)} keys matches the model (first key {next(iter(sd), None)!r}, the model '
                           f'expects e.g. {sorted(own)[0]!r}): wrong prefix or wrong checkpoint')
    missing, unexpected = model.load_state_dict(sd, strict=False)
    if missing or unexpected:
        msg = (f'{filename}: {len(missing)} missing key(s) {list(missing)[:8]}{" ..." if len(missing) > 8 else ""}; '
               f'{len(unexpected)} unexpected key(s) {list(unexpected)[:8]}{" ..." if len(unexpected) > 8 else ""}')
        if strict:
            raise RuntimeError('checkpoint mismatch: ' + msg)
        (logger.warning if logger is not None else warnings.warn)(msg)
    return ckpt
