"""Minimal counterpart of `mmcv.Config.fromfile` for the reference's python config files
(tools/test.py:72): executes the file (its `from mmedited.models...` imports resolve to the shim
package in this repo) and exposes the module namespace with attribute access."""
import os
import runpy
import sys


class ConfigDict(dict):
    def __getattr__(self, k):
        try:
            v = self[k]
        except KeyError as e:
            raise AttributeError(k) from e
        return v

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        if repo not in sys.path:
            sys.path.insert(0, repo)
        ns = runpy.run_path(path)
        cfg = Config({k: _wrap(v) for k, v in ns.items() if not k.startswith('__') and not isinstance(v, type(os))})
        cfg['filename'] = path
        return cfg
