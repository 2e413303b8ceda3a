"""Case-based YAML hyper-parameters: the reference's global ``hparam`` singleton.

Mirrors the interface of the reference's hparam.py:47-70 -- a module-level ``hparam``
object with ``set_hparam_yaml(case, default_file, user_file)``, dot and item access,
and the derived ``case`` / ``logdir`` attributes -- so code written against
``from hparam import hparam as hp`` keeps working.  Semantics kept on purpose:

* both files may hold several YAML documents; they are flattened into one dict
  (hparam.py:7-14);
* a case's dict wins, defaults fill in recursively, lists are replaced whole
  (hparam.py:17-24);
* an unknown case silently means "plain defaults" (hparam.py:59);
* ``logdir = logdir_path + '/' + case`` (hparam.py:64-68); paths are CWD-relative,
  with a fallback to the repo's own hparams/ directory.

Difference: ``yaml.safe_load_all`` is used (the reference's bare ``yaml.load_all(stream)``
raises TypeError on PyYAML >= 6).
"""
from __future__ import annotations

import os

import yaml

_REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _resolve(path: str) -> str:
    if os.path.exists(path):
        return path
    alt = os.path.join(_REPO_ROOT, path)
    return alt if os.path.exists(alt) else path


def load_hparam(filename: str) -> dict:
    flat = {}
    with open(_resolve(filename), 'r') as stream:
        for doc in yaml.safe_load_all(stream):
            if doc:
                flat.update(doc)
    return flat


def merge_dict(user, default):
    """Fill ``user`` with whatever ``default`` has and ``user`` lacks, recursively."""
    if isinstance(user, dict) and isinstance(default, dict):
        for key, dval in default.items():
            user[key] = merge_dict(user[key], dval) if key in user else dval
    return user


class Dotdict(dict):
    """dict with attribute access; nested mappings are wrapped on construction."""

    def __init__(self, dct=None):
        super().__init__()
        for key, value in (dct or {}).items():
            self[key] = Dotdict(value) if hasattr(value, 'keys') else value

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError:
            raise AttributeError(key)

    def __setattr__(self, key, value):
        self[key] = value

    def __delattr__(self, key):
        del self[key]


class Hparam(Dotdict):

    def set_hparam_yaml(self, case, default_file='hparams/default.yaml', user_file='hparams/hparams.yaml'):
        default_hp = load_hparam(default_file)
        user_hp = load_hparam(user_file)
        merged = merge_dict(user_hp[case], default_hp) if case in user_hp else default_hp
        for stale in list(self.keys()):
            del self[stale]
        for key, value in Dotdict(merged).items():
            setattr(self, key, value)
        self._auto_setting(case)
        return self

    def _auto_setting(self, case):
        self.case = case
        # a case's log directory is always <logdir_path>/<case>
        self.logdir = '{}/{}'.format(self.logdir_path, case)


hparam = Hparam()
