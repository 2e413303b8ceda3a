"""A small stand-in for TensorFlow's variable store / variable scopes.

The reference builds its weights with ``tf.get_variable`` inside nested
``tf.variable_scope``s (modules.py:131-165,179,210-248; models.py:24-35,114-115,128) and
restores them by *name* from a checkpoint, preferring the EMA shadow
``<name>/ExponentialMovingAverage`` (generate.py:55-66, models.py:72-76).  This module keeps
those semantics -- same names, same TF layouts ``[width, Cin, Cout]``, random (glorot-uniform)
initialisation when no checkpoint is found -- on torch tensors resident in HBM.
"""
from __future__ import annotations

import contextlib
import math
import zlib
from typing import Dict, Iterable, Optional, Sequence

import numpy as np
import torch

EMA_SUFFIX = '/ExponentialMovingAverage'


def is_trainable(name: str) -> bool:
    """tf.trainable_variables(): every variable the path creates is trainable except the moving statistics of
    tf.layers.batch_normalization (modules.py:266).  The reference keeps EMA shadows of the trainable ones only
    (models.py:72-75) and, with hp.train.use_ema, restores ONLY those (generate.py:57-63)."""
    return not (name.endswith('/moving_mean') or name.endswith('/moving_variance'))


class VariableStore:
    _uid_counter = 0

    def __init__(self, device: Optional[torch.device] = None, seed: int = 2):
        VariableStore._uid_counter += 1
        self.uid = VariableStore._uid_counter      # never reused (id() can be, after garbage collection)
        self.device = torch.device(device) if device is not None else None
        self.seed = seed
        self.vars: Dict[str, torch.Tensor] = {}
        self.version = 0            # bumped on every change; packed-weight caches key on it
        self.restored = set()                       # names set from a checkpoint / dict (what tf.train.Saver.restore covered)
        self.restored_without_shadow = set()        # ... of those, under use_ema, the ones whose shadow the checkpoint lacks
        self.left_at_init = set()                   # non-trainable variables a use_ema restore deliberately did not touch
        self._last_restore_ema = False              # the mode of the MOST RECENT load_dict (what not_restored() answers for)
        self._warned_left_at_init = False

    # -- tf.get_variable -------------------------------------------------------------------
    def get_variable(self, name: str, shape: Sequence[int], initializer: Optional[str] = None) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        if name in self.vars:
            v = self.vars[name]
            if tuple(v.shape) != shape:
                raise ValueError('variable %s exists with shape %s, requested %s' % (name, tuple(v.shape), shape))
            return v
        if self.device is None:
            raise RuntimeError('VariableStore has no device; create it with device=... before building a model')
        gen = torch.Generator().manual_seed((self.seed * 1000003 + zlib.crc32(name.encode())) % (2 ** 31))
        if initializer == 'zeros':
            v = torch.zeros(shape)
        elif initializer == 'ones':
            v = torch.ones(shape)
        else:   # TF's default for get_variable without an initializer: glorot_uniform
            rf = int(np.prod(shape[:-2])) if len(shape) > 2 else 1
            fan_in = rf * (shape[-2] if len(shape) > 1 else shape[0])
            fan_out = rf * shape[-1]
            limit = math.sqrt(6.0 / (fan_in + fan_out))
            v = (torch.rand(shape, generator=gen) * 2 - 1) * limit
        v = v.to(dtype=torch.float32, device=self.device).contiguous()
        self.vars[name] = v
        self.version += 1
        return v

    # -- checkpoints -------------------------------------------------------------------------
    def assign(self, name: str, value) -> None:
        t = torch.as_tensor(np.asarray(value), dtype=torch.float32)
        if name in self.vars and tuple(self.vars[name].shape) != tuple(t.shape):
            raise ValueError('shape mismatch for %s: %s vs %s' % (name, tuple(self.vars[name].shape), tuple(t.shape)))
        self.vars[name] = t.to(self.device).contiguous()
        self.version += 1

    def load_dict(self, weights: Dict[str, np.ndarray], use_ema: bool = False, strict: bool = False) -> int:
        """Restore by TF variable name.  With ``use_ema`` a shadow ``<name>/ExponentialMovingAverage``
        wins over the raw variable (generate.py:59-63) -- and, as there, ONLY trainable variables are restored: the Saver's
        var_list holds tf.trainable_variables('iaf_vocoder') and nothing else, so the moving statistics of a batch norm keep
        what global_variables_initializer gave them (zeros / ones, generate.py:56).  Returns the number of variables set."""
        loaded = 0
        self._last_restore_ema = bool(use_ema)
        if not use_ema:
            self.left_at_init.clear()               # (a plain restore covers the non-trainable variables too: judge it on its own)
        names = set(k[:-len(EMA_SUFFIX)] if k.endswith(EMA_SUFFIX) else k for k in weights)
        for name in sorted(names):
            if use_ema and not is_trainable(name.split(':')[0]):
                self.left_at_init.add(name.split(':')[0])
                if not self._warned_left_at_init:
                    self._warned_left_at_init = True
                    import warnings
                    warnings.warn("pwv: a use_ema restore sets trainable variables only (generate.py:57-63); non-trainable ones such as %s keep "
                                  "their initial values -- a 'bn' model then normalises with mean 0 / variance 1, exactly like the reference"
                                  % name.split(':')[0])
                continue
            key = name + EMA_SUFFIX if (use_ema and name + EMA_SUFFIX in weights) else name
            if key not in weights:
                if strict:
                    raise KeyError(name)
                continue
            self.assign(name.split(':')[0], weights[key])
            self.restored.add(name.split(':')[0])
            if use_ema and key == name:
                self.restored_without_shadow.add(name.split(':')[0])
            else:
                self.restored_without_shadow.discard(name.split(':')[0])
            loaded += 1
        return loaded

    def ema_missing(self, scope: str = 'iaf_vocoder'):
        """Model variables under `scope` that a use_ema restore took from the RAW variable because the checkpoint has no
        ``<name>/ExponentialMovingAverage`` for them.  The reference's Saver maps every trainable variable of 'iaf_vocoder'
        to its shadow name (generate.py:59-63) and fails on a missing key; callers that mirror it treat this list as an error."""
        return sorted(k for k in self.vars if k in self.restored_without_shadow and k.startswith(scope) and is_trainable(k))

    def load_npz(self, path: str, use_ema: bool = False) -> int:
        with np.load(path) as z:
            return self.load_dict({k: z[k] for k in z.files}, use_ema=use_ema)

    def load_checkpoint(self, path: str, use_ema: bool = False) -> int:
        """`path` is an .npz of TF-named arrays or a TensorFlow V2 checkpoint prefix (<path>.index +
        <path>.data-*), read without TensorFlow by tf_checkpoint.py."""
        import os
        if path.endswith('.npz') and os.path.exists(path):
            return self.load_npz(path, use_ema=use_ema)
        if path.endswith('.index'):
            path = path[:-len('.index')]
        if os.path.exists(path + '.index'):
            from .tf_checkpoint import read_tf_checkpoint
            keep = lambda n: not (n.endswith('/Adam') or n.endswith('/Adam_1'))
            return self.load_dict(read_tf_checkpoint(path, name_filter=keep), use_ema=use_ema)
        raise FileNotFoundError('no checkpoint at %s (.npz or TF V2 .index/.data)' % path)

    def not_restored(self):
        """Model variables that exist but were never set from a checkpoint (i.e. carry their random initialisation) although
        the MOST RECENT restore should have covered them.  After a use_ema restore the non-trainable ones are not listed (the
        reference's Saver does not look for them, generate.py:57-63); after a plain restore (Saver with var_list=None) they are."""
        return sorted(k for k in self.vars if k not in self.restored and not (self._last_restore_ema and not is_trainable(k)))

    def save_npz(self, path: str) -> None:
        np.savez(path, **{k: v.detach().cpu().numpy() for k, v in self.vars.items()})

    def numpy(self) -> Dict[str, np.ndarray]:
        return {k: v.detach().cpu().numpy() for k, v in self.vars.items()}

    def trainable_variables(self, scope: str = '') -> Iterable[str]:
        return [k for k in self.vars if k.startswith(scope) and is_trainable(k)]


_default_store: Optional[VariableStore] = None
# The scope stack is per THREAD, like TensorFlow's (a variable scope belongs to the thread that opened it): two threads that
# run forwards side by side must not see each other's `iaf_vocoder/iaf2/...` prefixes -- with a process-wide stack they read and
# created each other's variables (found by tests/test_safe_call.py::test_two_threads_on_two_streams_..., round 5).
import threading as _threading
_scopes = _threading.local()


def _stack() -> list:
    st = getattr(_scopes, 'stack', None)
    if st is None:
        st = _scopes.stack = []
    return st


def get_default_store() -> VariableStore:
    global _default_store
    if _default_store is None:
        dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else None
        _default_store = VariableStore(device=dev)
    return _default_store


def set_default_store(store: Optional[VariableStore]) -> None:
    global _default_store
    _default_store = store


def reset_default_store(device=None, seed: int = 2) -> VariableStore:
    """The analogue of starting a fresh ``tf.Graph()`` (generate.py:25)."""
    global _default_store
    if device is None and torch.cuda.is_available():
        device = torch.device('cuda', torch.cuda.current_device())
    _default_store = VariableStore(device=device, seed=seed)
    return _default_store


@contextlib.contextmanager
def variable_scope(name: str, absolute: bool = False):
    """tf.variable_scope: names created inside get the prefix ``<name>/``.  ``absolute`` replaces the
    enclosing scopes instead of nesting in them (re-entering a captured scope, as TF does when a
    VariableScope object is passed)."""
    if absolute:
        saved, _scopes.stack = _stack(), [name]
        try:
            yield
        finally:
            _scopes.stack = saved
        return
    st = _stack()
    st.append(name)
    try:
        yield
    finally:
        st.pop()


def current_scope() -> str:
    return '/'.join(s for s in _stack() if s)


def scoped(name: str) -> str:
    pre = current_scope()
    return pre + '/' + name if pre else name


def get_variable(name: str, shape: Sequence[int], initializer: Optional[str] = None,
                 store: Optional[VariableStore] = None) -> torch.Tensor:
    return (store or get_default_store()).get_variable(scoped(name), shape, initializer)
