"""The forward of an IAFVocoder captured once into a HIP graph and replayed.

The path is 16 dependent kernel launches per forward with the persistent stack launch (~115 on the per-layer path;
DESIGN.md section 4, "Launch structure": since round 5 one prologue launch or three plus ONE launch per flow); eager, every one costs a host enqueue and leaves a gap in front of the next
kernel.  Capturing the stream work (the launches of libpwv_hip.so on torch's current stream and, on the per-layer path,
on the two per-net side streams with their fork / join events) into one graph removes the host from the loop:
bit-identical results, 2 % faster at 160000 samples, 12 % at 16000 samples, 37 % for the one-flow configuration
(measured, tools/graph_bench.py).

No tracing and no compiler: the graph is exactly the launches `IAFVocoder.__call__` enqueues, with the buffers torch's
graph-private pool handed out during capture.  Shapes are fixed at capture (batch, length); the mel and the noise
are copied into static input tensors before each replay (or written there by the caller: `graphed.mel`), and the noise is
sampled by a node of the graph whose counter range lives in device memory (pwv_logistic_noise_stream_f32: a sampler with its
range passed by value would replay the same noise).
"""
from __future__ import annotations

from typing import Optional

import torch

from . import engine
from .hparam import hparam as hp
from .models import IAFVocoder


class GraphedVocoder(object):

    def __init__(self, model: IAFVocoder, device=None, warmup: int = 2):
        self.model = model
        store = model.store
        if store is None:
            from .variables import get_default_store
            store = get_default_store()
        self.store = store
        self.device = torch.device(device) if device is not None else store.device
        if self.device.type != 'cuda':
            raise engine._lib.PwvError('GraphedVocoder needs a GPU (cuda device); there is no CPU path')
        n, length = int(model.batch_size), int(model.length)
        self.mel = torch.zeros((n, model.t_mel, int(hp.signal.n_mels)), dtype=torch.float32, device=self.device)
        self.z = torch.zeros((n, length, 1), dtype=torch.float32, device=self.device)
        # the sampler is part of the graph: {seed, offset, ticket, skip} in device memory, advanced by the captured kernel itself
        # (pwv_logistic_noise_stream_f32), so a forward on sampled noise is ONE graph launch and nothing else
        self.noise_state = torch.zeros((4,), dtype=torch.int64, device=self.device)
        self._noise_mirror = (0, 0, 0)          # (seed, offset, skip) the device state holds
        self._last_drawn = 0                    # samples the replay that has not been verified yet took from the model's noise stream
        self._warmup = warmup
        self._capture()

    def _capture(self):
        # warm up on a side stream (plans packed, side streams created, allocator primed), as stream capture requires
        # (one stream for warm-up AND capture: what the warm-up forwards set up per stream -- the persistent launches' zeroed
        # workspace, engine._persist_ws -- is then found again by the captured forward instead of being allocated inside the graph)
        if getattr(self, '_stream', None) is None:
            self._stream = torch.cuda.Stream(device=self.device)
        side = self._stream
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            for _ in range(self._warmup):
                self.model(None, self.mel, is_training=False, z=self.z)
        torch.cuda.current_stream(self.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        # the captured launches carry THIS thread's sticky words (engine.current_words): verify() reads these, whoever replays
        self._words = engine.current_words(self.device)
        # thread_local: only this thread's calls are policed during capture (an RCCL watchdog thread of a multi-rank
        # job may touch the runtime meanwhile); everything captured here is enqueued from this thread
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode='thread_local'):
            engine.logistic_noise_stream_op(self.z, self.noise_state)
            self.out = self.model(None, self.mel, is_training=False, z=self.z)
        self._version = self.store.version
        self._mode = self._launch_mode()

    @staticmethod
    def _launch_mode():
        """what decides WHICH launches a forward enqueues, besides the weights: a graph captured under another value is stale"""
        return (engine.PERSIST, engine.persist_suspended(), engine.TWO_STREAMS, engine.FOLD_FIRST, engine.FUSE_FIRST, engine.FUSE_HEAD,
                engine.FUSE_TAIL, engine.HOIST_P, engine.PERSIST_MAX_LAYERS, engine.PERSIST_MIN_UNITS, engine.DEFAULT_PRECISION)

    def verify(self):
        """Replays only enqueue: wait for them and raise like IAFVocoder.verify().  After a PwvPersistError the engine has
        suspended the persistent launches (engine.suspend_persist); the graph is re-captured on the per-layer path here, so the
        caller's rerun replays launches that can complete -- and once the suspension has counted down (one tick per replay) the
        next call re-captures on the persistent path again (_launch_mode)."""
        try:
            engine.verify_enqueued(words=self._words)
        except engine._lib.PwvError as e:
            # the replay that failed drew its noise from the stream already: hand that range back, so that the caller's rerun
            # (z = None again) is a rerun ON THE SAME NOISE, like the eager path's (the device state is rewritten by the next call:
            # its mirror no longer matches)
            self.model.noise_offset -= self._last_drawn
            self._last_drawn = 0
            if isinstance(e, engine._lib.PwvPersistError):
                self._capture()
            raise
        self._last_drawn = 0

    def __call__(self, melspec: torch.Tensor, z: Optional[torch.Tensor] = None, seed: Optional[int] = None) -> torch.Tensor:
        """melspec [N, t_mel, n_mels] -- copied into the graph's input buffer, or that buffer itself (`graphed.mel`, filled by the caller:
        no copy); z [N, length, 1] or None (sample Logistic(0,1), models.py:32-33).
        Returns the graph's output buffer [N, length, 1]: valid until the next call (clone it to keep it).  Enqueue-only, like
        IAFVocoder.__call__(verify=False): call verify() before reading the result."""
        engine.note_forward()
        if self.store.version != self._version or self._mode != self._launch_mode():
            self._capture()      # weights changed (the captured launches point at stale packs) or the engine switched launch paths
        if tuple(melspec.shape) != tuple(self.mel.shape):
            raise ValueError('melspec must be %s (fixed at capture), got %s' % (tuple(self.mel.shape), tuple(melspec.shape)))
        if melspec is not self.mel:      # (a caller that writes its mel straight into the graph's input buffer `self.mel` passes that: no copy)
            self.mel.copy_(melspec, non_blocking=True)
        if z is None:
            # the captured sampler draws the model's next counter range (the stream IAFVocoder.sample_noise draws eagerly); the device
            # state is written from the host only when it is not what the previous replay left there (first call, another seed,
            # an eager draw in between, a call with explicit z before)
            want = (self.model._seed(seed), self.model.noise_offset, 0)
            self._set_noise_state(want)
            numel = self.z.numel()
            self.model.noise_offset += numel
            self._last_drawn = numel
            self._noise_mirror = (want[0], want[1] + numel, 0)
        else:
            if tuple(z.shape) != tuple(self.z.shape):
                raise ValueError('z must be %s, got %s' % (tuple(self.z.shape), tuple(z.shape)))
            self._last_drawn = 0
            self._set_noise_state((self._noise_mirror[0], self._noise_mirror[1], 1))      # skip: z is the caller's
            if z is not self.z:
                self.z.copy_(z, non_blocking=True)
        self.graph.replay()
        return self.out

    def _set_noise_state(self, want):
        if want != self._noise_mirror:
            # the state words are uint64 on the device (seeds up to 2**64 - 1, like the eager sampler's c_uint64): same bits as int64
            wrap = lambda v: (int(v) & ((1 << 64) - 1)) - (1 << 64) if (int(v) & (1 << 63)) else int(v) & ((1 << 64) - 1)  # noqa: E731
            self.noise_state.copy_(torch.tensor([wrap(want[0]), wrap(want[1]), 0, want[2]], dtype=torch.int64), non_blocking=False)
            self._noise_mirror = want
