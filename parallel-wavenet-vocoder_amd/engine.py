"""Host-side orchestration of the HIP kernels for one or two WaveNets evaluated side by side.

This is plumbing: it owns no arithmetic.  It packs TF-layout weights once per variable-store
version (pwv_pack_*), allocates the ping-pong activation buffers in HBM through torch's caching
allocator, and enqueues the per-layer launches of libpwv_hip.so on torch's current HIP stream.

Data layout in HBM (float32):
  residual stream   2 x (N*T rows x 64) per net in the "tile32" layout of include/pwv_hip.h (blocks of 32 rows
                    stored [16 channel quads][32 rows][4]: every wave-level load/store is 1 KB contiguous);
                    ping-pong: a layer reads x[t], x[t-d] and writes out[t].  fp16 blocks with precision 'f16'.
  projection P      [N*t_mel, 128*L] per net: relu(mel@dense) @ [gc_filter‖gc_gate] + biases for
                    all L layers at FRAME rate (hoisted: modules.py:216-228 are the same dot
                    products for all 80 samples of a frame), columns in the kernel's order
  net output        [N, T, Q]
"""
from __future__ import annotations

import ctypes
import os
import threading
import weakref
from ctypes import c_void_p
from typing import Dict, List, Optional, Sequence, Tuple

import torch

from . import _lib
from ._lib import StackArgs, check

# 'f32'  : v_mfma_f32_32x32x2_f32, exact fp32 fma chains (157 TFLOP/s class)
# 'f16x3': 3-term split-fp16 MFMA with fp32 accumulation (w*x ~= wh*xh + wh*xl + wl*xh, ~22-bit
#          mantissa products): same measured accuracy vs fp64 as 'f32' (~3e-6 on the full model,
#          bar 2e-5), 5.3x fewer matrix-pipe cycles.  Default; PWV_PRECISION overrides.
# 'f16'  : BUILD EXTENSION (BASELINE.json config 5; the reference is fp32 only): fp16 residual stream in HBM,
#          one fp16 MFMA product, fp32 accumulate.  ~1e-3 of the fp32 result -- never the default, never the
#          headline benchmark; tolerance stated in tests/test_gpu_f16.py.
PRECISIONS = {'f32': _lib.PREC_F32, 'f16x3': _lib.PREC_F16X3, 'f16': _lib.PREC_F16}
DEFAULT_PRECISION = os.environ.get('PWV_PRECISION', 'f16x3')

# When a list, run_nets brackets every fused-layer launch with HIP events recorded on the
# launch stream and appends (tag, start_event, end_event): bench.py's live kernel timing.
EVENT_LOG = None

# Environment knobs (six, all read once at import): PWV_PRECISION (above), PWV_PERSIST, PWV_TWO_STREAMS, PWV_FOLD_FIRST, PWV_FUSE_TAIL,
# PWV_ASYNC.  Everything else below is a module attribute that tests / tools set directly.
#
# The scalar and shifter nets of a flow are independent dependency chains.  On the per-layer path (PWV_PERSIST=0, or a shape
# the persistent launch does not cover) TWO_STREAMS gives each chain its own HIP stream and half of the CUs per launch
# (G = 1), so one chain's launch gap / tail is covered by the other's bulk; otherwise both nets share one G = 2 launch.
TWO_STREAMS = os.environ.get('PWV_TWO_STREAMS', '1') != '0'
# layer 0 rebuilds the causal layer from the flow's scalar input / the head runs behind the last layer (tests switch them
# off to check the fused forms bit for bit against the separate launches)
FUSE_FIRST = True
FUSE_HEAD = True
# the net's last layer + head -- and the flow's affine -- run INSIDE the persistent launch (pwv_persist_args.tail_*, split-fp16 path):
# a flow is one launch instead of three.  Tests switch it off to check the fused form bit for bit against the separate launches.
FUSE_TAIL = os.environ.get('PWV_FUSE_TAIL', '1') != '0'      # (PWV_FUSE_TAIL=0: A/B runs of tools/runs/r05_run2.sh)
HOIST_P = True               # one projection GEMM per forward (project_all); False: every net projects for itself (cross-check in tests)
# PWV_FOLD_FIRST=0: layer 0 runs its filter|gate GEMM on the rebuilt causal-layer rows (eight MFMA k-steps) instead of on the
# four scalars they are a function of (one k-step); default: folded.  The only knob that changes bits (measured in HISTORY.md section 4, K2).
FOLD_FIRST = os.environ.get('PWV_FOLD_FIRST', '1') != '0'
# The residual layers of a stack as ONE persistent launch (csrc/pwv_stack_persist.hip, bit-identical results) instead of one
# launch per layer.  PWV_PERSIST = 0 | 1 | auto (default: wherever the library supports the shape; '1' forces it).
_pm = os.environ.get('PWV_PERSIST', 'auto')
PERSIST = {'0': False, '1': True}.get(_pm, 'auto')
PERSIST_AUTO_MAX_ROWS = 750000     # rows (N*T) per launch up to which 'auto' takes the persistent launch: measured on C3's model, same box, per-layer vs
                                   # persistent -- round 4: 160000 rows -5.5 %, 320000 -3.5 %, 480000 -0.6 %, 640000 +0.3 %, 960000 +1.5 % per step; round 5, with
                                   # the last layer + head + affine riding in the launch: 480000 -0.5 %, 640000 -0.6 %, 960000 +1.1 % (profiles/r05_ab_experiments.md r05_i)
PERSIST_MIN_UNITS = 0         # short inputs: fewer workgroups rather than ranges below this many units (0 = the library's default, 4)
PERSIST_MAX_LAYERS = 32       # longest run of layers in one persistent launch (a stack is cut into equal runs that hand the ring on)
# PWV_ASYNC=1: the reference-shaped calls (IAFVocoder / WaveNet / LinearIAFLayer __call__) only ENQUEUE, like the C ABI; the
# caller then owns IAFVocoder.verify() / engine.verify_enqueued().  Default: a call returns only after its own launches
# have completed and the library's sticky words (range guard, persistent give-up) are clean -- or it has rerun itself on
# the path that is (per-layer launches / exact fp32).  bench.py and graph.py opt out per call (verify=False) and verify for
# themselves; generate() passes verify=True, which outranks this knob: nothing unverified is ever written to disk.
ASYNC = os.environ.get('PWV_ASYNC', '0') == '1'
_persist_ws = {}          # (device, stream) -> zeroed workspace of the persistent launches on that stream
_persist_ws_retired = []  # workspaces a launch that gave up may still be writing to (kept alive, never handed out again)
_side_streams = {}

# ---- the library's two sticky words (pinned host memory, read without a device -> host copy) -----------------------------------
# words[0]: a persistent stack launch gave up (pwv_persist_args.status);  words[1]: an operand left the range of the split-fp16
# arithmetic (the `range_flag` argument of every entry point that checks).  Every THREAD has its own pair per device
# (pwv_status_words_alloc): everything a thread enqueues -- on any stream, eagerly or while capturing a graph -- reports into its
# own words, and its verification reads only those, so two threads serving two streams cannot consume or clear each other's
# flags.  (A captured graph keeps reporting into the words of the thread that captured it: graph.GraphedVocoder remembers them.)
_tls = threading.local()      # .words: {device index: StatusWords};  .depth: nesting of reference-shaped calls (verified_call)
_verify = _tls


_free_words: List[int] = []      # pinned pairs whose owners (threads, graphs) are gone: handed out again instead of allocated
_free_words_lock = threading.Lock()


class StatusWords(object):
    """One pinned, mapped pair {give-up word, range word}.  Pairs are POOLED: a serving process that spawns short-lived worker
    threads would otherwise leave one page-granular pinned allocation behind per thread.  A pair goes back to the
    pool when its last owner is collected -- the thread's table (a thread-local) and every GraphedVocoder that captured launches
    carrying it hold references -- and is cleared when it is handed out again.  (A launch the dead thread left in flight can
    still raise such a pair: the next owner then sees one spurious give-up, i.e. one rerun -- never a wrong result.)"""
    __slots__ = ('addr', '__weakref__')

    def __init__(self):
        with _free_words_lock:
            addr = _free_words.pop() if _free_words else None
        if addr is None:
            p = c_void_p()
            check(_lib.lib().pwv_status_words_alloc(ctypes.byref(p)), 'pwv_status_words_alloc')
            addr = p.value
        self.addr = addr
        self.persist = 0
        self.range = 0

    def __del__(self):
        try:
            with _free_words_lock:
                _free_words.append(self.addr)
        except Exception:      # interpreter shutdown
            pass

    def _w(self, k):
        return ctypes.c_int.from_address(self.addr + 4 * k)

    persist = property(lambda self: self._w(0).value, lambda self, v: setattr(self._w(0), 'value', int(v)))
    range = property(lambda self: self._w(1).value, lambda self, v: setattr(self._w(1), 'value', int(v)))


def current_words(device=None) -> StatusWords:
    """This thread's pair of sticky words for `device` (default: the current device); allocated on first use, kept for the life of
    the process (a launch that was handed them may outlive any scope here)."""
    idx = torch.cuda.current_device() if device is None else (device.index if isinstance(device, torch.device) and device.index is not None
                                                               else (device if isinstance(device, int) else torch.cuda.current_device()))
    d = getattr(_tls, 'words', None)
    if d is None:
        d = _tls.words = {}
    w = d.get(idx)
    if w is None:
        w = d[idx] = StatusWords()
    return w


def persist_status_ptr() -> int:
    return current_words().addr


def persist_status(words: Optional[StatusWords] = None) -> int:
    """The sticky give-up word of this thread's persistent stack launches: 0 = every launch that has completed so far ran to its
    end; otherwise a launch gave up and its outputs are invalid."""
    return (words or current_words()).persist


def poke_persist_status(code: int, words: Optional[StatusWords] = None) -> None:
    """(tests / tools) Write the give-up word as a launch that gave up would: a real give-up needs a second process on the GPU."""
    (words or current_words()).persist = code


def clear_persist_status(words: Optional[StatusWords] = None) -> None:
    (words or current_words()).persist = 0


# A give-up means the launch's workgroups were not all resident -- another process held CUs.  That is a condition, not a property of
# this process: the engine SUSPENDS the persistent launches for a number of forwards (per-layer launches: same arithmetic, same
# bits, ~5 % slower), then tries them again; consecutive give-ups double the pause (16, 32, ... 1024 forwards), a forward that
# completes on persistent launches resets it.
PERSIST_RETRY_AFTER = 16
_persist_cooldown = 0
_persist_backoff = PERSIST_RETRY_AFTER


# NOTE the suspension below is PROCESS-wide (one counter for all threads) although the sticky words are per thread: a give-up means
# the chip is shared with somebody, which concerns every thread's launches alike; the other threads' graphs are re-captured
# on the per-layer path at their next call (graph.GraphedVocoder._launch_mode).
def persist_suspended() -> bool:
    return _persist_cooldown > 0


def suspend_persist() -> None:
    global _persist_cooldown, _persist_backoff
    _persist_cooldown = _persist_backoff
    _persist_backoff = min(2 * _persist_backoff, 1024)
    # the launch that gave up may still have workgroups to start (they see the abort word and leave), and it did not clean up
    # after itself: its workspace is retired, the next persistent launch gets fresh zeros
    _persist_ws_retired.extend(_persist_ws.values())
    del _persist_ws_retired[:-8]
    _persist_ws.clear()


def resume_persist() -> None:
    """(tests / tools) End a suspension now."""
    global _persist_cooldown, _persist_backoff
    _persist_cooldown, _persist_backoff = 0, PERSIST_RETRY_AFTER


def note_forward() -> None:
    """One forward has been enqueued (an outermost reference-shaped call, a graph replay): the suspension counts down."""
    global _persist_cooldown
    if _persist_cooldown > 0:
        _persist_cooldown -= 1


def raise_if_persist_failed(words: Optional[StatusWords] = None) -> None:
    """A give-up of the persistent kernel (its workgroups were not all resident, a poll ran into its bound) suspends the
    persistent launches (suspend_persist) and raises: the caller reruns the forward, which then takes per-layer launches.  The
    range word is cleared with it: whatever consumed the invalid rows may have raised it, and the rerun must not meet it."""
    w = words or current_words()
    code = w.persist
    if code != 0:
        w.persist = 0
        w.range = 0
        suspend_persist()
        raise _lib.PwvPersistError('the persistent stack kernel gave up (code %d); its outputs are invalid -- per-layer launches are used '
                                   'for the next %d forwards, then the persistent launch is tried again; rerun the forward' % (code, _persist_cooldown))


def verify_enqueued(where: str = '', words: Optional[StatusWords] = None) -> None:
    """Wait for everything enqueued on this device and raise PwvPersistError / PwvRangeError if a launch that has completed
    left one of this thread's sticky words raised (what a caller of the enqueue-only forms owes before it reads a result)."""
    torch.cuda.synchronize()
    raise_if_persist_failed(words)
    raise_if_range_flag(where, words)


def verified_call(run, verify: Optional[bool] = None):
    """The contract of the reference-shaped entry points (IAFVocoder / WaveNet / LinearIAFLayer / SharedIAFLayer __call__):
    `run(precision_override)` enqueues the forward and returns its result tensor.  By default the OUTERMOST such call then waits
    for its launches and looks at this thread's two sticky words (pinned host memory: no device -> host copy):
      * a persistent stack launch gave up (its workgroups were not all resident): the engine suspends the persistent launches
        (suspend_persist: per-layer launches -- same arithmetic, same bits -- for a while, then another try) and the forward is rerun;
      * an operand left the range of the split-fp16 arithmetic (the reference computes in fp32, models.py:81-82): the forward
        is rerun with precision 'f32'.
    So `pred = model(wav, mel, is_training=False); pred.cpu()` (generate.py:38,68) yields a correct result or an exception,
    never inf from a silent fp16 overflow.  verify=False (or PWV_ASYNC=1, or a call under stream capture, or a call nested in
    another reference-shaped call) only enqueues; the caller then owns verify_enqueued() / IAFVocoder.verify()."""
    global _persist_backoff
    if verify is None:
        verify = not ASYNC
    depth = getattr(_verify, 'depth', 0)
    if depth == 0:
        note_forward()
    if depth > 0 or not verify or torch.cuda.is_current_stream_capturing():
        _verify.depth = depth + 1
        try:
            return run(None)
        finally:
            _verify.depth = depth
    _verify.depth = 1
    try:
        out = run(None)
        torch.cuda.current_stream().synchronize()
        if persist_status() != 0:
            clear_persist_status()
            clear_range_flag()          # (whatever consumed the invalid rows may have raised it)
            suspend_persist()
            import warnings
            warnings.warn('pwv: a persistent stack launch gave up (its workgroups were not all resident -- another process on this GPU?); '
                          'the forward is rerun on per-layer launches (same arithmetic, same bits), which the next %d forwards use too '
                          'before the persistent launch is tried again' % _persist_cooldown)
            out = run(None)
            torch.cuda.current_stream().synchronize()
            raise_if_persist_failed()
        elif _persist_cooldown == 0:
            _persist_backoff = PERSIST_RETRY_AFTER      # (a forward on persistent launches -- or one that never needed them -- went through)
        if range_flag_raised():
            clear_range_flag()
            import warnings
            warnings.warn("pwv: an input or activation left the exponent range of the split-fp16 arithmetic; the forward is rerun in exact fp32 "
                          "(precision='f32', the reference's arithmetic) on the same noise")
            out = run('f32')
            torch.cuda.current_stream().synchronize()
            raise_if_persist_failed()
            raise_if_range_flag("its rerun in precision 'f32'")
        return out
    finally:
        _verify.depth = 0


def _persist_runs(L: int, first: int = 1) -> List[Tuple[int, int]]:
    """(first layer, count) of the persistent launches that cover the residual layers `first` .. L-2 (first = 0: layer 0
    rebuilds the causal layer from the scalar input inside the launch, pwv_persist_args.x_first)."""
    Lp = L - 1 - first
    nchunks = -(-Lp // max(2, PERSIST_MAX_LAYERS))
    runs, j0 = [], first
    for c in range(nchunks):
        cnt = Lp // nchunks + (1 if c < Lp % nchunks else 0)
        runs.append((j0, cnt))
        j0 += cnt
    if len(runs) > 1 and runs[-1][1] < 2:       # (a run has at least two layers)
        runs[-2:] = [(runs[-2][0], runs[-2][1] + runs[-1][1])]
    return runs


def _use_persist(G: int, n: int, t: int, dilations, first: int = 1, tail_q: int = 0) -> bool:
    """Can layers first .. L-2 (and, with tail_q > 0, the last layer + head behind the last run) run as persistent launches?"""
    L = len(dilations)
    if PERSIST is False or _persist_cooldown > 0 or L < 4:
        return False
    if PERSIST == 'auto' and n * t > PERSIST_AUTO_MAX_ROWS:
        return False
    lib = _lib.lib()
    runs = _persist_runs(L, first)
    for j0, cnt in runs:    # 0 bytes = the library cannot run this shape persistently: per-layer launches
        pa = _lib.PersistArgs()
        pa.G, pa.n_layers, pa.N, pa.T = G, cnt, n, t
        pa.dilations = (ctypes.c_int * cnt)(*[int(d) for d in dilations[j0:j0 + cnt]])
        pa.min_units_per_workgroup = PERSIST_MIN_UNITS
        if tail_q > 0 and (j0, cnt) == runs[-1]:      # the tail's own look-back (the LAST dilation) is part of the answer
            pa.tail_q, pa.tail_dilation = tail_q, int(dilations[L - 1])
        if lib.pwv_persist_workspace_bytes(ctypes.byref(pa)) == 0:
            return False
    return True


PERSIST_ARGS_HOOK = None      # tests: called with the filled-in pwv_persist_args right before every persistent launch


def _net_streams(device):
    key = (device.type, device.index)
    if key not in _side_streams:
        _side_streams[key] = [torch.cuda.Stream(device=device), torch.cuda.Stream(device=device)]
    return _side_streams[key]


# ---- range guard of the split-fp16 arithmetic (include/pwv_hip.h, "Range guard") -------------------------------
F16_LIMIT = 65000.0          # fp16 max is 65504; stay below the value that rounds up to inf
_range_warned = set()
RANGE_LOG = None             # a list: every run-time check of a split-fp16 forward appends (kind, observed max, limit, pack-time bounds)
                             # -- diagnostics only (range_report): each entry costs a device -> host read


def _log_range(kind: str, x: torch.Tensor, limit: float, bounds=None) -> None:
    if RANGE_LOG is not None:
        RANGE_LOG.append((kind, float(x.abs().max()), float(limit), bounds))


def range_report(forward) -> dict:
    """How far a split-fp16 forward stays from fp16's exponent range, per operand class: runs `forward()` (eager, precision 'f16x3')
    with RANGE_LOG on and returns, per class, the limit the guard derived, the maximum it observed (run-time classes) or bounded
    (pack-time classes), and their quotient -- `range_margin` = the smallest quotient: the factor by which every weight / input of that
    class could grow before the guard (deliberately conservative) would send the forward to the exact-fp32 kernels.
      weights         max |w| after the exp2 scale folding              vs 65000            (pack time)
      residual        sum_i (||dense_i||_1 + |bias_i|) + |h|max bound   vs 65000            (pack time + observed flow input)
      head_operand    ||skip||_1 + |bias| (relu(skip) feeds postprocess1) vs 65000          (pack time)
      flow_input      max |x| entering a flow                           vs x_limit           (run time, checked on the device)
      mel             max |mel|                                         vs mel_limit         (run time, checked on the device)"""
    global RANGE_LOG
    saved, RANGE_LOG = RANGE_LOG, []
    try:
        forward()
        torch.cuda.synchronize()
        log = RANGE_LOG
    finally:
        RANGE_LOG = saved
    classes = {}

    def put(name, limit, observed):
        cur = classes.get(name)
        q = limit / observed if observed > 0 else float('inf')
        if cur is None or q < cur['margin']:
            classes[name] = {'limit': limit, 'observed': observed, 'margin': q}

    for kind, seen, limit, bounds in log:
        put(kind, limit, seen)
        for b in bounds or ():
            put('weights', F16_LIMIT, b['w_max'])
            put('head_operand', F16_LIMIT, b['skip_bound'])
            put('residual', F16_LIMIT, b['res_bound'] + b['c0_max'] + b['c_causal'] * seen)
    return {'classes': classes, 'range_margin': min((c['margin'] for c in classes.values()), default=float('inf'))}


def range_flag_ptr() -> int:
    """Address of this thread's sticky overflow flag (pinned host memory, also valid on the device)."""
    return current_words().addr + 4


def range_flag_raised(words: Optional[StatusWords] = None) -> bool:
    """True once a split-fp16 forward of this thread that has COMPLETED met an out-of-range operand (no synchronisation here)."""
    return (words or current_words()).range != 0


def clear_range_flag(words: Optional[StatusWords] = None) -> None:
    (words or current_words()).range = 0


def raise_if_range_flag(where: str = '', words: Optional[StatusWords] = None) -> None:
    if range_flag_raised(words):
        clear_range_flag(words)
        raise _lib.PwvRangeError("a split-fp16 ('f16x3') forward%s met an activation or input beyond fp16's exponent range "
                                 "(or a non-finite one): its result is not trustworthy -- rerun with precision='f32'"
                                 % (' (%s)' % where if where else ''))


def range_check_op(x: torch.Tensor, limit: float, kind: Optional[str] = 'mel') -> None:
    """Enqueue the check |x| <= limit (and finite) on the current stream; a violation raises the sticky flag."""
    if kind:
        _log_range(kind, x, limit)
    check(_lib.lib().pwv_range_check_f32(_ptr(x), x.numel(), float(limit), range_flag_ptr(), _stream()), 'pwv_range_check_f32')


def _stream() -> c_void_p:
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _require_cuda_f32(t: torch.Tensor, what: str) -> torch.Tensor:
    if not isinstance(t, torch.Tensor):
        raise TypeError('%s must be a torch.Tensor' % what)
    if not t.is_cuda:
        raise _lib.PwvError('%s must live on the GPU (cuda device); there is no CPU path' % what)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


class RepeatedCondition:
    """Frame-rate condition + the 'repeat' upsampling rule (models.py:131-133), kept lazy:
    sample t of the upsampled condition is ``frames[:, (t + offset) // hop, :]``.  WaveNet layers
    project it at frame rate instead of materialising the [N, T, C] tensor."""

    def __init__(self, frames: torch.Tensor, hop: int, offset: int, length: int):
        self.frames = frames          # [N, t_mel, C]
        self.hop, self.offset, self.length = int(hop), int(offset), int(length)
        if (self.length - 1 + self.offset) // self.hop >= frames.shape[1]:
            raise ValueError('length %d needs more than %d frames' % (length, frames.shape[1]))

    @property
    def shape(self):
        return (self.frames.shape[0], self.length, self.frames.shape[2])

    def materialize(self) -> torch.Tensor:
        n, t_mel, c = self.frames.shape
        out = torch.empty((n, self.length, c), dtype=torch.float32, device=self.frames.device)
        check(_lib.lib().pwv_upsample_repeat_f32(_ptr(self.frames), _ptr(out), n, t_mel, c, self.length,
                                                 self.hop, self.offset, _stream()), 'pwv_upsample_repeat_f32')
        return out


# --------------------------------------------------------------------------------------------
# thin op wrappers
# --------------------------------------------------------------------------------------------
def causal_conv_op(value: torch.Tensor, filter_: torch.Tensor, dilation: int) -> torch.Tensor:
    value = _require_cuda_f32(value, 'value')
    filter_ = _require_cuda_f32(filter_, 'filter_')
    if value.dim() != 3 or filter_.dim() != 3 or value.shape[2] != filter_.shape[1]:
        raise ValueError('causal_conv: value [N,T,Cin] / filter [W,Cin,Cout] mismatch: %s vs %s'
                         % (tuple(value.shape), tuple(filter_.shape)))
    n, t, cin = value.shape
    w, _, cout = filter_.shape
    out = torch.empty((n, t, cout), dtype=torch.float32, device=value.device)
    if out.numel() == 0:
        return out
    check(_lib.lib().pwv_causal_conv_f32(_ptr(value), _ptr(filter_), _ptr(out), n, t, cin, cout, w, int(dilation),
                                         _stream()), 'pwv_causal_conv_f32')
    return out


def linear_op(x2d: torch.Tensor, w: torch.Tensor, bias: Optional[torch.Tensor], relu: bool,
              precision: Optional[str] = None) -> torch.Tensor:
    """y = act(x @ w + bias).  precision 'f32': exact fp32 MFMA; otherwise (default) split-fp16 MFMA -- the
    arithmetic of the fused kernels in the same mode."""
    m, k = x2d.shape
    nout = w.shape[1]
    y = torch.empty((m, nout), dtype=torch.float32, device=x2d.device)
    fn = _lib.lib().pwv_linear_f32 if (precision or DEFAULT_PRECISION) == 'f32' else _lib.lib().pwv_linear_split_f32
    check(fn(_ptr(x2d), _ptr(w), _ptr(bias), _ptr(y), m, k, nout, int(relu), _stream()), 'pwv_linear')
    return y


def crop_time_op(x: torch.Tensor, t_out: int, offset: int) -> torch.Tensor:
    n, t_in, c = x.shape
    out = torch.empty((n, t_out, c), dtype=torch.float32, device=x.device)
    check(_lib.lib().pwv_crop_time_f32(_ptr(x), _ptr(out), n, t_in, c, t_out, offset, _stream()), 'pwv_crop_time_f32')
    return out


def logistic_noise_op(shape: Sequence[int], device, seed: int, offset: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    z = out if out is not None else torch.empty(tuple(shape), dtype=torch.float32, device=device)
    if out is not None and (tuple(out.shape) != tuple(shape) or out.dtype != torch.float32 or not out.is_contiguous()):
        raise ValueError('out must be a contiguous float32 tensor of shape %s' % (tuple(shape),))
    check(_lib.lib().pwv_logistic_noise_f32(_ptr(z), z.numel(), seed, offset, _stream()), 'pwv_logistic_noise_f32')
    return z


def logistic_noise_stream_op(z: torch.Tensor, state: torch.Tensor) -> torch.Tensor:
    """The sampler in its capturable form (pwv_logistic_noise_stream_f32): `state` = int64[4] on the device, {seed, offset, 0, skip};
    every launch -- every replay of a captured one -- draws the next z.numel() counters of the stream and moves the offset on."""
    if state.dtype != torch.int64 or state.numel() != 4 or not state.is_cuda or not z.is_contiguous() or z.dtype != torch.float32:
        raise ValueError('state must be int64[4] on the GPU, z a contiguous float32 tensor')
    check(_lib.lib().pwv_logistic_noise_stream_f32(_ptr(z), z.numel(), state.data_ptr(), _stream()), 'pwv_logistic_noise_stream_f32')
    return z


def logistic_noise_window(n: int, total_length: int, first_sample: int, window: int, device, seed: int, first_item: int = 0) -> torch.Tensor:
    """[n, window, 1] Logistic(0,1) noise for the samples [first_sample, first_sample + window) of the utterances
    first_item .. first_item + n - 1 of ONE counter-based stream in which utterance i, sample t is counter
    i * total_length + t: what a rank needs for its share of a sharded job (utterance shards: first_item = the rank's first
    utterance; time shards: first_sample = the window's start) -- every rank draws exactly the values an unsharded run
    with the same seed would have drawn there, with no communication."""
    z = torch.empty((n, window, 1), dtype=torch.float32, device=device)
    for i in range(n):
        logistic_noise_op((1, window, 1), device, seed=seed, offset=(first_item + i) * total_length + first_sample, out=z[i:i + 1])
    return z


def instance_norm_op(x: torch.Tensor, gamma: Optional[torch.Tensor], beta: Optional[torch.Tensor], eps: float = 1e-8) -> torch.Tensor:
    """modules.instance_normalization (modules.py:274-284) on a channels-last [N, T, C] tensor."""
    x = _require_cuda_f32(x, 'input')
    n, t, c = x.shape
    y = torch.empty_like(x)
    if x.numel() == 0:
        return y
    lib = _lib.lib()
    nbytes = lib.pwv_instance_norm_workspace_bytes(n, t, c)
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=x.device)
    check(lib.pwv_instance_norm_f32(_ptr(x), _ptr(y), n, t, c, _ptr(gamma), _ptr(beta), float(eps), ws.data_ptr(), nbytes, _stream()),
          'pwv_instance_norm_f32')
    return y


def channel_affine_op(x: torch.Tensor, scale: Optional[torch.Tensor], bias: Optional[torch.Tensor], relu: bool = False,
                      tile32_rows: int = 0, channels: int = 0, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """y = act(x * scale[c] + bias[c]) over the last axis (or over a tile32 buffer of `tile32_rows` x `channels`)."""
    y = out if out is not None else torch.empty_like(x)
    if x.numel() == 0:
        return y
    if tile32_rows:
        rows, c, tiled = tile32_rows, channels, 1
    else:
        c = x.shape[-1]
        rows, tiled = x.numel() // c, 0
    check(_lib.lib().pwv_channel_affine_f32(_ptr(x), _ptr(y), rows, c, _ptr(scale), _ptr(bias), tiled, int(relu), _stream()),
          'pwv_channel_affine_f32')
    return y


def add_op(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(a)
    if a.numel() % 4 == 0:
        check(_lib.lib().pwv_add_f32(_ptr(a), _ptr(b), _ptr(out), a.numel(), _stream()), 'pwv_add_f32')
        return out
    # odd sizes (scalar-channel tensors): the affine kernel's scalar path with a per-element "bias" is not available;
    # pad-free fallback through two launches on the multiple-of-4 prefix and the tail
    n4 = a.numel() // 4 * 4
    af, bf, of = a.reshape(-1), b.reshape(-1), out.reshape(-1)
    if n4:
        check(_lib.lib().pwv_add_f32(_ptr(af), _ptr(bf), _ptr(of), n4, _stream()), 'pwv_add_f32')
    of[n4:] = af[n4:] + bf[n4:]
    return out


def gate_op(f: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(f)
    check(_lib.lib().pwv_gate_f32(_ptr(f), _ptr(g), _ptr(out), f.numel(), _stream()), 'pwv_gate_f32')
    return out


def iaf_affine_op(z: torch.Tensor, s: torch.Tensor, b: torch.Tensor, sb_stride: int = 1) -> torch.Tensor:
    """out = z*s + b (modules.py:59); s/b may be strided views (shared net: stride 2)."""
    n, t = z.shape[0], z.shape[1]
    out = torch.empty_like(z)
    check(_lib.lib().pwv_iaf_front_f32(_ptr(z), s.data_ptr(), b.data_ptr(), sb_stride, _ptr(out), 0, None, None,
                                       n, t, 1, 4, _stream()), 'pwv_iaf_front_f32')
    return out


# --------------------------------------------------------------------------------------------
# per-net plan: packed weights resident in HBM
# --------------------------------------------------------------------------------------------
class NetPlan:
    def __init__(self, net, cond_mode: str, precision: int):
        """cond_mode: 'none' | 'frames' (hoisted projection) | 'samples' (per-sample cond GEMM)."""
        lib = _lib.lib()
        self.cond_mode = cond_mode
        self.precision = precision
        L = len(net.dilations)
        dev = net.device()
        use_skip = bool(net.use_skip_connection)
        cond_c = net.condition_channels if cond_mode == 'samples' else 0
        self.with_skip = use_skip
        self.layer_floats = lib.pwv_layer_packed_floats(int(use_skip), cond_c)
        self.packed_layers = torch.empty((L, self.layer_floats), dtype=torch.float32, device=dev)
        s = _stream()
        # the frame-rate projection operands (gc_filter|gc_gate and filter_bias|gate_bias of every layer, in the kernels'
        # column order, exp2 scales folded in) are packed by one small HIP launch per layer (pwv_pack_proj_f32)
        cc = net.condition_channels if cond_mode == 'frames' else 0
        self.proj_b = torch.empty((128 * L,), dtype=torch.float32, device=dev)
        self.proj_w = torch.empty((cc, 128 * L), dtype=torch.float32, device=dev) if cc else None
        # normalize='bn' (inference): every batch norm of the net is folded into these tensors (WaveNet.folded_variables)
        folded = net.folded_variables(cond_mode != 'none') if net.normalize == 'bn' else None
        self._lv = (lambda j: folded['layers'][j]) if folded else (lambda j: net.layer_variables(j, with_cond=cond_mode != 'none'))
        self._hv = folded['head'] if folded else net.head_variables()
        for j in range(L):
            v = self._lv(j)
            check(lib.pwv_pack_layer_f32(_ptr(v['filter']), _ptr(v['gate']), _ptr(v['dense']), _ptr(v.get('dense_bias')),
                                         _ptr(v['skip']) if use_skip else None,
                                         _ptr(v.get('skip_bias')) if use_skip else None,
                                         _ptr(v.get('gc_filter')) if cond_c else None,
                                         _ptr(v.get('gc_gate')) if cond_c else None,
                                         int(use_skip), cond_c, precision, _ptr(self.packed_layers[j]), s),
                  'pwv_pack_layer_f32')
            check(lib.pwv_pack_proj_f32(_ptr(v['gc_filter']) if cc else None, _ptr(v['gc_gate']) if cc else None,
                                        _ptr(v.get('filter_bias')), _ptr(v.get('gate_bias')), cc, j, L, _ptr(self.proj_w), _ptr(self.proj_b), s),
                  'pwv_pack_proj_f32')
        hv = self._hv
        last = self._lv(L - 1)
        self.head_floats = lib.pwv_head_packed_floats(net.out_channels)
        self.packed_head = torch.empty((self.head_floats,), dtype=torch.float32, device=dev)
        check(lib.pwv_pack_head_f32(None if use_skip else _ptr(last['skip']),
                                    None if use_skip else _ptr(last.get('skip_bias')),
                                    _ptr(hv['postprocess1']), _ptr(hv.get('postprocess1_bias')),
                                    _ptr(hv['postprocess2']), _ptr(hv.get('postprocess2_bias')),
                                    net.out_channels, precision, _ptr(self.packed_head), s), 'pwv_pack_head_f32')
        self.causal_bias = folded['causal_bias'] if folded else None      # batch norm behind the causal layer: its shift
        self._causal_filter_folded = folded['causal_filter'].contiguous() if folded else None
        self.causal_filter = self._causal_filter_folded if self._causal_filter_folded is not None else net.causal_filter()
        self.n_layers = L
        self.f16x3_ok, self.x_limit = True, 3.0e38
        if precision == _lib.PREC_F16X3:
            self._range_analysis(net, L, use_skip, cond_mode)
        # scalar-input nets (split-fp16 and fp32 paths): layer 0's filter|gate GEMM folded onto the four scalars it is a function
        # of (first_fold in the C ABI's argument structs); on the split-fp16 path the scalars themselves are then fp16 operands,
        # hence the extra bound on them
        self.first_fold = None
        if (FOLD_FIRST and self.f16x3_ok and net.in_channels == 1 and net.filter_width == 2
                and net.residual_channels == 64 and net.dilation_channels == 64 and self.causal_bias is None):
            v0 = self._lv(0)
            ff = torch.zeros((_lib.FIRST_FOLD_FLOATS,), dtype=torch.float32, device=dev)
            # (the fp16 storage mode reads the `hi` fragments of the split-fp16 layout, as it does for every other weight)
            pack = lib.pwv_pack_first_fold_f32 if precision == _lib.PREC_F32 else lib.pwv_pack_first_fold_f16x3
            check(pack(_ptr(self.causal_filter), _ptr(v0['filter']), _ptr(v0['gate']), _ptr(ff), s), 'pwv_pack_first_fold')
            if precision == _lib.PREC_F32:
                self.first_fold = ff
            elif bool(torch.isfinite(ff.view(torch.float16).float()).all()):
                self.first_fold = ff
                if precision == _lib.PREC_F16X3:
                    self.x_limit = min(self.x_limit, F16_LIMIT)

    def _range_analysis(self, net, L, use_skip, cond_mode):
        """Bound every operand the split-fp16 kernels convert to fp16 (weights after the exp2 scale folding; the residual
        stream |x_j| <= |h| + sum_i (||dense_i||_1 + |dense_bias_i|) because |tanh * sigmoid| < 1; relu(skip) feeding the
        head) from the weights alone.  What remains is a limit on the flow input, checked on the device at run time."""
        kf, kg = 2.8853900817779268, 1.4426950408889634
        lib = _lib.lib()
        dev = self.causal_filter.device
        stats = torch.zeros((L, 8), dtype=torch.float32, device=dev)
        for j in range(L):          # one small HIP launch per layer (pwv_range_stats_f32) instead of ~15 torch reductions
            v = self._lv(j)
            with_gc = cond_mode != 'none'
            check(lib.pwv_range_stats_f32(_ptr(v['filter']), _ptr(v['gate']), _ptr(v['dense']), _ptr(v.get('dense_bias')), _ptr(v['skip']),
                                          _ptr(v.get('skip_bias')), _ptr(v['gc_filter']) if with_gc else None,
                                          _ptr(v['gc_gate']) if with_gc else None, net.condition_channels if with_gc else 0,
                                          _ptr(stats[j]), _stream()), 'pwv_range_stats_f32')
        hv = self._hv
        extra = torch.stack([hv['postprocess1'].abs().max(), self.causal_filter.abs().sum(dim=(0, 1)).max()])
        st, (post1_max, c_causal) = stats.cpu().numpy(), extra.cpu().tolist()
        w_max = max(float((st[:, [0, 4]].max()) * kf), float(st[:, [1, 5]].max() * kg), float(st[:, 2:4].max()), post1_max)
        res_bound = float(st[:L - 1, 6].sum())
        skip_bound = float(st[:, 7].sum() if use_skip else st[L - 1, 7])
        # normalize='bn': the causal layer is followed by a folded batch norm, h = a0 * conv(x) + c0 -- the shift is part of |h|
        c0_max = float(self.causal_bias.abs().max()) if self.causal_bias is not None else 0.0
        budget = F16_LIMIT - res_bound - c0_max
        self.f16x3_ok = w_max < F16_LIMIT and skip_bound < F16_LIMIT and res_bound < F16_LIMIT and budget > 0
        self.x_limit = budget / c_causal if c_causal > 0 else 3.0e38
        # what the guard knows at pack time (range_report / tools/precision_report.py: limit vs observed per operand class)
        self.range_bounds = {'w_max': w_max, 'res_bound': res_bound, 'skip_bound': skip_bound, 'c_causal': c_causal, 'c0_max': c0_max}


_plan_cache: Dict[Tuple, Tuple[int, NetPlan]] = {}
_cond_cache = None      # (weakref(condition tensor), its version, precision, converted copy, stream it was made on)


def get_plan(net, cond_mode: str, precision: int) -> NetPlan:
    # the architecture is part of the key: the same store + scope can be re-used with another dilation list or with
    # use_skip_connection toggled (no variable is created then, so the store version alone does not change)
    key = (net.store.uid, net.full_scope, cond_mode, precision, tuple(int(d) for d in net.dilations),
           bool(net.use_skip_connection), bool(net.use_biases), net.in_channels, net.out_channels,
           net.condition_channels, net.filter_width, net.residual_channels, net.dilation_channels, net.skip_channels,
           net.normalize or '', bool(FOLD_FIRST))
    hit = _plan_cache.get(key)
    if hit is not None and hit[0] == net.store.version:
        return hit[1]
    plan = NetPlan(net, cond_mode, precision)
    _plan_cache[key] = (net.store.version, plan)   # read AFTER planning: creating variables bumps it
    return plan


def clear_plan_cache() -> None:
    _plan_cache.clear()


def _same_structure(a, b) -> bool:
    return (list(a.dilations) == list(b.dilations) and a.use_skip_connection == b.use_skip_connection
            and a.in_channels == b.in_channels and a.out_channels == b.out_channels
            and a.condition_channels == b.condition_channels and a.filter_width == b.filter_width)


def tail_fusable(prec: int) -> bool:
    """Does the persistent launch of this arithmetic run the net's last layer + head (+ affine) as its tail?"""
    return bool(FUSE_TAIL and FUSE_HEAD and prec in (_lib.PREC_F16X3, _lib.PREC_F32))      # (round 6: the exact-fp32 kernel has the tail too)


def _run_stack_persist(lib, nets, plans, projs, bufs, outs, x_first, x_limit, row_stride, cond_geom, n, t, s, prec, affine=None, tail=None):
    """Layers 0 .. L-2 as ONE persistent launch (layer 0 a launch of its own when it cannot rebuild the causal layer itself), with
    layer L-1 + the head behind it -- and, given `affine` = (x, out), the flow's affine out = x*s + b -- INSIDE that launch on the
    split-fp16 and exact-fp32 paths (FUSE_TAIL; otherwise one more launch for them); all nets of the flow in every launch, all on the current
    stream.  `bufs[g]` holds THREE tile32 buffers: the persistent launch rotates through them (include/pwv_hip.h,
    pwv_persist_args.x_ring).  Returns True when the affine was evaluated by the launch."""
    G, L = len(nets), plans[0].n_layers
    net0 = nets[0]
    hop, offset, frames = cond_geom
    stride = plans[0].layer_floats

    def layer_args(j, src, dst):
        la = _lib.LayerArgs()
        la.G = G
        for g in range(G):
            la.x_in[g], la.x_out[g] = bufs[g][src].data_ptr(), bufs[g][dst].data_ptr()
            la.packed[g] = plans[g].packed_layers.data_ptr() + 4 * stride * j
            la.proj[g] = projs[g].data_ptr() + 4 * 128 * j
        la.proj_row_stride = row_stride
        la.skip_init = 1
        la.N, la.T, la.dilation = n, t, int(net0.dilations[j])
        la.cond_hop, la.cond_offset, la.cond_frames = hop, offset, frames
        la.precision = prec
        return la

    if x_first is None:                     # layer 0 on the causal layer's buffer (bufs[g][0]) as a launch of its own
        la = layer_args(0, 0, 2)
        la.out_mode = _lib.OUT_RESIDUAL
        check(lib.pwv_wavenet_layer_f32(ctypes.byref(la), s), 'pwv_wavenet_layer_f32')

    # the residual layers (0 or) 1 .. L-2 as persistent launches of at most PERSIST_MAX_LAYERS layers each, handing the ring on
    rot, out_slot = 0, 2
    runs = _persist_runs(L, 1 if x_first is None else 0)
    Q = net0.out_channels
    if tail is None:
        tail = tail_fusable(prec)
    affine_fused = bool(tail and affine is not None and ((G == 2 and Q == 1) or (G == 1 and Q == 2)))
    for j0, cnt in runs:
        pa = _lib.PersistArgs()
        pa.G, pa.n_layers = G, cnt
        dil = (ctypes.c_int * cnt)(*[int(d) for d in net0.dilations[j0:j0 + cnt]])
        pa.dilations = dil
        for g in range(G):
            pa.x_ring[g] = bufs[g][0].data_ptr()
            pa.packed_layers[g] = plans[g].packed_layers.data_ptr() + 4 * stride * j0
            pa.proj[g] = projs[g].data_ptr() + 4 * 128 * j0
        pa.ring_stride = bufs[0][0].numel()
        pa.ring_rotation = rot
        pa.min_units_per_workgroup = PERSIST_MIN_UNITS
        if j0 == 0:                          # the net's layer 0 rebuilds the causal layer from the scalar input
            pa.x_first, pa.x_limit, pa.range_flag = _ptr(x_first), x_limit, range_flag_ptr()
            for g in range(G):
                pa.causal_filter[g] = plans[g].causal_filter.data_ptr()
            if FOLD_FIRST and all(p.first_fold is not None for p in plans):
                for g in range(G):
                    pa.first_fold[g] = plans[g].first_fold.data_ptr()
        pa.packed_layer_stride = stride
        pa.proj_row_stride = row_stride
        pa.N, pa.T = n, t
        pa.precision = prec
        pa.cond_hop, pa.cond_offset, pa.cond_frames = hop, offset, frames
        nbytes = lib.pwv_persist_workspace_bytes(ctypes.byref(pa))
        if nbytes == 0:
            raise _lib.PwvError('pwv_persist_workspace_bytes: %s' % lib.pwv_last_error().decode())
        # one zero-initialised workspace per (device, stream), kept: a launch leaves it clean, so none needs a zeroing kernel
        wkey = (bufs[0][0].device, torch.cuda.current_stream().cuda_stream)
        ws = _persist_ws.get(wkey)
        if ws is None or ws.numel() < nbytes:
            ws = _persist_ws[wkey] = torch.zeros((max(nbytes, 1 << 16),), dtype=torch.uint8, device=bufs[0][0].device)
        pa.workspace, pa.workspace_bytes, pa.workspace_clean = ws.data_ptr(), nbytes, 1
        pa.status = persist_status_ptr()
        last_run = (j0, cnt) == runs[-1]
        if tail and last_run:             # layer L-1 + head (+ affine) behind this run's layers, inside the launch
            for g in range(G):
                pa.tail_layer[g] = plans[g].packed_layers.data_ptr() + 4 * stride * (L - 1)
                pa.tail_head[g] = plans[g].packed_head.data_ptr()
                pa.tail_out[g] = outs[g].data_ptr()
            pa.tail_q, pa.tail_dilation = Q, int(net0.dilations[L - 1])
            if affine_fused:
                pa.affine_x, pa.affine_out = affine[0].data_ptr(), affine[1].data_ptr()
        ev = None
        if EVENT_LOG is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        if PERSIST_ARGS_HOOK is not None:
            PERSIST_ARGS_HOOK(pa)
        check(lib.pwv_wavenet_stack_persist_f32(ctypes.byref(pa), s), 'pwv_wavenet_stack_persist_f32')
        if ev is not None:
            ev[1].record()
            EVENT_LOG.append(('persist', ev[0], ev[1], G, cnt, 1 if j0 == 0 else 0, 1 if (tail and last_run) else 0,
                              int(lib.pwv_persist_short_input(ctypes.byref(pa)))))      # [7]: the short-input instantiation (round 6)
        out_slot = (cnt - 1 + rot) % 3      # where this run left its last layer
        rot = (out_slot + 1) % 3            # the next run's input buffer is (2 + rot') % 3 == out_slot
    if tail:
        return affine_fused
    spare = (out_slot + 1) % 3

    la = layer_args(L - 1, out_slot, spare)
    la.out_mode = _lib.OUT_GATED
    fuse_head = prec == _lib.PREC_F16X3 or (prec == _lib.PREC_F32 and FUSE_HEAD)
    if fuse_head:                        # the head runs inside the last layer's launch
        for g in range(G):
            la.head_packed[g] = plans[g].packed_head.data_ptr()
            la.head_out[g] = outs[g].data_ptr()
        la.head_q = net0.out_channels
    check(lib.pwv_wavenet_layer_f32(ctypes.byref(la), s), 'pwv_wavenet_layer_f32')
    if not fuse_head:
        ha = _lib.HeadArgs()
        ha.G = G
        for g in range(G):
            ha.in_[g], ha.packed[g], ha.out[g] = bufs[g][spare].data_ptr(), plans[g].packed_head.data_ptr(), outs[g].data_ptr()
        ha.N, ha.T, ha.Q = n, t, net0.out_channels
        ha.in_mode, ha.precision = _lib.HEAD_IN_GATED, prec
        check(lib.pwv_wavenet_head_f32(ctypes.byref(ha), s), 'pwv_wavenet_head_f32')
    return False


_bank_cache = {}      # tuple of plan ids -> (plans (kept alive), concatenated proj_w [K, total], proj_b [total], column offsets)


FUSE_PROLOGUE = os.environ.get('PWV_FUSE_PROLOGUE', '1') != '0'      # 'repeat' conditioning: mel range check + dense/relu + the projection GEMM as ONE launch (pwv_cond_project_f32)
# ... where it pays.  The fused launch recomputes the condition per 128-column block of the bank and holds 113 KB of LDS (one
# workgroup per CU): same box, fused vs three launches (profiles/r05_ab_experiments.md): C1 (201 frames x 2048 columns) -5 %,
# default model at 16000 samples (201 x 15360) level, C3 (2001 x 15360) +1.3 %, C4 +1 %.  Bit-identical either way, so the choice may
# depend on the size: frames x columns up to this many outputs take the one launch.
FUSE_PROLOGUE_MAX_OUTPUTS = 1 << 21


def _projection_bank(nets: Sequence, channels: int, prec: int):
    """(plans, [C, total] weights, [total] biases, column offsets) of the one projection GEMM of a forward, or None when fewer than
    two nets can share it (not fused-capable, another arithmetic after the range fallback)."""
    for _ in range(2):
        # (a first forward creates its variables while the plans are built, and every new variable outdates the plans
        # made before it: plan again once the store has stopped changing, so that run_nets finds these very plans)
        versions = [net.store.version for net in nets]
        plans = []
        for net in nets:
            if not getattr(net, 'fused_supported', None) or channels != net.condition_channels:
                continue
            if not net.fused_supported(_FrameShape(channels)):
                continue
            p = get_plan(net, 'frames', prec)
            if prec == _lib.PREC_F16X3 and not (p.f16x3_ok and p.x_limit > 0):
                continue
            if all(p is not q for q in plans):
                plans.append(p)
        if versions == [net.store.version for net in nets]:
            break
    if len(plans) < 2:
        return None
    key = tuple(id(p) for p in plans)
    hit = _bank_cache.get(key)
    if hit is None:
        if len(_bank_cache) >= 8:
            _bank_cache.clear()
        offs, total = [], 0
        for p in plans:
            offs.append(total)
            total += p.proj_w.shape[1]
        hit = (plans, torch.cat([p.proj_w for p in plans], dim=1).contiguous(), torch.cat([p.proj_b for p in plans]).contiguous(), offs)
        _bank_cache[key] = hit
    return hit


class _FrameShape(RepeatedCondition):
    """(what WaveNet.fused_supported looks at: a frame-rate condition with this many channels)"""

    def __init__(self, channels):
        self.frames = torch.empty((0, 1, channels))
        self.hop = self.offset = self.length = 0


def project_all(nets: Sequence, cond, precision: Optional[str] = None) -> None:
    """The frame-rate projections P of EVERY net of a forward (all flows) as one GEMM: P depends on the mel frames only,
    not on a flow's input, so none of it has to sit between two flows (per flow that was two small launches plus their
    cross-stream dependencies on the critical path).  The result is attached to the RepeatedCondition; run_nets picks its
    nets' column blocks out of it (row stride = all columns).  Nets that cannot use it (not fused-capable, a different
    arithmetic after the range fallback) simply project for themselves as before."""
    if not HOIST_P or not isinstance(cond, RepeatedCondition) or not nets or getattr(cond, 'proj_bank', None) is not None:
        return
    prec = PRECISIONS[precision or DEFAULT_PRECISION]
    bank = _projection_bank(nets, cond.frames.shape[2], prec)
    if bank is None:
        return
    plans, w_all, b_all, offs = bank
    n, frames, c = cond.frames.shape
    f2d = _require_cuda_f32(cond.frames, 'frames').reshape(n * frames, c)
    p_all = linear_op(f2d, w_all, b_all, relu=False, precision=precision or DEFAULT_PRECISION)
    cond.proj_bank = {id(p): p_all[:, o:o + p.proj_w.shape[1]] for p, o in zip(plans, offs)}


def repeat_condition_with_projections(nets: Sequence, melspec: torch.Tensor, dense: torch.Tensor, hop: int, length: int,
                                      precision: Optional[str], mel_limit: Optional[float]):
    """The prologue of a 'repeat'-conditioned forward as ONE launch (pwv_cond_project_f32): the range check of the mel, the
    frame-rate condition relu(mel @ dense) (models.py:128-130) and the projections of every net (project_all) -- bit-identical to
    the three launches it replaces.  Returns the RepeatedCondition with its projection bank attached, or None when the shape is
    outside the fused kernel's (the caller then takes the separate launches)."""
    name = precision or DEFAULT_PRECISION
    if not (FUSE_PROLOGUE and HOIST_P) or name == 'f32' or not nets:
        return None
    n, t_mel, n_mels = melspec.shape
    c = dense.shape[1]
    if n_mels % 8 or c % 8 or n_mels > 80 or c > 80:
        return None
    bank = _projection_bank(nets, c, PRECISIONS[name])
    if bank is None:
        return None
    plans, w_all, b_all, offs = bank
    m = n * t_mel
    if m * w_all.shape[1] > FUSE_PROLOGUE_MAX_OUTPUTS:
        return None
    frames = torch.empty((n, t_mel, c), dtype=torch.float32, device=melspec.device)
    p_all = torch.empty((m, w_all.shape[1]), dtype=torch.float32, device=melspec.device)
    flag = range_flag_ptr() if (name == 'f16x3' and mel_limit is not None) else None
    if flag is not None:
        _log_range('mel', melspec, mel_limit)
    check(_lib.lib().pwv_cond_project_f32(_ptr(melspec), _ptr(dense), n_mels, _ptr(w_all), _ptr(b_all), _ptr(frames), _ptr(p_all), m, c,
                                          w_all.shape[1], float(mel_limit or 0.0), flag, _stream()), 'pwv_cond_project_f32')
    cond = RepeatedCondition(frames, hop, hop // 2, length)
    cond.proj_bank = {id(p): p_all[:, o:o + p.proj_w.shape[1]] for p, o in zip(plans, offs)}
    return cond


def run_nets(nets: Sequence, x: torch.Tensor, cond, precision: Optional[str] = None,
             max_workgroups: int = 0) -> List[torch.Tensor]:
    """Evaluate 1 or 2 structurally identical fused-capable WaveNets on the same input/condition.
    Returns one [N, T, Q] tensor per net."""
    return _run_nets(nets, x, cond, precision, max_workgroups, None)[0]


def run_flow(nets: Sequence, x: torch.Tensor, cond, precision: Optional[str] = None) -> torch.Tensor:
    """One IAF flow (modules.py:53-60): out = x * scale + shift with (scale, shift) = the outputs of two scalar-input nets, or
    the two outputs of one shared net.  On the default path the whole flow is ONE persistent launch (the affine is evaluated
    inside it, pwv_persist_args.affine_x); otherwise the nets' launches are followed by the affine kernel."""
    x = _require_cuda_f32(x, 'input')
    out = torch.empty_like(x)
    outs, done = _run_nets(nets, x, cond, precision, 0, out)
    if done:
        return out
    if len(outs) == 2:
        return iaf_affine_op(x, outs[0], outs[1], 1)
    flat = outs[0].reshape(-1)
    return iaf_affine_op(x, flat, flat[1:], 2)


def _run_nets(nets: Sequence, x: torch.Tensor, cond, precision: Optional[str], max_workgroups: int, affine_out: Optional[torch.Tensor]):
    """run_nets / run_flow: ([one [N, T, Q] tensor per net], whether `affine_out` = x*s + b was written by the launches)."""
    lib = _lib.lib()
    prec = PRECISIONS[precision or DEFAULT_PRECISION]
    x = _require_cuda_f32(x, 'input_batch')
    if x.dim() != 3:
        raise ValueError('input_batch must be [N, T, C], got %s' % (tuple(x.shape),))
    n, t, qin = x.shape
    G = len(nets)
    assert 1 <= G <= _lib.PWV_MAX_NETS
    net0 = nets[0]
    for other in nets[1:]:
        assert _same_structure(net0, other)
    if qin != net0.in_channels:
        raise ValueError('input has %d channels, net expects %d' % (qin, net0.in_channels))
    if n * t == 0:
        return [torch.empty((n, t, net0.out_channels), dtype=torch.float32, device=x.device) for _ in nets], False
    dev = x.device
    s = _stream()

    # ---- conditioning mode ------------------------------------------------------------------
    cond_t = None
    hop = offset = frames_per_utt = 0
    if cond is None:
        mode = 'none'
    elif isinstance(cond, RepeatedCondition):
        mode = 'frames'
        if cond.length != t or cond.frames.shape[0] != n:
            raise ValueError('condition %s does not match input %s' % (cond.shape, tuple(x.shape)))
        if cond.frames.shape[2] != net0.condition_channels:
            raise ValueError('condition has %d channels, net expects %d' % (cond.frames.shape[2], net0.condition_channels))
        hop, offset, frames_per_utt = cond.hop, cond.offset, cond.frames.shape[1]
    else:
        mode = 'samples'
        cond_t = _require_cuda_f32(cond, 'condition_batch')
        if tuple(cond_t.shape) != (n, t, net0.condition_channels):
            raise ValueError('condition_batch %s does not match input %s / %d channels'
                             % (tuple(cond_t.shape), tuple(x.shape), net0.condition_channels))
    half = prec == _lib.PREC_F16
    if half and (qin != 1 or net0.use_skip_connection):
        raise _lib.PwvError("precision 'f16' supports scalar-input nets without skip accumulation only")
    plans = [get_plan(net, mode, prec) for net in nets]
    if prec == _lib.PREC_F16X3 and not all(p.f16x3_ok and p.x_limit > 0 for p in plans):
        # some weight / bound leaves fp16's exponent range: this net runs in the exact fp32 arithmetic instead
        key = tuple(net.full_scope for net in nets)
        if key not in _range_warned:
            _range_warned.add(key)
            import warnings
            warnings.warn("pwv: weights of %s exceed the range of the split-fp16 arithmetic; using precision 'f32' for it" % (key,))
        return _run_nets(nets, x, cond, 'f32', max_workgroups, affine_out)
    x_limit = min(p.x_limit for p in plans)
    if prec == _lib.PREC_F16X3:
        _log_range('flow_input', x, x_limit, [p.range_bounds for p in plans])
    L = plans[0].n_layers
    for net, plan in zip(nets, plans):
        assert plan.n_layers == len(net.dilations) and plan.with_skip == bool(net.use_skip_connection)
    rows = n * t

    def tile_buf(channels, dtype=torch.float32):
        return torch.empty((lib.pwv_tile32_floats(rows, channels),), dtype=dtype, device=dev)

    if cond_t is not None:
        # per-sample condition -> the form the kernels of this precision read: fp32 tile32 ('f32'), pre-split fp16
        # hi / lo planes ('f16x3': split once per forward instead of once per net-layer), fp16 blocks ('f16').
        # Every flow of a forward pass is handed the same tensor: convert it once (the entry is valid while that
        # very tensor object is alive and has not been written to)
        global _cond_cache
        hit = _cond_cache
        if (hit is not None and hit[0]() is cond_t and hit[1] == cond_t._version and hit[2] == prec
                and hit[4] == torch.cuda.current_stream()):
            cond_t = hit[3]
        else:
            cc = cond_t.shape[2]
            if prec == _lib.PREC_F32:
                ct = tile_buf(cc)
                check(lib.pwv_rows_to_tile32_f32(_ptr(cond_t), _ptr(ct), rows, cc, s), 'pwv_rows_to_tile32_f32')
            elif prec == _lib.PREC_F16X3:
                ct = torch.empty((2 * lib.pwv_tile32_floats(rows, cc),), dtype=torch.float16, device=dev)
                check(lib.pwv_cond_split_f16(_ptr(cond_t), _ptr(ct), n, t, cc, s), 'pwv_cond_split_f16')
            else:
                ct = tile_buf(cc, torch.float16)
                check(lib.pwv_cond_to_f16(_ptr(cond_t), _ptr(ct), n, t, cc, s), 'pwv_cond_to_f16')
            _cond_cache = (weakref.ref(cond_t), cond_t._version, prec, ct, torch.cuda.current_stream())
            cond_t = ct

    # ---- frame-rate projection P (or bias-only row) -------------------------------------------
    main = torch.cuda.current_stream()
    use_skip = bool(net0.use_skip_connection)
    # layer 0 rebuilds the causal layer's output from the scalar input itself (pwv_layer_args.x_first), so the [rows, 64]
    # front buffer is neither written nor read
    first_fused = (FUSE_FIRST and prec in (_lib.PREC_F16X3, _lib.PREC_F32, _lib.PREC_F16) and qin == 1 and net0.filter_width == 2
                   and net0.residual_channels == 64 and not net0.use_skip_connection and plans[0].causal_bias is None)
    persist_able = ((prec == _lib.PREC_F32 or (prec == _lib.PREC_F16X3 and FUSE_HEAD)) and mode != 'samples' and not use_skip
                    and max_workgroups == 0)
    # the tail (last layer + head inside the launch) has a look-back of its own -- the stack's LAST dilation: ask for the launch that
    # will be made, and keep the persistent layers with the tail as a launch of its own where only the tail does not fit
    tail = bool(persist_able and tail_fusable(prec))
    persist = persist_able and _use_persist(G, n, t, net0.dilations, 0 if first_fused else 1, net0.out_channels if tail else 0)
    if persist_able and tail and not persist:
        tail = False
        persist = _use_persist(G, n, t, net0.dilations, 0 if first_fused else 1, 0)
    two = G == 2 and TWO_STREAMS and max_workgroups == 0 and not persist
    side = _net_streams(dev) if two else None
    row_stride = 128 * L
    bank = getattr(cond, 'proj_bank', None) if mode == 'frames' else None
    if bank is not None and all(id(p) in bank for p in plans):
        projs = [bank[id(p)] for p in plans]          # column blocks of the forward's one projection GEMM (project_all)
        row_stride = projs[0].stride(0)
    elif mode == 'frames':
        f2d = _require_cuda_f32(cond.frames, 'frames').reshape(n * frames_per_utt, -1)
        if two:
            # net 0's P on the main stream, net 1's on its own stream: its chain then starts one small GEMM
            # later than net 0's, which keeps the two chains out of phase (they would otherwise run in
            # lockstep and hit their launch gaps / tails together)
            pname = precision or DEFAULT_PRECISION
            projs = [linear_op(f2d, plans[0].proj_w, plans[0].proj_b, relu=False, precision=pname), None]
            side[1].wait_stream(main)
            with torch.cuda.stream(side[1]):
                projs[1] = linear_op(f2d, plans[1].proj_w, plans[1].proj_b, relu=False, precision=pname)
        else:
            projs = [linear_op(f2d, p.proj_w, p.proj_b, relu=False, precision=precision or DEFAULT_PRECISION) for p in plans]
    else:
        projs = [p.proj_b.reshape(1, -1) for p in plans]

    # ---- causal layer (modules.py:174-183) ----------------------------------------------------
    R = net0.residual_channels
    if persist:      # the persistent launch rotates through three buffers of one allocation (pwv_persist_args.x_ring)
        bufs = [list(torch.empty((3, lib.pwv_tile32_floats(rows, R)), dtype=torch.float32, device=dev).unbind(0)) for _ in nets]
    else:
        bufs = [[tile_buf(R, torch.float16 if half else torch.float32) for _ in range(2)] for _ in nets]
    # split-fp16 and fp32 kernels: layer 0 rebuilds the causal layer's output from the scalar input itself
    # (pwv_layer_args.x_first), so the [rows, 64] front buffer is neither written nor read
    if prec == _lib.PREC_F16X3 and not first_fused:
        range_check_op(x, x_limit, kind=None)          # (layer 0 checks its scalar input itself when it rebuilds the causal layer)
    if first_fused:
        pass
    elif qin == 1:
        filt = (c_void_p * G)(*[p.causal_filter.data_ptr() for p in plans])
        hout = (c_void_p * G)(*[b[0].data_ptr() for b in bufs])
        front = lib.pwv_iaf_front_f16 if half else lib.pwv_iaf_front_f32
        check(front(_ptr(x), None, None, 1, None, G, filt, hout, n, t, net0.filter_width, R, s), 'pwv_iaf_front')
        for g, p in enumerate(plans):
            if p.causal_bias is not None:      # normalize='bn': the shift of the batch norm behind the causal layer (modules.py:182)
                if half:
                    raise _lib.PwvError("precision 'f16' does not support normalize='bn'")
                channel_affine_op(bufs[g][0], None, p.causal_bias, tile32_rows=rows, channels=R, out=bufs[g][0])
    else:
        for g, p in enumerate(plans):      # multi-channel input: generic causal conv, then into tile32
            hrows = torch.empty((n, t, R), dtype=torch.float32, device=dev)
            check(lib.pwv_causal_conv_f32(_ptr(x), _ptr(p.causal_filter), _ptr(hrows), n, t, qin, R,
                                          net0.filter_width, 1, s), 'pwv_causal_conv_f32')
            check(lib.pwv_rows_to_tile32_f32(_ptr(hrows), _ptr(bufs[g][0]), rows, R, s), 'pwv_rows_to_tile32_f32')


    skips = [tile_buf(net0.skip_channels) for _ in nets] if use_skip else None
    Q = net0.out_channels
    outs = [torch.empty((n, t, Q), dtype=torch.float32, device=dev) for _ in nets]

    if persist:
        aff = (x, affine_out) if (affine_out is not None and qin == 1) else None
        done = _run_stack_persist(lib, nets, plans, projs, bufs, outs, x if first_fused else None, x_limit, row_stride,
                                  (hop, offset, frames_per_utt) if mode == 'frames' else (0, 0, 0), n, t, s, prec, aff, tail)
        return outs, done
    if two:
        for g in range(2):
            side[g].wait_stream(main)
    # the whole stack + head in ONE C call (launches interleaved over the two streams)
    sa = StackArgs()
    sa.G, sa.n_layers = G, L
    dil = (ctypes.c_int * L)(*[int(d) for d in net0.dilations])
    sa.dilations = dil
    for g in range(G):
        sa.buf0[g], sa.buf1[g] = bufs[g][0].data_ptr(), bufs[g][1].data_ptr()
        sa.packed_layers[g] = plans[g].packed_layers.data_ptr()
        sa.proj[g] = projs[g].data_ptr()
        sa.skip[g] = skips[g].data_ptr() if use_skip else None
        sa.packed_head[g] = plans[g].packed_head.data_ptr()
        sa.out[g] = outs[g].data_ptr()
    sa.packed_layer_stride = plans[0].layer_floats
    sa.proj_row_stride = row_stride
    sa.cond = _ptr(cond_t)
    sa.cond_channels = net0.condition_channels if mode == 'samples' else 0
    sa.Q, sa.N, sa.T = Q, n, t
    sa.cond_hop, sa.cond_offset, sa.cond_frames = (hop, offset, frames_per_utt) if mode == 'frames' else (0, 0, 0)
    sa.precision = prec
    sa.max_workgroups = max_workgroups
    sa.separate_head = 0 if FUSE_HEAD else 1
    if first_fused:
        sa.x_first = _ptr(x)
        sa.x_limit = x_limit
        sa.range_flag = range_flag_ptr()
        for g in range(G):
            sa.causal_filter[g] = plans[g].causal_filter.data_ptr()
        if FOLD_FIRST and all(p.first_fold is not None for p in plans):
            for g in range(G):
                sa.first_fold[g] = plans[g].first_fold.data_ptr()
    streams = (c_void_p * 2)(side[0].cuda_stream if two else s.value, side[1].cuda_stream if two else None)
    evs = []
    if EVENT_LOG is not None and L > 1:
        # bench.py's live kernel timing: the library records one event pair per chain around its run of residual-layer
        # launches (pwv_stack_args.ev_begin / ev_end) -- the production launch path plus two event records
        for c, so in enumerate(side[:2] if two else [main]):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(so)       # materialises the hipEvent_t handles; the library re-records them
            e1.record(so)
            sa.ev_begin[c], sa.ev_end[c] = e0.cuda_event, e1.cuda_event
            evs.append((e0, e1))
    check(lib.pwv_wavenet_stack_f32(ctypes.byref(sa), streams), 'pwv_wavenet_stack_f32')
    for e0, e1 in evs:
        EVENT_LOG.append(('layer_residual', e0, e1, 1 if two else G, L - 1))
    if two:
        for g in range(2):
            main.wait_stream(side[g])
    return outs, False
