"""A give-up is a condition of the GPU, not of the process: does the persistent launch COME BACK?

Process A (this script) runs verified forwards of the default model (C3 shape) in a loop and logs, per forward, which launch path
the engine took.  After `--before` forwards it starts a CO-TENANT (a second process on the same GPU running the same forwards
for `--tenant-seconds`): both processes' persistent launches want every CU, so launches give up (bounded polls, 20 ms), the calls
repair themselves on per-layer launches and the engine suspends the persistent path (engine.suspend_persist: 16 forwards, doubling
per consecutive give-up).  When the co-tenant has left, the suspension counts down and the persistent launches return.

    python tools/co_tenant_recovery.py            # prints one line per phase change and a summary
"""
import argparse
import os
import subprocess
import sys
import time
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def make_model(length):
    import torch
    from pwv_amd.hparam import hparam as hp
    from pwv_amd.models import IAFVocoder
    from pwv_amd.variables import VariableStore
    hp.set_hparam_yaml('bench/c3')
    dev = torch.device('cuda', 0)
    store = VariableStore(device=dev, seed=2)
    model = IAFVocoder(batch_size=1, length=length, store=store)
    mel = (torch.rand((1, 1 + length // hp.signal.hop_length, hp.signal.n_mels)) * 2 - 1).to(dev)
    return model, mel


def tenant(seconds, length):
    import torch
    model, mel = make_model(length)
    from pwv_amd import engine
    from pwv_amd._lib import PwvError
    t_end = time.time() + seconds
    n = gave_up = 0
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        while time.time() < t_end:
            # keeps the GPU's queue full: bursts of enqueue-only forwards (a caller that synchronises after every forward leaves
            # gaps in which the other process's launch simply runs alone)
            try:
                for _ in range(16):
                    model(None, mel, is_training=False, verify=False)
                    n += 1
                model.verify()
            except PwvError:
                gave_up += 1
                engine.resume_persist()          # (the tenant keeps insisting on persistent launches: the worst neighbour)
    import torch
    torch.cuda.synchronize()
    print('tenant: %d forwards in %.1f s, %d bursts with a give-up' % (n, seconds, gave_up), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--tenant', action='store_true')
    ap.add_argument('--tenant-seconds', type=float, default=4.0)
    ap.add_argument('--before', type=int, default=30)
    ap.add_argument('--after-seconds', type=float, default=6.0)
    ap.add_argument('--length', type=int, default=160000)
    ap.add_argument('--poke', type=int, default=0, help='no co-tenant: write the give-up word from the host behind this many forwards (what a launch '
                                                         'that gave up would do), and again right after the suspension ends (the back-off doubles)')
    a = ap.parse_args()
    if a.tenant:
        return tenant(a.tenant_seconds, a.length)
    import torch
    from pwv_amd import engine
    model, mel = make_model(a.length)
    model(None, mel, is_training=False)          # variables, plans
    log = []                                     # (t, path, ms, repaired)
    t0 = time.time()
    proc, proc_end = None, None

    def one():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter('always')
            suspended = engine.persist_suspended()
            t = time.time()
            model(None, mel, is_training=False)
            ms = (time.time() - t) * 1e3
        repaired = any('gave up' in str(x.message) for x in w)
        log.append((time.time() - t0, 'per-layer' if suspended else 'persistent', ms, repaired))

    if a.poke:
        real, state = engine._run_stack_persist, {'n': 0, 'at': [a.poke, a.poke + 1 + engine.PERSIST_RETRY_AFTER + 1]}

        def spy(*args, **kw):
            r = real(*args, **kw)
            if state['at'] and len(log) >= state['at'][0]:
                state['at'].pop(0)
                engine.poke_persist_status(4)
            return r
        engine._run_stack_persist = spy
        for _ in range(a.poke + 4 * engine.PERSIST_RETRY_AFTER + 40):
            one()
        prev = None
        for k, (t, path, ms, rep) in enumerate(log):
            key = (path, rep)
            if key != prev:
                print('forward %4d  %-10s %s  (%.2f ms)' % (k, path, 'GIVE-UP (poked), repaired inside the call' if rep else '', ms))
                prev = key
        sus = [k for k, x in enumerate(log) if x[1] == 'per-layer']
        print('forwards %d, give-ups %d, forwards on per-layer launches %d (engine.PERSIST_RETRY_AFTER = %d per pause; a give-up on the first try '
              'after a pause doubles it); back on persistent launches at the end: %s'
              % (len(log), sum(1 for x in log if x[3]), len(sus), engine.PERSIST_RETRY_AFTER, log[-1][1] == 'persistent'))
        return
    for _ in range(a.before):
        one()
    proc = subprocess.Popen([sys.executable, os.path.abspath(__file__), '--tenant', '--tenant-seconds', str(a.tenant_seconds), '--length', str(a.length)])
    t_spawn = time.time() - t0
    while proc.poll() is None:
        one()
    t_gone = time.time() - t0
    t_stop = time.time() + a.after_seconds
    while time.time() < t_stop:
        one()
    # phases
    print('co-tenant started at %.2f s (its first seconds are imports), left at %.2f s' % (t_spawn, t_gone))
    prev = None
    for t, path, ms, rep in log:
        key = (path, rep)
        if key != prev:
            print('%7.2f s  %-10s %s  (%.2f ms)' % (t, path, 'GIVE-UP, repaired inside the call' if rep else '', ms))
            prev = key
    give_ups = sum(1 for x in log if x[3])
    tail = [x for x in log if x[0] > t_gone + 0.5]
    back = [x for x in tail if x[1] == 'persistent' and not x[3]]
    import statistics
    first = [x[2] for x in log[:a.before]]
    print('forwards %d, give-ups %d; before the co-tenant: median %.2f ms (persistent); after it left: %d of %d forwards persistent, median %.2f ms'
          % (len(log), give_ups, statistics.median(first), len(back), len(tail), statistics.median([x[2] for x in back]) if back else float('nan')))
    print('RECOVERED' if back and len(back) > len(tail) // 2 else 'NOT RECOVERED')


if __name__ == '__main__':
    main()
