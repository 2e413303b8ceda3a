"""GPU: event timeline of one persistent launch on a SHORT input (needs the -DPWV_PTRACE build, tools/build_ptrace.sh):
   PWV_LIB=tools/libpwv_ptrace.so python tools/persist_timeline.py [rows] [layers] [G] [workgroup to print]

Every wave stamps the chip-wide 100 MHz clock (s_memrealtime) at: 1 unit computed (GEMM2 done), 2 its stores issued, 3 own stores
acknowledged (vmcnt(0) in front of a wait), 4 dependencies of the next task satisfied, 5 rows arrived + look-back split, 6 top drained
(P row there), 7 workgroup progress word published, 8 tail entered, 9 head weights resident, 10 tail unit done, 11 workgroup done.
The script splits a layer's period into its links: compute (6 -> 1), store issue (1 -> 2), acknowledgement (2 -> 3), producer-done -> consumer sees it
(3 of the LAST producer -> 4), row loads (4 -> 5), P row (5 -> 6).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from pwv_amd import engine
from pwv_amd.modules import WaveNet
from pwv_amd.variables import VariableStore

NAMES = {1: 'computed', 2: 'stores issued', 3: 'stores acked', 4: 'deps ok', 5: 'rows in', 6: 'top drained', 7: 'PROGRESS WORD', 8: 'tail entered',
         9: 'head weights in', 10: 'tail unit done', 11: 'wg done', 12: 'WAR ok', 13: 'stores acked (WAR wait)', 14: 'early half done', 15: 'P requested', 16: 'deps checked', 17: 'P landed', 18: 'look-back requested'}


def main():
    rows = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
    L = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    G = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    show = int(sys.argv[4]) if len(sys.argv) > 4 else 60
    dev = torch.device('cuda', 0)
    d10 = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512]
    dil = (d10 * 3)[:L]
    store = VariableStore(device=dev, seed=3)
    kw = dict(batch_size=1, dilations=dil, filter_width=2, residual_channels=64, dilation_channels=64, skip_channels=128,
              quantization_channels=1, use_biases=True, condition_channels=80, use_skip_connection=False, is_training=False, store=store)
    nets = [WaveNet(name='n%d' % g, **kw) for g in range(G)]
    hop = 80
    g = torch.Generator().manual_seed(0)
    x = torch.randn((1, rows, 1), generator=g).to(dev)
    frames = torch.rand((1, rows // hop + 1, 80), generator=g).to(dev)
    cond = engine.RepeatedCondition(frames, hop, hop // 2, rows)
    nw = 2 * 256 * 8
    trace = torch.zeros((nw, 24), dtype=torch.int64, device=dev)
    ev = torch.zeros((nw, 64, 4), dtype=torch.int64, device=dev)
    os.environ['PWV_PTRACE_PTR'] = str(trace.data_ptr())
    os.environ['PWV_PTRACE_EV_PTR'] = str(ev.data_ptr())
    engine.PERSIST = True
    for _ in range(3):
        engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    trace.zero_()
    ev.zero_()
    engine.run_nets(nets, x, cond)
    torch.cuda.synchronize()
    t = trace.cpu().numpy()
    e = ev.cpu().numpy()
    # wave -> (net, workgroup)
    recs = []      # (time_us, code, layer, unit, bits0, bitsl, polls, net, wg, wave)
    t0 = None
    for wv in range(nw):
        if t[wv, 5] <= 0 and e[wv, 0, 1] == 0:
            continue
        net, wg = int(t[wv, 16]), int(t[wv, 17])
        for k in range(64):
            tm, code, lay, uu = (int(v) for v in e[wv, k])
            if code == 0:
                break
            unit = uu & 0xffffffff
            if unit >= 1 << 31:
                unit -= 1 << 32
            recs.append([tm, code, lay, unit, (uu >> 32) & 0xff, (uu >> 40) & 0xff, (uu >> 48) & 0xffff, net, wg, wv % 8])
    recs.sort()
    t0 = recs[0][0]
    for r in recs:
        r[0] = (r[0] - t0) / 100.0
    print('rows %d, layers %d (+ tail), nets %d: %d events, launch span %.1f us' % (rows, L, G, len(recs), recs[-1][0]))
    units_per_wg = max(1, int(round(t[t[:, 5] > 0][:, 5].sum() / max(1, len({(r[7], r[8]) for r in recs})) / max(1, L - 1))))
    print('units per workgroup and layer ~ %d' % units_per_wg)

    # ---- 1. the timeline of one workgroup ----------------------------------------------------------------------------------
    print('\n== timeline of net 0, workgroup %d (us since the first event of the launch) ==' % show)
    for r in recs:
        if r[7] == 0 and r[8] == show:
            extra = ''
            if r[1] in (4, 12):
                extra = '  missing at first look: %s, last: %s, polls %d' % (bin(r[4]), bin(r[5]), r[6])
            print('%8.2f  wave %d  L%-2d u%-4d %s%s' % (r[0], r[9], r[2], r[3], NAMES.get(r[1], str(r[1])), extra))

    # ---- 2. links of the chain, over all workgroups -----------------------------------------------------------------------
    by_task = {}
    for r in recs:
        if r[1] in (1, 2, 4, 5, 6):
            by_task.setdefault((r[7], r[2], r[3]), {})[r[1]] = r
    # acknowledgement: event 3 follows event 2 of the same wave
    per_wave = {}
    for r in recs:
        per_wave.setdefault((r[7], r[8], r[9]), []).append(r)
    ack, issue = [], []
    acked_at = {}      # (net, layer, unit) -> time its stores were acknowledged
    for key, lst in per_wave.items():
        last2 = None
        for r in lst:
            if r[1] == 2:
                last2 = r
            elif r[1] in (3, 13) and last2 is not None:
                ack.append(r[0] - last2[0])
                acked_at[(key[0], last2[2], last2[3])] = r[0]
                last2 = None
            elif r[1] == 6 and last2 is not None:      # prefetched path: the drain at the top of the next unit is the acknowledgement
                acked_at[(key[0], last2[2], last2[3])] = r[0]
                last2 = None
    comp, st_issue, loads, prow, see, see_x, see_in = [], [], [], [], [], [], []
    for (net, lay, unit), evs in by_task.items():
        if 6 in evs and 1 in evs:
            comp.append(evs[1][0] - evs[6][0])
        if 1 in evs and 2 in evs:
            st_issue.append(evs[2][0] - evs[1][0])
        if 4 in evs and 5 in evs:
            loads.append(evs[5][0] - evs[4][0])
        if 5 in evs and 6 in evs:
            prow.append(evs[6][0] - evs[5][0])
        if 4 in evs and lay >= 1:
            d = dil[lay]
            prods = {unit, unit - ((d + 31) >> 5), (32 * unit + 31 - d) >> 5}
            tp = [acked_at.get((net, lay - 1, q)) for q in prods if q >= 0]
            tp = [v for v in tp if v is not None]
            if tp:
                dt = evs[4][0] - max(tp)
                see.append(dt)
                wgc = evs[4][8]
                lo = min(q for q in prods if q >= 0)
                (see_x if lo // units_per_wg != unit // units_per_wg else see_in).append(dt)

    def stat(name, v):
        v = np.array(v)
        if len(v):
            print('  %-74s n %5d  mean %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f us' % (name, len(v), v.mean(), np.percentile(v, 10), np.percentile(v, 50), np.percentile(v, 90)))
    print('\n== links (all workgroups) ==')
    stat('compute: top drained -> GEMM2 done', comp)
    stat('store issue: GEMM2 done -> stores issued', st_issue)
    stat('acknowledgement: stores issued -> vmcnt(0) (waiting path)', ack)
    stat('last producer acknowledged -> consumer sees its dependencies', see)
    stat('   ... look-back inside the workgroup', see_in)
    stat('   ... look-back from the left neighbour (progress word)', see_x)
    stat('row loads: deps ok -> rows arrived + split', loads)
    stat('P row: rows arrived -> top drained', prow)
    # layer period: median time of "computed" per layer
    per_layer = {}
    for r in recs:
        if r[1] == 1:
            per_layer.setdefault(r[2], []).append(r[0])
    ks = sorted(per_layer)
    med = [np.median(per_layer[k]) for k in ks]
    print('\nmedian "computed" time per layer: ' + ' '.join('L%d %.1f' % (k, m) for k, m in zip(ks, med)))
    if len(med) > 2:
        print('layer period (median of differences): %.2f us' % np.median(np.diff(med)))
    for code in (8, 9, 10, 11):
        v = [r[0] for r in recs if r[1] == code]
        if v:
            print('%-16s %.1f .. %.1f us (median %.1f)' % (NAMES[code], min(v), max(v), np.median(v)))


if __name__ == '__main__':
    main()
