#!/usr/bin/env python
"""Per-configuration table for profiles/r03_*_configs.md from the files tools/profile_round3.sh leaves behind:
bench line (value, roofline) + PMC traffic (2 x FETCH_SIZE + WRITE_SIZE, the gfx950 correction of MI355X_MICROARCH.md) +
matrix-pipe busy share of the dominant kernel.  usage: profile_round3_summarize.py OUTDIR TMPDIR"""
import collections
import csv
import re
import json
import os
import sys

out, tmp = sys.argv[1], sys.argv[2]


def sums(path, counter, pat):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and pat in r['Kernel_Name']:
            tot += float(r['Counter_Value'])
            n += 1
    return tot, n


def bench(name):
    for l in open(os.path.join(out, name + '_bench.json')):
        if l.startswith('{'):
            return json.loads(l)
    return None


rows = []
for name, label in (('c3', 'C3 default model 1 x 160000 (headline)'), ('c4', 'C4 share 8 x 64000'), ('c5', 'C5 transposed-conv 1 x 960000, f16x3'),
                    ('c5f16', 'C5, fp16 storage mode (reduced precision)')):
    b = bench(name)
    if not b:
        continue
    roof = b['roofline']
    pat = 'stack_persist' if 'persist' in roof['kernel'] else roof['kernel'].split(' ')[0].split('<')[0]
    if pat == 'layer_f16x3_kernel' and name.startswith('c5'):
        pat = 'layer_f16x3_kernel<false, true, false, false'      # the per-sample-condition residual variant (not FIRST)
    if pat == 'layer_h16_kernel' and name.startswith('c5'):
        pat = 'layer_h16_kernel<true, false'
    f, nf = sums(os.path.join(tmp, 'p3_%s_fetch.csv' % name), 'FETCH_SIZE', pat)
    w, nw = sums(os.path.join(tmp, 'p3_%s_write.csv' % name), 'WRITE_SIZE', pat)
    busy, nb = sums(os.path.join(tmp, 'p3_%s_sq.csv' % name), 'SQ_VALU_MFMA_BUSY_CYCLES', pat)
    act, _ = sums(os.path.join(tmp, 'p3_%s_sq.csv' % name), 'GRBM_GUI_ACTIVE', pat)
    valu, _ = sums(os.path.join(tmp, 'p3_%s_sq.csv' % name), 'SQ_INSTS_VALU', pat)
    mfma, _ = sums(os.path.join(tmp, 'p3_%s_sq.csv' % name), 'SQ_INSTS_MFMA', pat)
    traffic = (2 * f + w) * 1024.0                    # bytes over the launches of the two timed eager steps (+1 warm-up)
    # algorithmic bytes of the same launches: the bench line's per-net-layer (or per-launch) figure x what those launches ran
    if 'alg_bytes_per_net_layer' in roof:
        # persistent launches: every MFMA count / 120 = one 32-row unit of one net-layer -- except the units of a net's layer 0
        # when it runs folded onto its four input scalars (36 MFMAs): every launch of the bench configurations starts with one
        units = mfma / 120.0
        if str(roof.get('first_layer', '')).startswith('folded'):
            nets = int(re.search(r'x (\d+) nets per launch', roof['kernel']).group(1))
            first_units = nf * nets * (b['config']['utterances_per_gpu'] * b['config']['samples_per_utterance'] + 31) // 32
            units = (mfma + 84.0 * first_units) / 120.0
        alg = units * 32 * (roof['alg_bytes_per_net_layer'] / (b['config']['utterances_per_gpu'] * b['config']['samples_per_utterance']))
    else:
        alg = roof['alg_bytes_per_launch'] * nf
    if name == 'c3' and 'alg_bytes_per_net_layer' in roof:
        rows_per_launch = b['config']['utterances_per_gpu'] * b['config']['samples_per_utterance']
        net_layers = units * 32 / rows_per_launch
        tj = dict({'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python bench.py --no-cpu-baseline --no-f32-exact --no-graph --steps 2 --warmup 1',
                   'workload': 'bench/c3, 1 x 160000 samples', 'kernel': 'pwv::stack_persist_kernel<false>',
                   'FETCH_SIZE_KB_total': f, 'WRITE_SIZE_KB_total': w, 'launches': nf, 'net_layers_in_those_launches': net_layers,
                   'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); write bytes = WRITE_SIZE x 1024',
                   'traffic_bytes_per_net_layer': traffic / net_layers, 'algorithmic_bytes_per_net_layer': alg / net_layers, 'ratio': traffic / alg})
        st = os.path.join(tmp, 'p3_c3_stats.csv')
        if os.path.exists(st):      # rocprofv3 --stats of the same build: the kernel's average launch duration, for bench.py's roofline line
            for r in csv.DictReader(open(st)):
                if 'stack_persist' in r['Name']:
                    tj['rocprof_kernel_us'], tj['rocprof_kernel_calls'] = float(r['AverageNs']) / 1e3, int(r['Calls'])
        json.dump(tj, open(os.path.join(out, 'hbm_traffic.json'), 'w'), indent=1)
    simds = 1024.0 if 'persist' in roof['kernel'] else 512.0 * (2 if 'G = 2' in roof.get('note', '') else 1)
    rows.append((label, b['value'] / 1e6, b['ms_per_step'], b['model']['hbm_frac_of_8TBs'], roof['kernel'].split(' (')[0], roof['frac'],
                 traffic / alg if alg else float('nan'), busy / (act / 8.0 * simds) if act else float('nan'), valu / mfma * 120 if mfma else float('nan')))
print('# per-configuration table (tools/profile_round3.sh): bench line + PMC passes of the same build\n')
print('| config (one MI355X) | M samples/s | ms/step | whole model, frac of 8 TB/s | dominant kernel | its frac of 8 TB/s (bench `roofline`) | HBM traffic / algorithmic bytes | matrix pipe busy | VALU per 120 MFMA |')
print('|---|---|---|---|---|---|---|---|---|')
for r in rows:
    print('| %s | %.1f | %.3f | %.3f | `%s` | %.3f | %.2f | %.0f %% | %.0f |' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], 100 * r[7], r[8]))
pl = bench('c3_perlayer')
if pl:
    print('\nC3 with the per-layer launches (PWV_PERSIST=0), same box: %.1f M samples/s, %.3f ms/step, whole model %.3f of 8 TB/s.'
          % (pl['value'] / 1e6, pl['ms_per_step'], pl['model']['hbm_frac_of_8TBs']))
print('\nTraffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B summed over the dominant kernel\'s launches of an eager 2-step run (gfx950 counts 128-byte read '
      'requests at 64 B: MI355X_MICROARCH.md, HBM); algorithmic bytes = the `roofline` figure of the bench line for the same launches; matrix pipe busy = '
      'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x SIMDs the launch occupies) -- GRBM_GUI_ACTIVE is summed over the 8 XCDs.')
