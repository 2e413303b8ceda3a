#!/usr/bin/env python
"""Round-4 profile summaries from the files tools/profile_round4.sh leaves behind.  Everything is derived from the bench line of
the SAME configuration (what one forward launches of the dominant kernel: `roofline.launches_per_forward`,
`alg_bytes_per_forward` -- layer 0 folded onto its scalars counted at its 260 B per sample) and the raw counter rows; nothing
is inferred from instruction counts and nothing is edited by hand.

  <tag>_<cfg>_hbm_traffic.json   per configuration: raw FETCH_SIZE / WRITE_SIZE totals over the dominant kernel's launches of the PMC
                                 runs, the forwards those launches make up, traffic and algorithmic bytes per launch, their
                                 ratio, the rocprofv3 --stats average launch duration and the roofline fraction it implies
                                 (bench.py reads this file for `roofline.traffic` / `frac_rocprof` of the matching --case)
  stdout                         the per-configuration table (<tag>_configs.md)
usage: profile_round4_summarize.py OUTDIR TMPDIR TAG"""
import csv
import json
import os
import sys

out, tmp, tag = sys.argv[1], sys.argv[2], sys.argv[3]
PEAK = 8000.0e9


def sums(path, counter, pat):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and pat in r['Kernel_Name']:
            tot += float(r['Counter_Value'])
            n += 1
    return tot, n


def bench(name):
    path = os.path.join(out, '%s_%s_bench.json' % (tag, name))
    if os.path.exists(path):
        for l in open(path):
            if l.startswith('{'):
                return json.loads(l)
    return None


rows_md = []
for name, label in (('c3', 'C3 default model 1 x 160000 (headline)'), ('c4', 'C4 share 8 x 64000'), ('c5', 'C5 transposed-conv 1 x 960000, f16x3'),
                    ('c5_f16', 'C5, fp16 storage mode (reduced precision)'),
                    ('skip', 'default model with `use_skip_connection: True` (modules.py:147), 1 x 160000 (round 6)')):
    b = bench(name)
    if not b:
        continue
    roof = b['roofline']
    rows = b['config']['utterances_per_gpu'] * b['config']['samples_per_utterance']
    persist = 'stack_persist' in roof['kernel']
    if persist:
        pat = 'stack_persist_kernel'
    elif 'layer_f16x3' in roof['kernel']:
        pat = ('layer_f16x3_kernel<false, true, false, false' if name.startswith('c5') else
               'layer_f16x3_kernel<true, false, false, false' if name == 'skip' else 'layer_f16x3_kernel<false, false, false, false')
    else:
        pat = 'layer_h16_kernel<true, false' if name.startswith('c5') else 'layer_h16_kernel<false, false'
    f, nf = sums(os.path.join(tmp, 'p4_%s_fetch.csv' % name), 'FETCH_SIZE', pat)
    w, nw = sums(os.path.join(tmp, 'p4_%s_write.csv' % name), 'WRITE_SIZE', pat)
    sq = os.path.join(tmp, 'p4_%s_sq.csv' % name)
    busy, _ = sums(sq, 'SQ_VALU_MFMA_BUSY_CYCLES', pat)
    act, _ = sums(sq, 'GRBM_GUI_ACTIVE', pat)
    valu, _ = sums(sq, 'SQ_INSTS_VALU', pat)
    mfma, _ = sums(sq, 'SQ_INSTS_MFMA', pat)
    assert nf == nw and nf > 0, (name, nf, nw)
    traffic = (2 * f + w) * 1024.0
    if persist:
        fwd = nf / float(roof['launches_per_forward'])
        alg = fwd * roof['alg_bytes_per_forward']
        units = fwd * roof['net_layers_per_forward'] * ((rows + 31) // 32)
        conc = 1.0
    else:
        fwd = None
        alg = nf * roof['alg_bytes_per_launch']      # (per-layer launches: the bench line's figure is per launch; the PMC runs are the same launches)
        units = nf * ((rows + 31) // 32)      # one net per launch on its own stream
        conc = float(roof.get('concurrent_launches', 1.0))      # launches of this kernel sharing the chip (two chains on two streams), measured by bench.py
    tj = {'command': 'rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE -- python bench.py --no-cpu-baseline --no-f32-exact --no-graph --steps 2 --warmup 1 '
                     + ' '.join(['--case', b['config']['case']] + (['--precision', b['precision']] if b['precision'] != 'f16x3' else [])),
          'workload': b['config']['workload'], 'rows': rows, 'kernel_pattern': pat,
          'FETCH_SIZE_KB_total': f, 'WRITE_SIZE_KB_total': w, 'launches': nf, 'forwards': fwd,
          'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); write bytes = WRITE_SIZE x 1024',
          'traffic_bytes_total': traffic, 'algorithmic_bytes_total': alg,
          'traffic_bytes_per_launch': traffic / nf, 'algorithmic_bytes_per_launch': alg / nf, 'ratio': traffic / alg}
    st = os.path.join(tmp, 'p4_%s_stats.csv' % name)
    if os.path.exists(st):
        for r in csv.DictReader(open(st)):
            if pat in r['Name']:
                tj['rocprof_kernel_us'], tj['rocprof_kernel_calls'] = float(r['AverageNs']) / 1e3, int(r['Calls'])
                tj['concurrent_launches'] = conc
                tj['frac_rocprof'] = conc * tj['algorithmic_bytes_per_launch'] / (tj['rocprof_kernel_us'] * 1e-6) / PEAK
                tj['frac_rocprof_note'] = ('concurrent_launches x algorithmic bytes per launch / rocprofv3 --stats average launch duration / 8 TB/s '
                                           '(concurrent_launches: the bench line\'s measured overlap of the two chains\' launches; 1 for the persistent launch)')
                break
    json.dump(tj, open(os.path.join(out, '%s_%s_hbm_traffic.json' % (tag, name)), 'w'), indent=1)
    simds = 1024.0 if persist else 512.0
    rows_md.append((label, b['value'] / 1e6, b['ms_per_step'], b['model']['hbm_frac_of_8TBs'], roof['kernel'].split(' (')[0], roof['frac'],
                    tj.get('frac_rocprof', float('nan')), traffic / alg, busy / (act / 8.0 * simds) if act else float('nan'),
                    valu / units, mfma / units))
print('# per-configuration table (tools/profile_round4.sh / profile_round6.sh, %s): bench line + rocprofv3 stats + PMC passes of the same build on one box\n' % tag)
print('| config (one MI355X) | M samples/s | ms/step | whole model, frac of 8 TB/s | dominant kernel | its frac of 8 TB/s, HIP events (bench `roofline.frac`) | same from the rocprofv3 --stats average | HBM traffic / algorithmic bytes | matrix pipe busy | VALU / MFMA instructions per 32-row unit |')
print('|---|---|---|---|---|---|---|---|---|---|')
for r in rows_md:
    print('| %s | %.1f | %.3f | %.3f | `%s` | %.3f | %.3f | %.2f | %.0f %% | %.0f / %.0f |' % (r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], 100 * r[8], r[9], r[10]))
for name, label in (('in', "default model with `normalize_wavenet: 'in'` (modules.py:263-284), 1 x 16000: the composed path (round 6)"),
                    ('16k', 'default model 1 x 16000, f16x3'), ('16k_f32', 'default model 1 x 16000, exact fp32 (the range guard\'s rerun path)'),
                    ('c3_f32', 'C3, exact fp32'), ('c1', 'C1 one flow, 1 x 16000'), ('c2', 'C2 shared nets, 1 x 160000')):
    b = bench(name)
    if b:
        print('\n%s: %.2f M samples/s, %.4f ms/step, whole model %.3f of 8 TB/s.' % (label, b['value'] / 1e6, b['ms_per_step'], b['model']['hbm_frac_of_8TBs']))
pl = bench('c3_perlayer')
if pl:
    print('\nC3 with the per-layer launches (PWV_PERSIST=0), same box: %.1f M samples/s, %.3f ms/step, whole model %.3f of 8 TB/s.'
          % (pl['value'] / 1e6, pl['ms_per_step'], pl['model']['hbm_frac_of_8TBs']))
print('\nTraffic = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 B summed over the dominant kernel\'s launches of an eager 2-step run (gfx950 counts 128-byte read '
      'requests at 64 B: MI355X_MICROARCH.md, HBM).  Algorithmic bytes of the same launches = forwards x `roofline.alg_bytes_per_forward` of the bench line '
      '(512 B per sample, net and layer; 260 B for a net\'s layer 0 folded onto its scalars; + the per-sample condition in transposed-conv mode).  Matrix pipe busy = '
      'SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x SIMDs the launch occupies) -- GRBM_GUI_ACTIVE is summed over the 8 XCDs; instructions per unit = '
      'SQ_INSTS_VALU / SQ_INSTS_MFMA over the units those launches ran (a persistent launch averages its folded layer 0: 36 MFMAs, with the others: 120).  The per-configuration '
      'raw totals are in %s_<cfg>_hbm_traffic.json (what bench.py reads for `roofline.traffic` / `frac_rocprof`).' % tag)
