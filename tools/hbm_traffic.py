#!/usr/bin/env python
"""HBM traffic per launch of the dominant kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected
as MI355X_MICROARCH.md prescribes for gfx950 (FETCH_SIZE counts 128-byte requests as 64: read bytes = 2 x FETCH_SIZE KB;
WRITE_SIZE is 1:1).  Usage: hbm_traffic.py fetch_counter_collection.csv write_counter_collection.csv kernel_substring
algorithmic_bytes_per_launch out.json "command line" "workload" [kernel_stats.csv]."""
import collections
import csv
import json
import sys


def means(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] == counter and 'pwv::' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ', '')].append(float(r['Counter_Value']))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main():
    fetch_csv, write_csv, kernel, alg, out, cmd, workload = sys.argv[1:8]
    stats_csv = sys.argv[8] if len(sys.argv) > 8 else None
    f, w = means(fetch_csv, 'FETCH_SIZE'), means(write_csv, 'WRITE_SIZE')
    name = [k for k in f if kernel in k][0]
    fetch_kb, write_kb = f[name][0], w[name][0]
    traffic = 2 * fetch_kb * 1024 + write_kb * 1024
    res = {
        'command': cmd, 'workload': workload, 'kernel': name,
        'FETCH_SIZE_KB': fetch_kb, 'WRITE_SIZE_KB': write_kb,
        'correction': 'read bytes = 2 x FETCH_SIZE x 1024 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM section); '
                      'write bytes = WRITE_SIZE x 1024',
        'traffic_bytes_per_launch': traffic, 'algorithmic_bytes_per_launch': float(alg), 'ratio': traffic / float(alg),
        'all_kernels': {k: {'FETCH_SIZE_KB_mean': f[k][0], 'launches_FETCH_SIZE': f[k][1],
                            'WRITE_SIZE_KB_mean': w.get(k, (None, 0))[0], 'launches_WRITE_SIZE': w.get(k, (None, 0))[1]} for k in f},
    }
    if stats_csv:          # rocprofv3 --stats: the kernel's average duration, for bench.py's roofline line
        for r in csv.DictReader(open(stats_csv)):
            if kernel in r['Name']:
                res['rocprof_kernel_us'] = float(r['AverageNs']) / 1e3
                res['rocprof_kernel_calls'] = int(r['Calls'])
                break
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps({k: res[k] for k in ('kernel', 'FETCH_SIZE_KB', 'WRITE_SIZE_KB', 'traffic_bytes_per_launch', 'ratio')}))


if __name__ == '__main__':
    main()
