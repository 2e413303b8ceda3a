#!/usr/bin/env python
"""Turn a rocprofv3 --kernel-trace --stats *_kernel_stats.csv into a compact table
(kernel names truncated) for profiles/.  Usage: summarize_rocprof.py in.csv out.md [title]"""
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else src
    rows = list(csv.DictReader(open(src)))
    with open(dst, 'w') as f:
        f.write('# %s\n\n' % title)
        f.write('| kernel | calls | total ms | avg us | % | min us | max us |\n|---|---|---|---|---|---|---|\n')
        for r in rows:
            name = r['Name'].replace('|', '/')
            if len(name) > 110:
                name = name[:107] + '...'
            f.write('| `%s` | %s | %.3f | %.2f | %.2f | %.2f | %.2f |\n' % (
                name, r['Calls'], float(r['TotalDurationNs']) / 1e6, float(r['AverageNs']) / 1e3,
                float(r['Percentage']), float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))


if __name__ == '__main__':
    main()
