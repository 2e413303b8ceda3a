#!/usr/bin/env python
"""Per-dispatch durations of the kernels whose name contains PATTERN, from a rocprofv3 --kernel-trace CSV, in dispatch order.
usage: kernel_durations.py kernel_trace.csv PATTERN [max_rows]"""
import csv
import sys

src, pat = sys.argv[1], sys.argv[2]
limit = int(sys.argv[3]) if len(sys.argv) > 3 else 64
rows = [r for r in csv.DictReader(open(src)) if pat in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print('%d dispatches of *%s*; durations (us) of the last %d in order:' % (len(d), pat, min(limit, len(d))))
print(' '.join('%.0f' % v for v in d[-limit:]))
