#!/usr/bin/env python
"""Kernel timeline of ONE bench step from a rocprofv3 kernel trace: per kernel start offset, duration and the gap to the
previous kernel on the same queue, plus copies.
usage (GPU box):  cd /tmp && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d OUT -- python bench.py --steps 3 --warmup 1 ...
                  python tools/step_timeline.py OUT [n_last_kernels]"""
import csv
import glob
import sys

out = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows = []
for f in glob.glob(out + '/**/*kernel_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'][:70], r.get('Queue_Id', '?'), 'K'))
for f in glob.glob(out + '/**/*memory_copy_trace.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', ''), 'copy', 'C'))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
last_end = {}
for s, e, name, q, kind in rows:
    gap = (s - last_end[q]) / 1e3 if q in last_end else 0.0
    last_end[q] = e
    print('%9.1f us  dur %7.1f  gap(q) %6.1f  q=%-6s %s' % ((s - t0) / 1e3, (e - s) / 1e3, gap, q, name))
