#!/usr/bin/env python
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel (means), short kernel names."""
import collections
import csv
import sys

for f in sys.argv[1:]:
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'pwv::' not in k:
            continue
        k = k.split('(')[0].replace('void ', '')[:70]
        agg[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, v in agg.items():
        print(k)
        for c, vals in sorted(v.items()):
            print('   %-30s n=%-3d mean=%.5g' % (c, len(vals), sum(vals) / len(vals)))
