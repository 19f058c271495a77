"""per kernel and counter: how a dimensioned counter (one row per TCC instance x XCC) spreads over its
instances - min / mean / max over the rows of a dispatch, averaged over the dispatches, and max / mean
(1.0 = every channel carries the same load)"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
rows = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(list)))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        rows[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']].append(float(r['Counter_Value']))
for k in sorted(rows):
    if 'rocclr' in k:
        continue
    print(k if len(k) < 100 else k[:97] + '...')
    for c, disp in sorted(rows[k].items()):
        n = [len(v) for v in disp.values()]
        mn = sum(min(v) for v in disp.values()) / len(disp)
        mx = sum(max(v) for v in disp.values()) / len(disp)
        me = sum(sum(v) / len(v) for v in disp.values()) / len(disp)
        print('   %-36s rows %3d  min %14.0f  mean %14.0f  max %14.0f  max/mean %.3f  (%d dispatches)'
              % (c, n[0], mn, me, mx, mx / me if me else 0.0, len(disp)))
