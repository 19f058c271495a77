"""mean of every collected counter per kernel (summed over the rows of a dispatch)"""
import collections
import csv
import glob
import os
import sys

d = sys.argv[1]
per = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
    for r in csv.DictReader(open(f)):
        per[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
for k, cs in per.items():
    name = k if len(k) < 90 else k[:87] + '...'
    print(name)
    for c, v in sorted(cs.items()):
        print('   %-28s %16.0f  (mean of %d dispatches)' % (c, sum(v.values()) / len(v), len(v)))
