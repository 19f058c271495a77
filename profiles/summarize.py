#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/...) into the small files committed here.

    python profiles/summarize.py <round-tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> \
           [--variant fused_map --batch 16 --height 2160 --width 3840]

Writes
  profiles/<tag>_kernel_stats.csv   the `rocprofv3 --kernel-trace --stats` per-kernel summary
  profiles/<tag>_pmc.csv            per-kernel mean FETCH_SIZE / WRITE_SIZE (raw counter units)
  profiles/pmc_summary.json         bytes per launch of the dominant kernel, corrected as
                                    MI355X_MICROARCH.md prescribes: FETCH_SIZE reports half the
                                    bytes of wide coalesced reads on gfx950 (x2), WRITE_SIZE is
                                    exact; both are in KiB.  Note that FETCH_SIZE counts L2-side
                                    fabric requests, Infinity-Cache hits included.
"""
import argparse
import collections
import csv
import glob
import json
import os
import shutil

HERE = os.path.dirname(os.path.abspath(__file__))


def find(d, pattern):
    return glob.glob(os.path.join(d, pattern)) + glob.glob(os.path.join(d, '*', pattern))


def mean_counter(d, counter):
    """per kernel: mean over dispatches of the counter summed over its rows of one dispatch"""
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in find(d, '*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                per[r['Kernel_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    return {k: (sum(v.values()) / len(v), len(v)) for k, v in per.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('tag')
    ap.add_argument('stats_dir')
    ap.add_argument('fetch_dir')
    ap.add_argument('write_dir')
    ap.add_argument('--variant', default='fused_map')
    ap.add_argument('--batch', type=int, default=16)
    ap.add_argument('--height', type=int, default=2160)
    ap.add_argument('--width', type=int, default=3840)
    ap.add_argument('--csv-only', action='store_true',
                    help='write only <tag>_pmc.csv (e.g. a pass over the other_configs kernels)')
    ap.add_argument('--kernel', default='',
                    help='substring of the dominant kernel (default: the kernel with the largest fetch)')
    a = ap.parse_args()

    stats = find(a.stats_dir, '*kernel_stats.csv')
    if stats:
        shutil.copy(stats[0], os.path.join(HERE, '%s_kernel_stats.csv' % a.tag))
    # the same per LAUNCH SHAPE: one kernel name serves one-frame pipeline calls and 64-frame
    # launches alike, and a per-name average says nothing about either
    trace = find(a.stats_dir, '*kernel_trace.csv')
    if trace:
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(trace[0])):
            grid = 'x'.join(r[k] for k in ('Grid_Size_X', 'Grid_Size_Y', 'Grid_Size_Z'))
            per[(r['Kernel_Name'], grid, r['VGPR_Count'], r['LDS_Block_Size'])].append(
                (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
        with open(os.path.join(HERE, '%s_kernel_stats_by_grid.csv' % a.tag), 'w') as f:
            w = csv.writer(f)
            w.writerow(['kernel', 'grid (threads)', 'VGPRs', 'LDS bytes', 'launches', 'avg_us',
                        'median_us', 'min_us', 'max_us'])
            for (k, grid, vg, lds), v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
                v = sorted(v)
                w.writerow([k, grid, vg, lds, len(v), '%.1f' % (sum(v) / len(v)),
                            '%.1f' % v[len(v) // 2], '%.1f' % v[0], '%.1f' % v[-1]])
    fetch = mean_counter(a.fetch_dir, 'FETCH_SIZE')
    write = mean_counter(a.write_dir, 'WRITE_SIZE')
    with open(os.path.join(HERE, '%s_pmc.csv' % a.tag), 'w') as f:
        w = csv.writer(f)
        w.writerow(['kernel', 'launches', 'FETCH_SIZE_mean_KiB_raw', 'WRITE_SIZE_mean_KiB'])
        for k in sorted(set(fetch) | set(write)):
            w.writerow([k, fetch.get(k, (0, 0))[1], '%.1f' % fetch.get(k, (0, 0))[0],
                        '%.1f' % write.get(k, (0, 0))[0]])
    if a.csv_only:
        return
    # dominant kernel = largest fetch
    # (round 6: bench.py also times the dense loop beside the separable one - same traffic, fewer launches)
    cand = [k for k in fetch if a.kernel in k] if a.kernel else list(fetch)
    dom = max(cand, key=lambda k: (fetch[k][1] if a.kernel else 0, fetch[k][0]))
    traffic = (2 * fetch[dom][0] + write.get(dom, (0, 0))[0]) * 1024
    path = os.path.join(HERE, 'pmc_summary.json')
    summ = json.load(open(path)) if os.path.exists(path) else {}
    summ[a.variant] = {'kernel': dom, 'batch': a.batch, 'height': a.height, 'width': a.width,
                       'fetch_size_kib_raw': fetch[dom][0], 'write_size_kib': write[dom][0],
                       'traffic_bytes_per_launch': int(traffic), 'round': a.tag,
                       'correction': 'FETCH_SIZE x2 (gfx950 wide coalesced reads) + WRITE_SIZE, KiB'}
    json.dump(summ, open(path, 'w'), indent=1, sort_keys=True)
    print(json.dumps(summ[a.variant], indent=1))


if __name__ == '__main__':
    main()
