"""Aggregates rocprofv3 --pmc counter_collection CSVs: mean counter value per (kernel, counter).

    python tools/pmc_summary.py <dir-with-*counter_collection.csv> [kernel-substring ...]
"""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    subs = sys.argv[2:]
    agg = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if subs and not any(s in k for s in subs):
                continue
            agg[(k[:60], r.get('Grid_Size', '?'), r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, grid, c), v in sorted(agg.items()):
        print(f'{k:60s} grid {grid:>8s} {c:28s} n={len(v):4d} mean={sum(v) / len(v):16.1f}')


if __name__ == '__main__':
    main()
