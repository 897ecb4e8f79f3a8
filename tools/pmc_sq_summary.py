"""Summarise a rocprofv3 --pmc pass of SQ / GRBM counters into profiles/<tag>_pmc_sq.json (average per launch).

Usage: python tools/pmc_sq_summary.py <counter_collection.csv> <out.json>
"""
import collections
import csv
import json
import sys


def main():
    src, out = sys.argv[1:3]
    agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(src)):
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].split('::')[-1]
        if not name or name.startswith('Cijk') or 'at::native' in r['Kernel_Name']:
            continue
        a = agg[name][r['Counter_Name']]
        a[0] += float(r['Counter_Value'])
        a[1] += 1
    res = {}
    for k, cs in agg.items():
        res[k] = {c: v[0] / max(v[1], 1) for c, v in cs.items()}
        res[k]['launches'] = max(v[1] for v in cs.values())
    json.dump(res, open(out, 'w'), indent=1, sort_keys=True)
    print('wrote', out, len(res), 'kernels')


if __name__ == '__main__':
    main()
