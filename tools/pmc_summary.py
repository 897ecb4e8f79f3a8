"""Summarise rocprofv3 --pmc passes into profiles/<tag>_pmc_traffic.json.

Usage: python tools/pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
FETCH_SIZE / WRITE_SIZE are reported in KiB by rocprofv3; on gfx950 FETCH_SIZE counts 128-B
requests as 64 B, so the read side is doubled (MI355X_MICROARCH.md, section HBM) -- WRITE_SIZE is
used as is.  Values are averaged per launch of each kernel.
"""
import collections
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def per_kernel(path, counter):
    agg = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        name = r['Kernel_Name'].split('(')[0].replace('void ', '').split('<')[0].split('::')[-1]
        agg[name][0] += float(r['Counter_Value'])
        agg[name][1] += 1
    return agg


def main():
    fetch, write, out = sys.argv[1:4]
    f = per_kernel(fetch, 'FETCH_SIZE')
    w = per_kernel(write, 'WRITE_SIZE')
    res = {}
    for k in sorted(set(f) | set(w)):
        fb = 2.0 * 1024.0 * f[k][0] / max(f[k][1], 1) if k in f else 0.0
        wb = 1024.0 * w[k][0] / max(w[k][1], 1) if k in w else 0.0
        res[k] = {'launches_profiled': int(max(f[k][1] if k in f else 0, w[k][1] if k in w else 0)),
                  'fetch_bytes_per_launch': fb, 'write_bytes_per_launch': wb, 'hbm_bytes_per_launch': fb + wb}
    meta = {'_note': 'FETCH_SIZE x2 (gfx950 correction) + WRITE_SIZE, KiB -> bytes, average per launch; '
                     'separate rocprofv3 --pmc passes of `bench.py --steps 2 --warmup 1`'}
    # which sources the profiled library was built from: bench.py quotes these figures only for the same sources
    from stove_amd import build
    meta['_source_hash'] = build.source_hash()
    # steps the passes held: the optimiser's kernel runs once per step (bench.py --steps 2 --warmup 1 plus the step that precedes the
    # warm-up: four)
    meta['_steps_profiled'] = res['flat_adam_k']['launches_profiled'] if 'flat_adam_k' in res else 3
    meta.update(res)
    json.dump(meta, open(out, 'w'), indent=1)
    print('wrote', out, len(res), 'kernels')


if __name__ == '__main__':
    main()
