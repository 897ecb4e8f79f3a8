"""Timeline of the MEDIAN training step from a rocprofv3 --kernel-trace CSV (a step ends with the optimiser's adam_tick_k launch):
start offset, duration, queue of every kernel of that step, the device's idle gaps, and p50 / p99 / max duration of every
kernel over all traced steps (first step excluded: it carries one-time initialisation).
Usage: python tools/timeline.py gpurun_out/rNN/ks/..._kernel_trace.csv [min_us]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i + 1 for i, r in enumerate(rows) if 'adam_tick_k' in r['Kernel_Name'] and i + 1 < len(rows)]          # first kernel after each step's end


def short(name):
    return name.replace('void ', '').replace('stove::', '').split('(')[0][:60]


steps = []
for a, b in zip(marks[1:-1], marks[2:]):            # complete steps, without the first
    steps.append((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp']), a, b))
durs = sorted(s[0] for s in steps)
print('%d complete steps: min %.3f  p50 %.3f  max %.3f ms' % (len(steps), durs[0] / 1e6, durs[len(durs) // 2] / 1e6, durs[-1] / 1e6))
med = sorted(steps)[len(steps) // 2]
_, a, b = med
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print('median step: %d kernels, %.3f ms' % (len(step), med[0] / 1e6))
busy_end = t0
idle = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    if (e - s) / 1e3 >= min_us or gap > 10:
        print('%8.1f  +%7.1f us  q%-2s %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], short(r['Kernel_Name']), '   [gap %.1f]' % gap if gap > 5 else ''))
    busy_end = max(busy_end, e)
print('device idle inside the step: %.1f us' % idle)
per = defaultdict(list)
for _, a, b in steps:
    for r in rows[a:b]:
        per[short(r['Kernel_Name'])].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print('\nper-kernel durations over %d steps (us): launches/step, p50, p99, max' % len(steps))
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    v.sort()
    if sum(v) / len(steps) < 10:
        continue
    print('%-62s %5.1f x  p50 %8.1f  p99 %8.1f  max %8.1f' % (k, len(v) / len(steps), v[len(v) // 2], v[min(len(v) - 1, int(0.99 * len(v)))], v[-1]))
