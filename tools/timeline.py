"""Timeline of one training step from a rocprofv3 --kernel-trace CSV: start offset, duration, queue of every kernel of the
last complete step (steps are delimited by bw_transform_k launches) and the idle gaps of the device.
Usage: python tools/timeline.py gpurun_out/r01/ks/ks_kernel_trace.csv [min_us]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 15.0
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marks = [i for i, r in enumerate(rows) if 'bw_transform_k' in r['Kernel_Name']]
a, b = marks[-2], marks[-1]
step = rows[a:b]
t0 = int(step[0]['Start_Timestamp'])
print('step: %d kernels, %.3f ms' % (len(step), (int(rows[b]['Start_Timestamp']) - t0) / 1e6))
busy_end = t0
idle = 0.0
for r in step:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - busy_end) / 1e3
    if gap > 0:
        idle += gap
    name = r['Kernel_Name'].replace('void ', '').replace('stove::', '')
    name = name.split('(')[0][:60]
    if (e - s) / 1e3 >= min_us or gap > 10:
        print('%8.1f  +%7.1f us  q%-2s %s%s' % ((s - t0) / 1e3, (e - s) / 1e3, r['Queue_Id'], name, '   [gap %.1f]' % gap if gap > 5 else ''))
    busy_end = max(busy_end, e)
print('device idle inside the step: %.1f us' % idle)
