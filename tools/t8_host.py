"""Is the replayed step at the reference's default training shape host-bound?  Host time to ENQUEUE n steps (no synchronisation)
against the time until the device has run them.  Usage: python tools/t8_host.py [frames]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
sys.argv = [sys.argv[0]]
import torch  # noqa: E402

import bench  # noqa: E402

dev = torch.device('cuda:0')
data = bench.make_batch('billiards', 256, T, 0)
job = bench.Job('billiards', dev, data, 'bf16x3', 'f32', 1)
job.step(0)
for i in range(20):
    job.step(i)
torch.cuda.synchronize()
for n in (100, 400):
    t0 = time.perf_counter()
    for i in range(n):
        job.step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print('T=%d  %d steps: host enqueue %.3f ms/step, until the device is done %.3f ms/step' % (T, n, (t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))
ev = [torch.cuda.Event(enable_timing=True) for _ in range(201)]
ev[0].record()
for i in range(200):
    job.step(i)
    ev[i + 1].record()
torch.cuda.synchronize()
ts = [ev[i].elapsed_time(ev[i + 1]) for i in range(200)]
s = sorted(ts)
print('event intervals: mean %.3f  p10 %.3f  p50 %.3f  p90 %.3f  p99 %.3f  max %.3f' % (sum(ts) / 200, s[20], s[100], s[180], s[198], s[-1]))
print('first 48:', ' '.join('%.2f' % t for t in ts[:48]))
