"""Same-process A/B at the reference's default training shape (256 clips x 8 frames): the replayed step with module-level switches
of stove_amd.ops flipped.  Usage: python tools/t8_ab.py NAME=a,b [NAME=a,b ...]   e.g.  SMALL_M_TILE=3,2"""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv, switches = [sys.argv[0]], sys.argv[1:]
import torch  # noqa: E402

import bench  # noqa: E402
from stove_amd import graphed, ops  # noqa: E402

dev = torch.device('cuda:0')
T = int(os.environ.get('T8_FRAMES', '8'))
WL = os.environ.get('T8_WORKLOAD', 'billiards')
data = bench.make_batch(WL, 256, T, 0)
names = [s.split('=')[0] for s in switches]
values = [[int(v) for v in s.split('=')[1].split(',')] for s in switches]
for rep in range(int(os.environ.get('T8_REPS', '3'))):
    for combo in itertools.product(*values):
        for n, v in zip(names, combo):
            if n == 'OVERLAP':
                from stove_amd import settings
                settings.set_overlap(bool(v))
            elif hasattr(graphed, n):
                setattr(graphed, n, v)
            else:
                setattr(ops, n, v)
        job = bench.Job(WL, dev, data, 'bf16x3', 'f32', 1)
        job.step(0)
        ms, ms_max, _ = job.median_ms(60)
        print(dict(zip(names, combo)), 'ms/step %.4f (max %.4f)' % (ms, ms_max), flush=True)
        del job
