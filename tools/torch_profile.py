"""Measurement aid: per-kernel GPU time of the bench step via torch.profiler (cheap alternative to a rocprofv3 run)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from stove_amd.arena import ParamArena
from stove_amd.video_prediction.stove import Stove

dev = torch.device('cuda:0')
workload = sys.argv[1] if len(sys.argv) > 1 else 'billiards'
cfg = bench.build_config(workload, dev)
torch.manual_seed(0)
model = Stove(cfg).to(dev)
bucket = ParamArena(model, 1)
from stove_amd.optim import FlatAdam
opt = FlatAdam(bucket, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)
data = bench.make_batch(workload, 256, 100, 0)
x = torch.from_numpy(data['X']).to(dev).contiguous()
actions = torch.from_numpy(data['action']).float().to(dev) if 'action' in data else None

def step(i):
    bucket.zero()
    elbo, _, _ = model(x, i + 1, actions)
    (-elbo).backward()
    opt.step(max_norm=1.0)

for i in range(3):
    step(i)
torch.cuda.synchronize()
S = 5
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for i in range(S):
        step(3 + i)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages():
    t = getattr(e, 'device_time_total', None)
    if t is None:
        t = e.cuda_time_total
    if t > 0:
        rows.append((t / S, e.count / S, e.key))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print('total GPU kernel time %.1f us/step, %d launches/step' % (tot, sum(r[1] for r in rows)))
for t, c, k in rows[:70]:
    print('%8.1f us %6.1f x  %s' % (t, c, k.replace('void at::native::', '').replace('(anonymous namespace)::', '')[:130]))
