"""Measurement aid: ATen operators (with input shapes) ranked by GPU time in one bench step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from stove_amd.arena import ParamArena
from stove_amd.optim import FlatAdam
from stove_amd.video_prediction.stove import Stove
dev = torch.device('cuda:0')
cfg = bench.build_config('billiards', dev)
torch.manual_seed(0)
model = Stove(cfg).to(dev)
bucket = ParamArena(model, 1)
opt = FlatAdam(bucket, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)
data = bench.make_batch('billiards', 256, 100, 0)
x = torch.from_numpy(data['X']).to(dev)
def step(i):
    bucket.zero()
    elbo, _, _ = model(x, i + 1, None)
    (-elbo).backward()
    opt.step(max_norm=1.0)
for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', None)
    if t is None:
        t = e.self_cuda_time_total
    if t > 1 and (e.key.startswith('aten::') or 'Backward' in e.key):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
for t, c, k, s in rows[:25]:
    print('%8.1f us %3d x  %-28s %s' % (t, c, k, s))
