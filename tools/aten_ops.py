"""Which ATen ops still launch kernels inside the bench step (name, shapes, count per step)."""
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
x = torch.from_numpy(bench.make_batch('billiards', 256, 100, 0)['X']).to(dev).contiguous()
m1 = torch.tensor(-1.0, device=dev)


def step(i):
    bucket.zero()
    elbo, _, _ = model(x, i + 1, None)
    elbo.backward(m1)
    opt.step(max_norm=1.0)


for i in range(3):
    step(i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    step(3)
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    t = getattr(e, 'self_device_time_total', 0)
    if t > 0 and e.key.startswith('aten::'):
        rows.append((t, e.count, e.key, str(e.input_shapes)[:110]))
for t, c, k, sh in sorted(rows, reverse=True):
    print('%7.1f us %3d x  %-28s %s' % (t, c, k, sh))
