"""Which ATen calls (fills, copies, adds, the noise draw ...) a replayed step still contains, and who issues them: one step of the
graphed form run eagerly under torch.profiler with stacks; prints every non-stove kernel with the Python frame that launched it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
from stove_amd.arena import ParamArena
from stove_amd.graphed import GraphedTrainStep
from stove_amd.optim import FlatAdam
from stove_amd.utils.utils import bw_transform
from stove_amd.video_prediction.stove import Stove

dev = torch.device('cuda:0')
workload = sys.argv[1] if len(sys.argv) > 1 else 'billiards'
cfg = bench.build_config(workload, dev)
torch.manual_seed(0)
model = Stove(cfg).to(dev)
bucket = ParamArena(model, 1)
opt = FlatAdam(bucket, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)
data = bench.make_batch(workload, 64, 100, 0)
x = torch.from_numpy(data['X']).to(dev).contiguous()
actions = torch.from_numpy(data['action']).float().to(dev) if 'action' in data else None
if actions is None:
    x = bw_transform(x)
    cfg.input_bw_plane = True
step = GraphedTrainStep(model, bucket, opt, 1.0)
for i in range(3):
    step.eager(x, actions)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step.eager(x, actions)
    torch.cuda.synchronize()
evs = [e for e in prof.events() if e.device_type == torch.autograd.DeviceType.CPU and e.name.startswith('aten::') and not e.cpu_parent or True]
seen = []
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.name.startswith('aten::'):
        continue
    if e.cpu_parent is not None and e.cpu_parent.name.startswith('aten::'):
        continue
    kern = [k.name for k in e.kernels] if hasattr(e, 'kernels') else []
    child_k = []
    stack_ = [e]
    while stack_:
        n = stack_.pop()
        child_k += [k.name for k in getattr(n, 'kernels', [])]
        stack_ += list(n.cpu_children)
    if not child_k:
        continue
    where = [s for s in (e.stack or []) if 'stove_amd' in s or 'tools/' in s][:2]
    seen.append((e.time_range.start, e.name, [k[:50] for k in child_k], where))
seen.sort()
for t, name, ks, where in seen:
    print('%-28s %-60s %s' % (name, ','.join(ks)[:60], ' <- '.join(w.split('/')[-1] for w in where)))
print(len(seen), 'ATen calls with kernels')
