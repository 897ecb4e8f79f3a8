"""Host-side enqueue time of one training step vs its GPU time (is the step launch-bound?).
python tools/host_time.py [batch [frames]]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stove_amd.arena import ParamArena  # noqa: E402
from stove_amd.optim import FlatAdam  # noqa: E402
from stove_amd.video_prediction.stove import Stove  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device('cuda:0')
cfg = bench.build_config('billiards', dev)
torch.manual_seed(0)
model = Stove(cfg).to(dev)
arena = ParamArena(model, 1)
opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)
x = torch.from_numpy(bench.make_batch('billiards', B, 100, 0)['X'])[:, :T].to(dev).contiguous()


def step(i):
    arena.zero()
    t0 = time.perf_counter()
    elbo, _, _ = model(x, i + 1, None)
    t1 = time.perf_counter()
    (-elbo).backward()
    t2 = time.perf_counter()
    opt.step(max_norm=cfg.clip_grad_norm if hasattr(cfg, 'clip_grad_norm') else 1.0)
    return t1 - t0, t2 - t1, time.perf_counter() - t2


for i in range(5):
    step(i)
torch.cuda.synchronize()
host = []
for i in range(10):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    f, b, o = step(i)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    host.append((f, b, o, t1 - t0, t2 - t0))
for h in host[-4:]:
    print('T=%d ' % T + 'B=%d host fwd %.2f ms  bwd %.2f ms  opt %.2f ms | enqueue total %.2f ms | step incl. GPU %.2f ms' % ((B,) + tuple(1e3 * v for v in h)))
