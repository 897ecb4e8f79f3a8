"""Measurement aid: per-stage cycle stamps of one GNN forward+backward step (workgroup 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stove_amd import _lib, ops
from stove_amd._lib import ptr, stream, check
from stove_amd.video_prediction.config import StoveConfig
from stove_amd.video_prediction.dynamics import Dynamics

dev = torch.device('cuda:0')
cfg = StoveConfig(); cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
cfg.device, cfg.dtype, cfg.random_seed = dev, torch.float32, 42
cfg.action_conditioned = False
dyn = Dynamics(cfg).to(dev)
img = dyn.param_image(0)
params = torch.cat([img[0], img[2], img[1]]).detach().contiguous()
B, N = 256, 3
lib = _lib.load()
s = torch.rand(B, N, 16, device=dev) - 0.5
d = torch.rand(B, N, 32, device=dev)
ds = torch.empty_like(s)
ws = torch.empty(lib.stove_gnn_bwd_ws_bytes(B, N) // 4 + 1, device=dev)
stamps = torch.zeros(64, dtype=torch.int64, device=dev)
for it in range(3):
    check(lib.stove_gnn_debug_stamps(ptr(s), ptr(params), ptr(d), ptr(ds), ptr(ws), ptr(stamps), B, N, 16, 2, 0, stream()), 'dbg')
torch.cuda.synchronize()
t = stamps.cpu().tolist()
names = {0: 'fwd start'}
prev = None
for k, v in enumerate(t):
    if v == 0:
        continue
    if prev is not None:
        print('stage %2d  +%6d cycles' % (k, v - prev))
    prev = v
fw = [k for k, v in enumerate(t) if v and k < 20]; bw = [k for k, v in enumerate(t) if v and k >= 20]
print('forward total', t[fw[-1]] - t[fw[0]], 'backward total', t[bw[-1]] - t[bw[0]], 'cycle counter units')
