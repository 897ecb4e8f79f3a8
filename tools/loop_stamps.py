"""Measurement aid: per-wave phase stamps of the small-graph time loops (workgroup 0, last step)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stove_amd import _lib, ops
from stove_amd.video_prediction.config import StoveConfig
from stove_amd.video_prediction.dynamics import Dynamics

dev = torch.device('cuda:0')
cfg = StoveConfig(); cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
cfg.device, cfg.dtype, cfg.random_seed = dev, torch.float32, 42
cfg.action_conditioned = False
dyn = Dynamics(cfg).to(dev)
B, Ts, N = 256, 98, 3
lib = _lib.load()
stamps = torch.zeros(2 * 64, dtype=torch.int64, device=dev)
z1 = (torch.rand(B, N, 18, device=dev) - 0.5).requires_grad_()
zsup = torch.rand(B, Ts, N, 6, device=dev) - 0.5
zsstd = torch.rand(B, Ts, N, 6, device=dev) * 0.1 + 0.05
eps = torch.randn(B, Ts, N, 18, device=dev)
image, _ = dyn.kernel_params(0)
for it in range(3):
    lib.stove_debug_set_stamps(stamps.data_ptr())
    out = ops.dyn_loop(z1, zsup, zsstd, eps, None, image, 2, False, dyn.loop_consts())
    torch.cuda.synchronize()
    fw = stamps.cpu().view(2, 4, 16).clone()      # [fwd | bwd][wave][phase]
    (out[0].sum() + out[1].sum() + out[3].sum() + out[4].sum()).backward()      # all four state gradients: the headline instantiation
    torch.cuda.synchronize()
    bw = stamps.cpu().view(2, 4, 16).clone()
lib.stove_debug_set_stamps(None)
for name, t in (('forward', fw[0]), ('backward', bw[1])):
    base = int(t[t > 0].min()) if (t > 0).any() else 0
    print(name, '(cycles since the earliest stamp of the step)')
    for w in range(4):
        print('  wave %d: ' % w + ' '.join('%6d' % (int(v) - base) if v > 0 else '     -' for v in t[w].tolist()))
