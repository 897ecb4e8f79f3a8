"""Captured-graph training step vs the eager step at the reference's training shape (256 clips x 8 frames):
time per step and agreement of the parameters after the same number of steps from the same seed."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from stove_amd.arena import ParamArena  # noqa: E402
from stove_amd.graphed import GraphedTrainStep  # noqa: E402
from stove_amd.optim import FlatAdam  # noqa: E402
from stove_amd.video_prediction.stove import Stove  # noqa: E402

B, T = 256, int(sys.argv[1]) if len(sys.argv) > 1 else 8
STEPS = int(sys.argv[3]) if len(sys.argv) > 3 else 40
dev = torch.device('cuda:0')
data = torch.from_numpy(bench.make_batch('billiards', B, 100, 0)['X'])
batches = [data[:, s:s + T].to(dev).contiguous() for s in range(0, 80, 2)]


def build():
    cfg = bench.build_config('billiards', dev)
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    arena = ParamArena(model, 1)
    opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=cfg.debug_amsgrad)
    return model, arena, opt


FIXED = len(sys.argv) > 2 and sys.argv[2] == 'fixed'      # the same draws every step: eager and graphed must then agree


def run(graphed):
    model, arena, opt = build()
    torch.manual_seed(1)
    if FIXED:
        table = {}

        def noise(kind, shape):
            key = (kind, tuple(shape))
            if key not in table:
                table[key] = torch.randn(shape, generator=torch.Generator().manual_seed(len(table) + 5)).to(dev)
            return table[key]
        model.noise_fn = noise
    step = GraphedTrainStep(model, arena, opt, clip=1.0)
    elbos = []
    step(batches[0]) if graphed else step._eager(batches[0], None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(STEPS):
        e = step(batches[i % len(batches)]) if graphed else step._eager(batches[i % len(batches)], None)
        elbos.append(e.detach().clone())
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / STEPS
    return dt, torch.stack(elbos).cpu(), arena.data.detach().clone(), opt._steps


dt_e, el_e, p_e, n_e = run(False)
dt_g, el_g, p_g, n_g = run(True)
print('eager   %.3f ms/step, steps %d, elbo first/last %.2f %.2f' % (dt_e * 1e3, n_e, el_e[0], el_e[-1]))
print('graphed %.3f ms/step, steps %d, elbo first/last %.2f %.2f' % (dt_g * 1e3, n_g, el_g[0], el_g[-1]))
print('elbo diff per step:', ' '.join('%.1e' % float(v) for v in (el_e - el_g).abs()[:12]))
print('max |param diff| %.3e (max |param| %.3e), max |elbo diff| %.3e' % (float((p_e - p_g).abs().max()), float(p_e.abs().max()), float((el_e - el_g).abs().max())))
