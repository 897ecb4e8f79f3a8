"""Wall time per training step of the Trainer's own loop body (loader batch + replayed step) at the reference's default training shape
(256 clips x 8 visible frames), against the device time of the step: is the host (loader indexing, copies, Python) in the way?
Usage: python tools/trainer_loop_probe.py [steps]"""
import os
import pickle
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stove_amd.envs import envs  # noqa: E402
import model.main as M  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
tmp = tempfile.mkdtemp()
d = envs.synth_sequences('billiards', 300, 100, seed0=0)
data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2}
path = os.path.join(tmp, 'train.pkl')
with open(path, 'wb') as f:
    pickle.dump(data, f)
args = {'traindata': path, 'testdata': path, 'experiment_dir': tmp, 'dtype': 'torch.float', 'random_seed': '42', 'nolog': 'True',
        'print_every': str(10 ** 9), 'num_workers': '0', 'save_every': str(10 ** 9), 'long_rollout_every': str(10 ** 9)}
tr = M.main(sh_args=args)
for mode in ('graph', 'eager'):
    tr.c.graph_step = mode == 'graph'
    it = iter(tr.dataloader)
    n, t0, ev0, ev1 = 0, None, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    host = []
    while n < steps + 20:
        try:
            batch = next(it)
        except StopIteration:
            it = iter(tr.dataloader)
            continue
        if n == 20:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record()
        h0 = time.perf_counter()
        if tr._graph_ok(n + 1):
            tr._graph_step(batch, n + 1)
            out = tr._graphed.static_images()
            tr.dataloader.present_images_out = out
        else:
            tr.train_step(batch, n + 1)
        host.append(time.perf_counter() - h0)
        n += 1
    ev1.record()
    t_enq = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print('%-5s: %.3f ms wall per step (%.3f ms to enqueue), device %.3f ms per step, host %.3f ms per step body' % (
        mode, t_all / steps * 1e3, t_enq / steps * 1e3, ev0.elapsed_time(ev1) / steps, float(np.median(host[20:])) * 1e3))
