"""Training sanity run through the reference's entry point (model.main.main -> Trainer.train) on generated billiards data:
prints the ELBO / position-error log lines and the wall time per training step.  python tools/train_demo.py [epochs [eager]]"""
import os
import pickle
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from stove_amd.envs import envs  # noqa: E402
import model.main as M  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 2
graph = not (len(sys.argv) > 2 and sys.argv[2] == 'eager')      # config.graph_step (default on): non-logging steps replayed as captured hipGraphs
tmp = tempfile.mkdtemp()
paths = {}
for name, n_seq, seed0 in (('train', 300, 0), ('test', 40, 10 ** 5)):
    d = envs.synth_sequences('billiards', n_seq, 100, seed0=seed0)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2}
    paths[name] = os.path.join(tmp, name + '.pkl')
    with open(paths[name], 'wb') as f:
        pickle.dump(data, f)
args = {'traindata': paths['train'], 'testdata': paths['test'], 'experiment_dir': tmp, 'dtype': 'torch.float', 'random_seed': '42',
        'num_epochs': str(epochs), 'print_every': '50', 'num_workers': '0', 'save_every': '1000000', 'long_rollout_every': '1000000', 'graph_step': str(graph)}
trainer = M.main(sh_args=args)
t0 = time.time()
trainer.train()
dt = time.time() - t0
steps = epochs * len(trainer.dataloader)
print('trained %d steps (batch %d clips of %d frames) in %.1f s incl. tests and logging' % (steps, trainer.c.batch_size, trainer.c.num_visible, dt))
perf = os.path.join(trainer.logger.exp_dir, 'performance.csv')
lines = open(perf).read().strip().splitlines()
print(lines[0][:200])
for l in lines[1:4] + lines[-4:]:
    print(l[:200])
print(sorted(os.listdir(trainer.logger.rollout_gifs_dir)))
