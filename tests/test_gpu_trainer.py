"""End-to-end on the GPU through the reference's entry point: model.main.main(sh_args) -> Trainer.train()."""
import pickle

import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _write(tmp_path, preset, n_seq, t_len):
    from stove_amd.envs import envs
    d = envs.synth_sequences(preset, n_seq, t_len)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2}
    if 'action' in d:
        data.update(action=d['action'], reward=d['reward'], done=np.zeros_like(d['reward']), action_space=9)
    path = str(tmp_path / f'{preset}.pkl')
    with open(path, 'wb') as f:
        pickle.dump(data, f)
    return path


@pytest.mark.parametrize('preset', ['billiards', 'avoidance'])
def test_main_train_test_rollout(tmp_path, preset):
    import model.main as M
    path = _write(tmp_path, preset, 6, 24)
    args = {'traindata': path, 'testdata': path, 'nolog': 'True', 'experiment_dir': str(tmp_path), 'batch_size': '4',
            'num_visible': '6', 'num_rollout': '4', 'num_workers': '0', 'dtype': 'torch.float', 'random_seed': '42',
            'print_every': '1', 'num_epochs': '1', 'long_rollout_every': '1000000', 'save_every': '1000000'}
    if preset == 'avoidance':
        args['debug_core_appearance'] = 'True'
    trainer = M.main(sh_args=args)
    assert trainer.c.num_obj == 3 and trainer.c.action_conditioned == (preset == 'avoidance')
    before = [p.detach().clone() for p in trainer.stove.parameters()]
    it = iter(trainer.dataloader)
    elbos = []
    for step in range(1, 4):
        elbo, prop, rewards, min_ll, _ = trainer.train_step(next(it), step)
        assert torch.isfinite(elbo) and torch.isfinite(min_ll)
        elbos.append(float(elbo))
    changed = sum(int(not torch.equal(a, b)) for a, b in zip(before, trainer.stove.parameters()))
    assert changed > 100                         # every used parameter moved
    flat = trainer.bucket.pack()                 # the DP exchange buffer: one contiguous tensor holding every gradient
    assert flat.is_contiguous() and flat.numel() >= sum(p.numel() for p in trainer.stove.parameters())
    assert flat.numel() < sum(p.numel() + 4 for p in trainer.stove.parameters())      # 16-byte alignment pads only
    trainer.test(3, 0.0)                         # ELBO + 4-frame rollout errors through the logger
    out = trainer.long_rollout(idx=[0, 1], num=12)
    assert out['z_pred'].shape == (2, 12, 3, 18) and np.isfinite(out['z_pred']).all()
    # rendered clips (reference train.py:803-815): real frames after `skip`, reconstruction + rollout, reconstruction
    skip = trainer.c.skip
    assert out['frames_real'].shape == (2, 6 - skip, 1, 32, 32) and out['frames_real'].dtype == np.uint8
    assert out['frames_recon'].shape == (2, 6 - skip, 1, 32, 32)
    assert out['frames_rollout'].shape == (2, 6 - skip + 12, 1, 32, 32)
    assert (out['frames_rollout'][:, :6 - skip] == out['frames_recon']).all()
    assert out['frames_recon'].std() > 0
    # with logging on, the clips land in the experiment's gifs directory
    trainer.c.nolog = False
    from stove_amd.utils.utils import ExperimentLogger
    trainer.logger = ExperimentLogger(trainer.c)
    trainer.long_rollout(idx=[0], num=3)
    files = sorted(os.listdir(trainer.logger.rollout_gifs_dir))
    assert [f.split('.')[0] for f in files] == ['real', 'recon', 'rollout']


FULL_SIZE = {
    'billiards': dict(num_obj=3),                                                         # BASELINE.json configs[1]
    'gravity': dict(num_obj=3),                                                           # configs[2] (per-GPU shard of 256)
    'multibilliards': dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22),      # configs[3]
    'avoidance': dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True),             # configs[4]
}


@pytest.mark.parametrize('workload', list(FULL_SIZE))
def test_full_size_properties(workload):
    """BASELINE.json's configurations at full size (256 sequences x 100 frames per GPU) are too large for the CPU oracle;
    at that size the hot path is checked through size-independent properties:
      * reproducibility: no atomics anywhere -- two runs give bit-identical ELBO and gradients;
      * linearity of the batch mean: ELBO(256 sequences) == mean of the ELBOs of its four 64-sequence shards, and the
        gradient of the whole batch == mean of the shard gradients (what data parallelism relies on).
    N = 6 runs the two-rows-per-wave small-graph recursion (csrc/gnn_small*.hip) and the greedy matcher, avoidance the
    action-conditioned inputs and the appearance embedding."""
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.width, cfg.height, cfg.random_seed = 32, 32, 42
    cfg.device, cfg.dtype = dev, torch.float32
    cfg.action_conditioned, cfg.action_space = False, None
    for k, v in FULL_SIZE[workload].items():
        setattr(cfg, k, v)
    o = cfg.num_obj
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    arena = ParamArena(model, 1)
    B, T = 256, 100
    data = envs.synth_sequences(workload, B, T, seed0=7)
    x = torch.from_numpy(data['X']).to(dev).contiguous()
    actions = torch.from_numpy(data['action']).float().to(dev) if 'action' in data else None
    g = torch.Generator(device='cpu').manual_seed(3)
    noise = {'latent': torch.randn(B, o, 12, generator=g).to(dev), 'std': torch.randn(B, o, 12, generator=g).to(dev),
             'steps': torch.randn(B, T - 2, o, 18, generator=g).to(dev)}

    def run(lo, hi):
        model.noise_fn = lambda kind, shape: noise[kind][lo:hi].reshape(shape)
        arena.zero_grad()
        elbo, _, _ = model(x[lo:hi], 1, actions[lo:hi] if actions is not None else None)
        (-elbo).backward()
        return float(elbo.detach()), arena.grad.clone()

    e1, g1 = run(0, B)
    e2, g2 = run(0, B)
    assert e1 == e2 and torch.equal(g1, g2)                       # bitwise reproducible
    assert np.isfinite(e1) and float(g1.abs().max()) > 0
    es, gs = zip(*[run(k * 64, (k + 1) * 64) for k in range(4)])
    assert abs(np.mean(es) - e1) < 5e-6 * abs(e1), (np.mean(es), e1)
    gm = torch.stack(gs).mean(0)
    assert float((gm - g1).abs().max()) < 3e-4 * float(g1.abs().max())


def test_device_clip_loader_equals_dataloader(tmp_path):
    """Batches gathered on the GPU from the device-resident dataset == the reference-style host DataLoader's batches."""
    from torch.utils.data import DataLoader
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.load_data import DeviceClipLoader, StoveDataset
    cfg = StoveConfig()
    cfg.num_episodes, cfg.num_visible, cfg.num_rollout, cfg.frame_step = 5, 6, 3, 2
    d = envs.synth_sequences('avoidance', 5, 30)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2,
            'action': d['action'], 'reward': d['reward'], 'done': np.zeros_like(d['reward']), 'action_space': 9}
    ds = StoveDataset(cfg, data=data)
    dev = torch.device('cuda:0')
    ours = DeviceClipLoader(ds, 7, dev, torch.float32, shuffle=False)
    ref = DataLoader(ds, batch_size=7, shuffle=False, drop_last=True)
    assert len(ours) == len(ref)
    n = 0
    for a, b in zip(ours, ref):
        assert set(a) == set(b)
        for k in a:
            assert a[k].shape == b[k].shape and a[k].is_cuda
            assert torch.equal(a[k].cpu(), b[k].float()), k
        n += 1
    assert n == len(ref) and n > 3


# the replayed step against the eager step, bit for bit, with poisoned scratch between replays: tests/test_gpu_replay.py


def test_trainer_graph_step(tmp_path):
    """config.graph_step through the reference's entry point: non-logging steps are replayed, logging steps run eagerly."""
    import model.main as M
    path = _write(tmp_path, 'billiards', 12, 24)
    args = {'traindata': path, 'testdata': path, 'nolog': 'True', 'experiment_dir': str(tmp_path), 'batch_size': '8',
            'num_visible': '6', 'num_rollout': '4', 'num_workers': '0', 'dtype': 'torch.float', 'random_seed': '42',
            'print_every': '5', 'num_epochs': '1', 'long_rollout_every': '1000000', 'save_every': '1000000', 'graph_step': 'True'}
    trainer = M.main(sh_args=args)
    before = trainer.bucket.data.clone()
    trainer.train()
    n_steps = len(trainer.dataloader)
    assert n_steps >= 10 and trainer.optimizer._steps == n_steps
    assert trainer._graphed is not None and trainer._graphed.graphs is not None
    assert torch.isfinite(trainer.bucket.data).all() and not torch.equal(before, trainer.bucket.data)


def test_direct_arena_gradients_equal_autograd_accumulation():
    """The recognition network adds its parameter gradients into the arena's views inside the producing kernels
    (ops._grad_views; weight-gradient GEMMs on the second stream) instead of returning them to autograd.  Same numbers bit
    for bit as the AccumulateGrad path, gradients accumulate over two backward passes, and a parameter without a registered
    view (no arena) still gets its gradient from autograd."""
    from stove_amd import ops
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.width, cfg.height, cfg.random_seed, cfg.num_obj = 32, 32, 42, 3
    cfg.device, cfg.dtype = dev, torch.float32
    cfg.action_conditioned, cfg.action_space = False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    arena = ParamArena(model, 1)
    B, T = 16, 12
    x = torch.from_numpy(envs.synth_sequences('billiards', B, T, seed0=3)['X']).to(dev).contiguous()
    g = torch.Generator(device='cpu').manual_seed(3)
    noise = {'latent': torch.randn(B, 3, 12, generator=g).to(dev), 'std': torch.randn(B, 3, 12, generator=g).to(dev),
             'steps': torch.randn(B, T - 2, 3, 18, generator=g).to(dev)}
    model.noise_fn = lambda kind, shape: noise[kind].reshape(shape)

    def run(direct, passes=1):
        ops.DIRECT_GRADS = direct
        arena.zero_grad()
        for _ in range(passes):
            elbo, _, _ = model(x, 1, None)
            (-elbo).backward()
        torch.cuda.synchronize()
        return arena.grad.clone()
    try:
        ref = run(False)
        got = run(True)
        assert float(ref.abs().max()) > 0
        assert torch.equal(ref, got)
        twice = run(True, passes=2)
        assert float((twice - 2 * ref).abs().max()) <= 1e-6 * float(ref.abs().max())
        assert torch.equal(twice, run(False, passes=2))
    finally:
        ops.DIRECT_GRADS = True
    # no arena: gradients come back through autograd
    torch.manual_seed(0)
    plain = Stove(cfg).to(dev)
    plain.noise_fn = model.noise_fn
    elbo, _, _ = plain(x, 1, None)
    (-elbo).backward()
    enc = plain.sup.encoder
    for p, q in zip(enc.parameters(), model.sup.encoder.parameters()):
        want = arena.view_of(q, ref)
        assert p.grad is not None and float((p.grad - want).abs().max()) <= 1e-5 * float(want.abs().max())
