"""Data parallelism with the real HIP model: two processes (gloo, sharing cuda:0 on the 1-GPU box) each
run forward+backward on half of the batch and all-reduce the flat gradient bucket; the result must equal
the single-process gradient of the whole batch (the ELBO is a batch mean, stove.py:748)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from gpu_helpers import fill_analytic
from helpers import load_golden, t_

pytestmark = pytest.mark.gpu


def _cfg():
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
    cfg.device, cfg.dtype, cfg.random_seed = torch.device('cuda:0'), torch.float32, 42
    cfg.action_conditioned, cfg.action_space = False, None
    return cfg


def _run(model, x, gold, lo, hi):
    lat, sd = t_(gold['eps_lat'])[lo:hi, ..., 0].float(), t_(gold['eps_std'])[lo:hi, ..., 0].float()
    steps = t_(gold['eps_steps']).float().permute(1, 0, 2, 3)[lo:hi].contiguous()
    table = {'latent': lat, 'std': sd, 'steps': steps}
    model.noise_fn = lambda kind, shape: table[kind].reshape(shape)
    elbo, _, _ = model(x[lo:hi].to('cuda:0'), 1)
    (-elbo).backward()
    return float(elbo)


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    gold = load_golden('g7_stove_n3_f32')
    model = fill_analytic(Stove(_cfg())).to('cuda:0')
    bucket = ParamArena(model, world)
    x = t_(gold['x']).float()
    per = x.shape[0] // world
    elbo = _run(model, x, gold, rank * per, (rank + 1) * per)
    bucket.all_reduce()
    e = torch.tensor([elbo], dtype=torch.float64)
    dist.all_reduce(e)
    if rank == 0:
        torch.save({'grads': {n: p.grad.cpu() for n, p in model.named_parameters()}, 'elbo': float(e) / world,
                    'bucket_numel': bucket.grad.numel()}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_single_process(tmp_path):
    from stove_amd.video_prediction.stove import Stove
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'dp.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    gold = load_golden('g7_stove_n3_f32')
    model = fill_analytic(Stove(_cfg())).to('cuda:0')
    x = t_(gold['x']).float()
    elbo = _run(model, x, gold, 0, x.shape[0])
    used = [n for n, p in model.named_parameters() if p.grad is not None]
    ref = torch.cat([p.grad.reshape(-1) for n, p in model.named_parameters() if p.grad is not None]).cpu()
    flat = torch.cat([got['grads'][n].reshape(-1) for n in used])
    assert abs(got['elbo'] - elbo) < 1e-5 * abs(elbo)
    rel = float((flat - ref).abs().max() / ref.abs().max())
    assert rel < 2e-4, rel
    assert flat.numel() == 1410255                 # the parameters that receive a gradient
    for n, g in got['grads'].items():              # the unused dynamics cores 1-2 stay exactly zero in the bucket
        if n not in used:
            assert float(g.abs().max()) == 0.0, n
    assert got['bucket_numel'] >= 1410255          # exchanged as ONE flat buffer (all parameters, 16-byte aligned)


# ---------------------------------------------------------------------------------------------------------------------
# The Trainer's data-parallel path, through the reference's entry point (model.main.main under a process group):
# rank 0's random_seed and parameters everywhere, disjoint clip shards, per-rank noise, one all-reduce, identical
# parameters after the step -- and the step equals ONE process stepping on the concatenated batch.
# ---------------------------------------------------------------------------------------------------------------------
def _write_data(tmp, n_seq=6, t_len=24):
    import pickle
    from stove_amd.envs import envs
    d = envs.synth_sequences('billiards', n_seq, t_len)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2}
    path = os.path.join(tmp, 'billiards.pkl')
    with open(path, 'wb') as f:
        pickle.dump(data, f)
    return path


def _args(path, tmp, **kw):
    a = {'traindata': path, 'testdata': path, 'nolog': 'True', 'experiment_dir': tmp, 'batch_size': '4',
         'num_visible': '6', 'num_rollout': '4', 'num_workers': '0', 'dtype': 'torch.float',
         'print_every': '1', 'num_epochs': '1', 'long_rollout_every': '1000000', 'save_every': '1000000'}
    a.update(kw)
    return a


def _spy_noise(stove):
    """records the step's draws: torch's (Stove._noise) or the library generator's (Stove._draw_ahead, config.device_noise 'philox')"""
    rec, orig, orig_ahead = {}, stove._noise, stove._draw_ahead

    def spy(kind, shape, like):
        t = orig(kind, shape, like)
        rec[kind] = t.detach().clone().cpu()
        return t

    def spy_ahead(numel, dev):
        t, ev = orig_ahead(numel, dev)
        torch.cuda.synchronize()
        rec['pooled'] = t.detach().clone().cpu()
        return t, ev
    stove._noise, stove._draw_ahead = spy, spy_ahead
    return rec


def _trainer_worker(rank, world, port, path, tmp):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK='0', STOVE_DIST_BACKEND='gloo')           # both ranks share cuda:0 on the 1-GPU box
    np.random.seed(1000 + rank)                     # the per-process seed draw of reference main.py:166-168 would differ
    torch.manual_seed(50 + rank)                    # ... and so would the initial weights
    import model.main as M
    trainer = M.main(sh_args=_args(path, tmp))      # random_seed None
    assert trainer.world_size == 2 and trainer.rank == rank and trainer.c.world_size == 2
    assert (trainer.logger is not None) == (rank == 0)
    start = trainer.bucket.data.clone().cpu()
    rec = _spy_noise(trainer.stove)
    it = iter(trainer.dataloader)
    data = next(it)
    ids = trainer.dataloader.last_clip_ids.clone()
    elbo, _, _, _, _ = trainer.train_step(data, 1)
    out = {'seed': trainer.c.random_seed, 'dp_seed': trainer.c.dp_seed, 'start': start, 'ids': ids, 'noise': dict(rec),
           'images': data['present_images'].cpu(), 'elbo': float(elbo.detach()), 'grad': trainer.bucket.grad.clone().cpu(),
           'after': trainer.bucket.data.clone().cpu(), 'len': len(trainer.dataloader),
           'steps': trainer.optimizer.state_dict()['state'][0]['step']}
    # a second step keeps the replicas together (moments and step counts are part of the state)
    trainer.train_step(next(it), 2)
    out['after2'] = trainer.bucket.data.clone().cpu()
    # a third step as the Trainer runs its non-logging steps: two replayed graphs around the all-reduce (stove_amd/graphed.py)
    trainer.c.print_every = 10 ** 9
    del trainer.stove._noise, trainer.stove._draw_ahead            # the recording wrappers copy to the host: not inside a capture
    assert trainer._graph_ok(3)
    trainer._graph_step(next(it), 3)
    out['after3'] = trainer.bucket.data.clone().cpu()
    out['graphs'] = len(trainer._graphed.graphs)
    trainer.c.print_every = 1
    # sync_replicas carries ALL of rank 0's optimiser state, the per-tensor step counts included
    if rank == 1:
        trainer.optimizer._seg_steps.add_(3.0)
        trainer.optimizer._flat['exp_avg'].add_(1.0)
    trainer.sync_replicas()
    out['seg_steps'] = trainer.optimizer._seg_steps.clone().cpu()
    out['exp_avg_sum'] = float(trainer.optimizer._flat['exp_avg'].double().sum())
    trainer.test(2, 0.0)                            # evaluation: rank 0 only, no collective inside
    torch.save(out, os.path.join(tmp, 'rank%d.pt' % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_trainer_data_parallel_step(tmp_path):
    tmp = str(tmp_path)
    path = _write_data(tmp)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_trainer_worker, args=(2, port, path, tmp), nprocs=2, join=True)
    r0, r1 = torch.load(os.path.join(tmp, 'rank0.pt')), torch.load(os.path.join(tmp, 'rank1.pt'))
    assert r0['seed'] == r1['seed'] and r0['dp_seed'] == r1['dp_seed']                # rank 0's draws on both ranks
    assert torch.equal(r0['start'], r1['start'])                                      # parameters broadcast at start
    assert r0['len'] == r1['len'] and not (set(r0['ids'].tolist()) & set(r1['ids'].tolist()))      # disjoint clips
    assert not torch.equal(r0['images'], r1['images'])
    assert not torch.equal(r0['noise']['pooled'], r1['noise']['pooled'])              # per-rank noise streams
    assert torch.equal(r0['grad'], r1['grad'])                                        # the reduced gradient ...
    assert torch.equal(r0['after'], r1['after']) and torch.equal(r0['after2'], r1['after2'])     # ... and the replicas stay equal
    assert not torch.equal(r0['after'], r0['start']) and float(r0['steps']) == 1.0
    assert torch.equal(r0['after3'], r1['after3']) and not torch.equal(r0['after3'], r0['after2']) and r0['graphs'] == 2
    assert torch.equal(r0['seg_steps'], r1['seg_steps']) and float(r0['seg_steps'].max()) == 3.0 and r0['exp_avg_sum'] == r1['exp_avg_sum']

    # one process, the concatenated batch, the same noise: the same step
    import model.main as M
    trainer = M.main(sh_args=_args(path, tmp, random_seed=str(r0['seed']), batch_size='8'))
    with torch.no_grad():
        trainer.bucket.data.copy_(r0['start'].to(trainer.bucket.data.device))
    # the ranks drew [latent | std | steps] as one buffer each (stove.py `pooled`): cut and concatenate along the batch
    nb, o, Ts = r0['images'].shape[0], 3, r0['images'].shape[1] - 2
    nl = nb * o * 12
    cut = lambda p: {'latent': p[:nl].view(nb, o, 12), 'std': p[nl:2 * nl].view(nb, o, 12), 'steps': p[2 * nl:].view(nb, Ts, o, 18)}
    c0, c1 = cut(r0['noise']['pooled']), cut(r1['noise']['pooled'])
    noise = {k: torch.cat([c0[k], c1[k]]) for k in c0}
    trainer.stove.noise_fn = lambda kind, shape: noise[kind].reshape(shape)
    images = torch.cat([r0['images'], r1['images']])
    elbo, _, _, _, _ = trainer.train_step({'present_images': images}, 1)
    assert abs(float(elbo) - (r0['elbo'] + r1['elbo']) / 2) < 1e-5 * abs(float(elbo))
    g1, gd = trainer.bucket.grad.cpu(), r0['grad']
    assert float((g1 - gd).abs().max()) < 2e-4 * float(g1.abs().max())
    # Adam's first step is lr * g / (|g| + eps): elements whose gradient is ~0 may step either way, the rest agree
    diff = (trainer.bucket.data.cpu() - r0['after']).abs()
    assert float(diff.median()) < 1e-6 and float((diff > 1e-4).float().mean()) < 2e-3, (float(diff.median()), float((diff > 1e-4).float().mean()))


def _rccl_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from stove_amd.arena import ParamArena
    from stove_amd import parallel
    from stove_amd.video_prediction.stove import Stove
    model = Stove(_cfg()).to('cuda:0')
    arena = ParamArena(model)
    arena.grad.copy_(torch.sin(torch.arange(arena.numel, device='cuda:0') * 0.01))
    before = arena.grad.clone()
    arena.all_reduce(force=True)                    # the flat bucket through RCCL (one rank: sum == identity)
    dist.broadcast(arena.data, 0)                   # what ParamArena.sync issues
    ok = torch.equal(arena.grad, before) and parallel.agree_on_seed(None) >= 0 and parallel.broadcast_int(41) == 41
    torch.cuda.synchronize()
    torch.save({'ok': bool(ok), 'backend': dist.get_backend()}, out)
    dist.destroy_process_group()


def test_arena_all_reduce_through_rccl(tmp_path):
    """RCCL itself (backend 'nccl' on ROCm) moves the arena's flat gradient buffer: a one-rank group on the 1-GPU box."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rccl.pt')
    mp.spawn(_rccl_worker, args=(port, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got['ok'] and got['backend'] == 'nccl'


def _rccl_graph_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    torch.cuda.set_device(0)
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.graphed import GraphedTrainStep
    from stove_amd.optim import FlatAdam
    from stove_amd.video_prediction.stove import Stove
    cfg = _cfg()
    cfg.print_every, cfg.plot_every = 10 ** 9, 1e19
    torch.manual_seed(0)
    model = Stove(cfg).to('cuda:0')
    arena = ParamArena(model, 1)
    opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=True)
    x = torch.from_numpy(envs.synth_sequences('billiards', 8, 10, seed0=3)['X']).to('cuda:0').contiguous()
    # the process group (and its watchdog thread) is alive while the step is captured; every replay sends the flat gradient
    # through RCCL, as a data-parallel run does: as a node of the optimiser graph (the default) and as an eager call between the
    # graphs (capture_reduce=False, the fallback) -- the same parameters after four steps either way
    snap = (arena.data.clone(), {k: v.clone() for k, v in opt._flat.items()}, opt._seg_steps.clone(), torch.cuda.get_rng_state(0))
    res = {}
    for captured in (True, False):
        with torch.no_grad():
            arena.data.copy_(snap[0])
            for k, v in snap[1].items():
                opt._flat[k].copy_(v)
            opt._seg_steps.copy_(snap[2])
        opt._steps = 0
        torch.cuda.set_rng_state(snap[3], 0)
        step = GraphedTrainStep(model, arena, opt, clip=1.0, force_reduce=True, capture_reduce=captured)
        elbos = [float(step(x)) for _ in range(4)]
        torch.cuda.synchronize()
        res[captured] = dict(elbos=elbos, data=arena.data.clone(), graphs=len(step.graphs), reduce_captured=step.reduce_captured, steps=opt._steps)
        del step
    ok = (all(np.isfinite(res[k]['elbos']).all() for k in res) and not torch.equal(snap[0], res[True]['data']) and res[True]['graphs'] == 2
          and res[True]['steps'] == 4 and torch.equal(res[True]['data'], res[False]['data']) and res[True]['elbos'] == res[False]['elbos']
          and res[False]['reduce_captured'] is False)
    torch.save({'ok': bool(ok), 'elbos': res[True]['elbos'], 'reduce_captured': bool(res[True]['reduce_captured'])}, out)
    dist.destroy_process_group()


def test_graph_replay_with_rccl_between_the_graphs(tmp_path):
    """The replayed data-parallel step on RCCL itself: capture with a live NCCL process group, all-reduce between the captured
    graphs (a one-rank group on the 1-GPU box: the collective is the identity, the call path is the real one)."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'rccl_graph.pt')
    mp.spawn(_rccl_graph_worker, args=(port, out), nprocs=1, join=True)
    got = torch.load(out)
    assert got['ok'], got
    assert got['reduce_captured'], 'the all-reduce was not captured into the optimiser graph (fell back to the eager call)'


def _nccl2_worker(rank, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    torch.cuda.set_device(rank)
    dev = torch.device('cuda', rank)
    dist.init_process_group('nccl', rank=rank, world_size=2, device_id=dev)
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.graphed import GraphedTrainStep
    from stove_amd.optim import FlatAdam
    from stove_amd.video_prediction.stove import Stove
    cfg = _cfg()
    cfg.device = dev
    cfg.print_every, cfg.plot_every = 10 ** 9, 1e19
    torch.manual_seed(rank)                      # different initialisations: sync() has to make them one
    model = Stove(cfg).to(dev)
    arena = ParamArena(model, 2)
    arena.sync(0)
    opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=True)
    x = torch.from_numpy(envs.synth_sequences('billiards', 8, 10, seed0=3 + 8 * rank)['X']).to(dev).contiguous()
    torch.manual_seed(100 + rank)
    step = GraphedTrainStep(model, arena, opt, clip=1.0, world_size=2)
    elbos = [float(step(x)) for _ in range(3)]
    torch.cuda.synchronize()
    mine = arena.data.clone()
    other = [torch.empty_like(mine) for _ in range(2)]
    dist.all_gather(other, mine)
    same = torch.equal(other[0], other[1])
    if rank == 0:
        torch.save({'ok': bool(same and np.isfinite(elbos).all()), 'reduce_captured': bool(step.reduce_captured), 'elbos': elbos}, out)
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason='needs two GPUs (the driver\'s multi-GPU box); the one-rank RCCL tests above run everywhere')
def test_two_ranks_on_rccl_replayed_step(tmp_path):
    """Two processes, two GPUs, RCCL over xGMI: three replayed data-parallel steps (all-reduce captured into the optimiser graph)
    leave bit-identical parameters on both ranks."""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / 'nccl2.pt')
    mp.spawn(_nccl2_worker, args=(port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert got['ok'], got


def test_bench_eight_ranks_over_gloo_on_one_gpu():
    """The driver's multi-GPU invocation shape -- `bench.py --gpus 8` -> torch.distributed.run, 8 ranks, rendezvous on 127.0.0.1 --
    on the one-GPU box: STOVE_DIST_BACKEND=gloo, the ranks share cuda:0 (16 sequences each).  Everything that is not the wire runs:
    launch_ranks, the wait for rank 0's library build, per-rank data generation with cores // 8 workers, one seed / one parameter
    set on all ranks, the captured step with the eager all-reduce fallback between its graphs, barrier + max-over-ranks timing, and
    ONE JSON line from rank 0 with the whole-job aggregate at N = 8.  No scaling number comes out of it (one GPU)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, STOVE_DIST_BACKEND='gloo', STOVE_BENCH_NO_PARITY='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('WORLD_SIZE', None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--batch', '16', '--steps', '4', '--warmup', '2',
           '--no-variants', '--no-cpu-baseline', '--profile-steps', '0']
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 8 and d['steps'] == 4 and d['scaling'] == 'weak'
    assert d['config']['global_batch'] == 8 * 16 and d['config']['parallelism'] == 'dp8'
    assert abs(d['value'] - 8 * 16 * 100 / d['ms_per_step'] * 1e3) < 1e-6 * d['value']          # whole-job frames / max-over-ranks time
    assert d['all_reduce_in_graph'] is False                                                    # gloo: the eager all-reduce between the graphs
    assert np.isfinite(d['config']['elbo_last_step'])
    assert d['comm'] is not None and d['comm']['world_size_reported'] == 8 and d['comm']['backend'] == 'gloo', d['comm']
