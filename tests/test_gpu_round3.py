"""Round-3 additions on the GPU: the device frame stores (bw plane / uint8), bench.py's own rank launcher, the gradient-view
registry across arenas, FlatAdam's zero-gradient reading, the action-conditioned replayed step."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _dataset(preset='billiards', n_seq=5, t_len=30):
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.load_data import StoveDataset
    cfg = StoveConfig()
    cfg.num_episodes, cfg.num_visible, cfg.num_rollout, cfg.frame_step = n_seq, 6, 3, 2
    d = envs.synth_sequences(preset, n_seq, t_len)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 10, 'r': 1.2}
    return StoveDataset(cfg, data=data)


def test_frame_store_formats():
    """DeviceClipLoader frame stores: 'bw32' batches are bit-identical to bw_transform of the colour
    batches, 'u8' batches are round(255 x); gathering into a caller's static buffer gives the same batch; footprint of the
    reference's training-set size (1000 x 100 frames of 32 x 32)."""
    from stove_amd.utils.utils import bw_transform
    from stove_amd.video_prediction.load_data import DeviceClipLoader
    ds = _dataset()
    dev = torch.device('cuda:0')
    f32 = DeviceClipLoader(ds, 7, dev, torch.float32, shuffle=False, frame_store='f32')
    bw = DeviceClipLoader(ds, 7, dev, torch.float32, shuffle=False, frame_store='bw32')
    u8 = DeviceClipLoader(ds, 7, dev, torch.float32, shuffle=False, frame_store='u8')
    assert bw.store_bytes() < f32.store_bytes() and u8.store_bytes() < f32.store_bytes()
    static = None
    n = 0
    for a, b, c in zip(f32, bw, u8):
        for k in ('present_images', 'future_images'):
            assert b[k].shape == a[k].shape[:2] + (1,) + a[k].shape[3:] and b[k].dtype == torch.float32
            assert torch.equal(b[k], bw_transform(a[k])), k
            assert c[k].dtype == torch.uint8 and torch.equal(c[k].cpu(), torch.round(a[k].cpu().double() * 255).to(torch.uint8)), k
        for k in a:
            if 'images' not in k:
                assert torch.equal(a[k], b[k]) and torch.equal(a[k], c[k])
        if static is None:
            static = torch.empty_like(b['present_images'])
            bw.present_images_out = static
        else:
            assert b['present_images'].data_ptr() == static.data_ptr()      # gathered straight into the caller's buffer
        n += 1
    assert n > 3

    class Fake:                      # the reference's data set size without allocating it
        rl = False
        total_img = np.broadcast_to(np.zeros(1, np.float32), (1000, 100, 3, 32, 32))
        total_data = np.broadcast_to(np.zeros(1), (1000, 100, 3, 4))
    gb = {fs: DeviceClipLoader.nbytes(Fake, torch.float32, fs) / 1e9 for fs in ('f32', 'bw32', 'u8')}
    assert 1.2 < gb['f32'] < 1.3 and gb['bw32'] < 0.42 and gb['u8'] < 0.32, gb


def test_stove_forward_takes_every_frame_store_format():
    """Stove.forward on colour fp32 frames, on the precomputed bw plane (config.input_bw_plane) and on uint8 colour frames:
    the bw plane gives the bit-identical ELBO, uint8 the ELBO of the quantised frames."""
    from stove_amd.envs import envs
    from stove_amd.utils.utils import bw_transform
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype, cfg.action_conditioned, cfg.action_space = dev, torch.float32, False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    B, T = 8, 10
    x = torch.from_numpy(envs.synth_sequences('billiards', B, T, seed0=5)['X']).to(dev).contiguous()
    g = torch.Generator().manual_seed(3)
    noise = {'latent': torch.randn(B, 3, 12, generator=g).to(dev), 'std': torch.randn(B, 3, 12, generator=g).to(dev),
             'steps': torch.randn(B, T - 2, 3, 18, generator=g).to(dev)}
    model.noise_fn = lambda kind, shape: noise[kind].reshape(shape)

    def elbo(inp, bw_plane=False):
        cfg.input_bw_plane = bw_plane
        with torch.no_grad():
            e, _, _ = model(inp, 1)
        cfg.input_bw_plane = False
        return float(e)
    e_colour = elbo(x)
    assert elbo(bw_transform(x), bw_plane=True) == e_colour
    q = torch.round(x * 255).to(torch.uint8)
    e_u8 = elbo(q)
    e_q = elbo(q.float() / 255)
    assert abs(e_u8 - e_q) <= 1e-6 * abs(e_q), (e_u8, e_q)
    assert abs(e_u8 - e_colour) <= 2e-2 * abs(e_colour)           # quantisation moves the ELBO, visibly but not wildly
    # the u8 kernel against the reference formula on the quantised frames
    ref = torch.clamp((q.float() / 255).sum(2), 0, 1).unsqueeze(2)
    assert float((bw_transform(q) - ref).abs().max()) <= 1.2e-7


@pytest.mark.parametrize('mode', ['graph', 'eager'])
def test_bench_launches_its_own_ranks(mode):
    """`python bench.py --gpus 2` with no rendezvous in the environment starts two ranks itself (gloo here: both share cuda:0)
    and rank 0 prints ONE JSON line for the whole job."""
    env = dict(os.environ, STOVE_DIST_BACKEND='gloo')
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--batch', '8',
                          '--frames', '12', '--no-cpu-baseline', '--no-variants', '--profile-steps', '0', '--step-mode', mode],
                         env=env, capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, res.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_batch'] == 16 and np.isfinite(out['value']) and out['value'] > 0
    assert out['scaling'] == 'weak' and out['config']['parallelism'] == 'dp2'


def test_bench_refuses_mismatched_world():
    env = dict(os.environ, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    res = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '1', '--warmup', '0'],
                         env=env, capture_output=True, text=True, timeout=300)
    assert res.returncode != 0 and 'WORLD_SIZE' in (res.stderr + res.stdout)


def test_grad_view_registry_follows_the_live_arena():
    """ops._GRAD_VIEWS is keyed on parameter objects: a replaced arena's views are not written any more, the new arena gets the
    recognition network's gradients, and torch.autograd.grad on an arena-bound weight still returns its gradient."""
    from stove_amd import ops
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype, cfg.action_conditioned, cfg.action_space = dev, torch.float32, False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(dev)
    x = torch.from_numpy(envs.synth_sequences('billiards', 8, 10, seed0=1)['X']).to(dev).contiguous()
    old = ParamArena(model, 1)
    old_grad = old.grad
    new = ParamArena(model, 1)                    # rebinds every parameter and its gradient view
    w_ih = model.sup.encoder.rnn.weight_ih_l0
    torch.manual_seed(1)
    elbo, _, _ = model(x, 1)
    (-elbo).backward()
    assert float(old_grad.abs().max()) == 0.0                                   # nothing lands in the dead arena
    g_new = new.view_of(w_ih, new.grad).clone()
    assert float(g_new.abs().max()) > 0 and w_ih.grad.data_ptr() == new.view_of(w_ih, new.grad).data_ptr()
    n_live = len(ops._GRAD_VIEWS)
    del old, old_grad
    assert len(ops._GRAD_VIEWS) == n_live                                       # entries belong to parameters, not arenas
    # autograd.grad asks for ONE weight: the direct path steps aside and autograd returns the gradient
    new.zero()
    torch.manual_seed(1)
    elbo, _, _ = model(x, 1)
    (g_auto,) = torch.autograd.grad(-elbo, [w_ih])
    assert g_auto is not None and float((g_auto - g_new).abs().max()) <= 1e-5 * float(g_new.abs().max())
    # a parameter whose .grad was detached from the arena (optimizer.zero_grad(set_to_none=True)) is not written behind autograd's back
    w_hh = model.sup.encoder.rnn.weight_hh_l0
    g_hh = new.view_of(w_hh, new.grad).clone()
    new.zero()
    torch.manual_seed(1)
    elbo, _, _ = model(x, 1)
    (-elbo).backward()
    g_hh = new.view_of(w_hh, new.grad).clone()
    assert float(g_hh.abs().max()) > 0
    w_hh.grad = None
    torch.manual_seed(1)
    elbo, _, _ = model(x, 1)
    (-elbo).backward()
    assert w_hh.grad is not None and w_hh.grad.data_ptr() != new.view_of(w_hh, new.grad).data_ptr()
    assert float((w_hh.grad - g_hh).abs().max()) <= 1e-5 * float(g_hh.abs().max())
    # ... and die with their parameters: a model built and dropped inside a scope of its own leaves nothing behind
    import gc

    def scoped():
        m = Stove(cfg).to(dev)
        ParamArena(m, 1)
        return set(ops._GRAD_VIEWS)
    gc.collect()
    before = set(ops._GRAD_VIEWS)
    inside = scoped()
    gc.collect()
    assert len(inside - before) > 100 and not (set(ops._GRAD_VIEWS) - before)       # every entry made inside is gone again


def test_flat_adam_zero_gradient_is_read_as_no_gradient():
    """The documented deviation (stove_amd/optim.py): torch.optim.Adam keeps moving a parameter whose .grad is a ZERO tensor on
    its momentum; FlatAdam reads an all-zero slice as `grad is None` and leaves the tensor (and its step count) alone."""
    from stove_amd.arena import ParamArena
    from stove_amd.optim import FlatAdam
    dev = torch.device('cuda:0')

    def make():
        torch.manual_seed(0)
        return torch.nn.Linear(8, 4).to(dev)
    a, b = make(), make()
    arena = ParamArena(a, 1)
    fo, to = FlatAdam(arena, lr=1e-2, amsgrad=True), torch.optim.Adam(b.parameters(), lr=1e-2, amsgrad=True)
    x = torch.randn(16, 8, device=dev)
    for net, opt, zero in ((a, fo, arena.zero), (b, to, lambda: to.zero_grad(set_to_none=False))):
        zero()
        net(x).pow(2).sum().backward()
        opt.step()
    assert float((a.weight - b.weight).abs().max()) < 1e-6                      # one real step: the same
    wa, wb = a.weight.detach().clone(), b.weight.detach().clone()
    arena.zero()
    to.zero_grad(set_to_none=False)                                               # zero tensors, as the reference's zero_grad leaves them
    fo.step()
    to.step()
    assert torch.equal(a.weight, wa)                                              # FlatAdam: untouched
    assert not torch.equal(b.weight, wb)                                          # torch Adam: moved by its momentum
    assert float(fo._seg_steps.max()) == 1.0


def test_graphed_step_action_conditioned_matches_eager():
    """The replayed step with actions, reward targets and the host-ramped reward weight (train.py:452-465) == the eager step."""
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.graphed import GraphedTrainStep
    from stove_amd.optim import FlatAdam
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    dev = torch.device('cuda:0')
    d = envs.synth_sequences('avoidance', 8, 10, seed0=2)
    x = torch.from_numpy(d['X']).to(dev).contiguous()
    act = torch.from_numpy(d['action']).float().to(dev)
    target = torch.from_numpy(d['reward'] + 1).float().to(dev)[:, 2:]

    def run(graphed):
        cfg = StoveConfig()
        cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
        cfg.device, cfg.dtype = dev, torch.float32
        cfg.action_conditioned, cfg.action_space, cfg.debug_core_appearance = True, 9, True
        cfg.print_every, cfg.plot_every = 10 ** 9, 1e19
        torch.manual_seed(0)
        model = Stove(cfg).to(dev)
        table = {}

        def noise(kind, shape):
            key = (kind, tuple(shape))
            if key not in table:
                table[key] = torch.randn(shape, generator=torch.Generator().manual_seed(len(table) + 5)).to(dev)
            return table[key]
        model.noise_fn = noise
        arena = ParamArena(model, 1)
        opt = FlatAdam(arena, lr=cfg.learning_rate, amsgrad=True)
        step = GraphedTrainStep(model, arena, opt, clip=1.0, reward_loss=torch.nn.BCELoss())
        out = []
        for w in (10.0, 20.0):
            e = step(x, act, target, reward_weight=w) if graphed else step.eager(x, act, target, reward_weight=w)
            out.append((float(e), float(step.reward_value), arena.data.clone()))
        return out
    eager, graph = run(False), run(True)
    assert abs(eager[0][0] - graph[0][0]) <= 1e-5 * abs(eager[0][0]) and abs(eager[0][1] - graph[0][1]) <= 1e-5
    assert float((eager[0][2] - graph[0][2]).abs().max()) < 2e-4
    assert float((eager[1][2] - graph[1][2]).abs().max()) < 2e-3 and np.isfinite(graph[1][0])
    assert not torch.equal(graph[0][2], graph[1][2])


def test_caller_owned_fork_stream():
    """stove_set_fork_stream: the scene calls fork their background chain onto the caller's stream, onto no stream at all, or (restored)
    onto the library's own one -- same numbers every way."""
    from stove_amd import _lib
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.supair import Supair
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed, cfg.channels = 3, 32, 32, 42, 1
    cfg.device, cfg.dtype = dev, torch.float32
    torch.manual_seed(0)
    sup = Supair(cfg).to(dev)
    x = torch.from_numpy(envs.synth_sequences('billiards', 4, 6, seed0=9)['X']).to(dev).sum(2, keepdim=True).clamp(0, 1)
    z = torch.rand(4 * 6 * 3, 4, device=dev) * torch.tensor([0.3, 0.3, 1.6, 1.6], device=dev) + torch.tensor([0.15, 0.15, -0.8, -0.8], device=dev)
    lib = _lib.load()

    def run():
        zz = z.clone().requires_grad_()
        ll, _ = sup.likelihood(x, zz)
        ll.sum().backward()
        torch.cuda.synchronize()
        return ll.detach().clone(), zz.grad.clone()
    ref = run()
    mine = torch.cuda.Stream(device=dev)
    try:
        assert lib.stove_set_fork_stream(0, mine.cuda_stream, 0) == 0
        a = run()
        assert lib.stove_set_fork_stream(0, None, 0) == 0            # no fork: the chain runs on the call's stream
        b = run()
    finally:
        assert lib.stove_set_fork_stream(0, None, 1) == 0            # back to the library's own stream
    c = run()
    for got in (a, b, c):
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1])
    assert lib.stove_set_fork_stream(99, None, 1) != 0


# config.align_corners=True is served by the general likelihood path since round 4: tests/test_gpu_round4.py (g13 goldens)


@pytest.mark.parametrize('items,n_obj', [(1, 3), (37, 3), (25088, 3), (1000, 6)])
def test_reward_head_kernels_against_torch(items, n_obj):
    """csrc/reward_head.hip == the reference's nn.Sequential reward head (dynamics.py:62-70, 254-263): values and every gradient."""
    from stove_amd import ops
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.dynamics import Dynamics
    dev = torch.device('cuda:0')
    cfg = StoveConfig()
    cfg.num_obj, cfg.device, cfg.dtype, cfg.action_conditioned, cfg.action_space = n_obj, dev, torch.float32, True, 9
    torch.manual_seed(items)
    dyn = Dynamics(cfg).to(dev)
    pred = (torch.randn(items, n_obj, 32, device=dev) * 1.5).requires_grad_()
    w = torch.randn(items, 1, device=dev)
    ps = list(dyn.reward_head0.parameters()) + list(dyn.reward_head1.parameters())

    def run(fused):
        cfg.fused_reward_head = fused
        for p in ps + [pred]:
            p.grad = None
        r = dyn.reward_from_pred(pred)
        (r * w).sum().backward()
        return r.detach().clone(), pred.grad.clone(), [p.grad.clone() for p in ps]
    r0, gp0, g0 = run(False)          # the library path, fp32
    r1, gp1, g1 = run(True)
    r2, gp2, g2 = run(True)
    assert r1.shape == (items, 1) and float((r1 - r0).abs().max()) <= 2e-6
    assert float((gp1 - gp0).abs().max()) <= 2e-5 * float(gp0.abs().max()) + 1e-9
    for a, b in zip(g1, g0):
        assert float((a - b).abs().max()) <= 3e-5 * float(b.abs().max()) + 1e-7, (a.shape,)
    assert torch.equal(r1, r2) and torch.equal(gp1, gp2) and all(torch.equal(a, b) for a, b in zip(g1, g2))      # fixed summation order


def test_narrow_linear_against_torch():
    """The action embedding Linear(9, 4 N) through ops.linear's narrow-layer kernel == F.linear, with all three gradients."""
    from stove_amd import ops
    dev = torch.device('cuda:0')
    torch.manual_seed(0)
    for rows, i, o in ((25088, 9, 12), (5, 9, 24), (3000, 64, 64)):
        x = torch.randn(rows, i, device=dev, requires_grad=True)
        w = torch.randn(o, i, device=dev, requires_grad=True)
        b = torch.randn(o, device=dev, requires_grad=True)
        g = torch.randn(rows, o, device=dev)
        y = ops.linear(x, w, b)
        y.backward(g)
        got = (y.detach().clone(), x.grad.clone(), w.grad.clone(), b.grad.clone())
        x.grad = w.grad = b.grad = None
        yr = torch.nn.functional.linear(x.double(), w.double(), b.double())
        yr.backward(g.double())
        ref = (yr.detach(), x.grad.double(), w.grad.double(), b.grad.double())
        for a, c in zip(got, ref):
            assert float((a.double() - c).abs().max()) <= 2e-5 * float(c.abs().max()) + 1e-6, (rows, i, o)
