"""Round-4 additions on the GPU: the library's counter-based noise generator, the step head off the serial chain."""
import numpy as np
import pytest
import torch

import stove_oracle as O
from gpu_helpers import check, check_grad, err, fill_analytic
from helpers import load_golden, oracle_setup, t_

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def test_noise_source_statistics_and_reproducibility():
    """ops.NoiseSource (csrc/state.hip noise_normal_k: Philox-4x32-10 + Box-Muller): standard normal moments, a pure function of
    (seed, call number, element), seeded by torch.manual_seed, advanced on the device."""
    from stove_amd import ops
    torch.manual_seed(123)
    src = ops.NoiseSource(DEV)
    a = src.normal(4_000_001)
    b = src.normal(4_000_001)
    torch.cuda.synchronize()
    assert torch.isfinite(a).all() and not torch.equal(a, b)
    x = a.double()
    assert abs(float(x.mean())) < 2e-3 and abs(float(x.var()) - 1.0) < 3e-3
    assert abs(float((x ** 3).mean())) < 1e-2 and abs(float((x ** 4).mean()) - 3.0) < 3e-2
    assert float(x.abs().max()) > 4.5 and float(x.abs().max()) < 6.5              # tails present, nothing absurd
    assert abs(float((a[:-1] * a[1:]).double().mean())) < 2e-3                    # neighbours (the two halves of a Box-Muller pair) uncorrelated
    assert abs(float((a * b).double().mean())) < 2e-3                             # successive calls uncorrelated
    assert int(src.state[1]) == 2
    # same seed, same call -> same numbers, whatever the length of the draw, also from another source object
    torch.manual_seed(123)
    src2 = ops.NoiseSource(DEV)
    c = src2.normal(1003)
    assert torch.equal(c, a[:1003])
    assert torch.equal(src2.normal(1003), b[:1003])
    # reseeding torch's generator -- with another value or the same one again -- restarts the stream, restoring its state resumes it
    torch.manual_seed(124)
    d = src2.normal(1003)
    assert not torch.equal(d, c) and int(src2.state[1]) == 1
    torch.manual_seed(123)
    assert torch.equal(src.normal(1003), a[:1003])
    keep = torch.cuda.get_rng_state(DEV)
    e1 = src.normal(64)
    torch.cuda.set_rng_state(keep, DEV)
    assert torch.equal(src.normal(64), e1) and torch.equal(e1, b[:64])


def test_model_draws_from_the_library_generator_by_default():
    """Stove.forward with config.device_noise = 'philox' (default): different noise every call, the same after a reseed; 'torch'
    keeps torch.randn."""
    from stove_amd.envs import envs
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype = DEV, torch.float32
    cfg.action_conditioned, cfg.action_space = False, None
    torch.manual_seed(0)
    model = Stove(cfg).to(DEV)
    x = torch.from_numpy(envs.synth_sequences('billiards', 4, 8, seed0=1)['X']).to(DEV)
    for mode in ('philox', 'torch'):
        cfg.device_noise = mode
        torch.manual_seed(5)
        e1 = float(model(x, 1)[0])
        e2 = float(model(x, 1)[0])
        torch.manual_seed(5)
        e3 = float(model(x, 1)[0])
        assert np.isfinite(e1) and e1 != e2 and e1 == e3, (mode, e1, e2, e3)
    assert model._noise_source.state is not None


def _greedy_serial(feat):
    """Stove._greedy_match_objects (reference stove.py:432-514) on one track, slot order, first minimum wins: numpy, exact inputs"""
    T, N, _ = feat.shape
    idx = np.zeros((T, N), dtype=np.int64)
    idx[0] = np.arange(N)
    for t in range(1, T):
        prev = (feat[t - 1][idx[t - 1]] + 1.0) * 0.5
        cur = (feat[t] + 1.0) * 0.5
        e = ((prev[:, None, :] - cur[None, :, :]) ** 2).sum(-1)
        for _ in range(N):
            a, j = np.unravel_index(np.argmin(e), e.shape)       # first occurrence in row-major (slot-major) order
            idx[t, a] = j
            e[a, :] = 3.0e38
            e[:, j] = 3.0e38
    return idx


@pytest.mark.parametrize('N,T,F', [(6, 100, 2), (6, 7, 2), (2, 2, 2), (8, 33, 2), (5, 100, 5), (4, 50, 2)])
def test_greedy_matcher_frames_in_parallel_equals_the_serial_walk(N, T, F):
    """csrc/match.hip match_greedy_frames_k + match_greedy_compose_k against the serial rule on tracks that sit on a coarse grid
    (every distance exact in fp32, exact ties in most frames: the tie flag and the slot-order redo are exercised) and on generic tracks."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(N * 100 + T)
    B = 48
    coarse = torch.randint(-8, 9, (B, T, N, F), generator=g).float() / 8.0
    smooth = (torch.rand(B, 1, N, F, generator=g) * 2 - 1) + 0.02 * torch.randn(B, T, N, F, generator=g).cumsum(1)
    smooth = (smooth * 4096).round() / 4096              # exact products: the numpy rule decides on the same numbers
    for name, feat in (('ties', coarse), ('generic', smooth)):
        idx, _ = ops.match_objects(feat.to(DEV), 'greedy')
        want = np.stack([_greedy_serial(feat[b].double().numpy()) for b in range(B)])
        got = idx.cpu().numpy()
        assert got.shape == (B, T, N) and (np.sort(got, -1) == np.arange(N)).all(), name        # permutations
        assert (got == want).all(), (name, int((got != want).sum()))
    # the fused state pipeline runs the same kernels (stove_supair_state_fwd2): exercised by the N = 6 goldens (g6 / g7_stove_n6)


# ---------------------------------------------------------------------------------------------------------------------
# beyond the 32 x 32 / align_corners=False contract (VERDICT r03 "what's missing" 1 and 3): the background SPN operator over any
# number of dimensions (csrc/spn_bg_generic.hip) and Supair.likelihood / Stove.forward composed from the transformer API + the
# two SPN operators, against reference-generated goldens (g13) at 50 x 50 and under align_corners=True
# ---------------------------------------------------------------------------------------------------------------------
def _cfg(**kw):
    from stove_amd.video_prediction.config import StoveConfig
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height = 3, 32, 32
    cfg.device, cfg.dtype, cfg.random_seed = DEV, torch.float32, 42
    cfg.action_conditioned, cfg.action_space = False, None
    cfg.debug = True
    for k, v in kw.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize('side,n', [(50, 7), (35, 33), (24, 130), (32, 9), (50, 600)])
def test_background_spn_operator_over_any_number_of_dimensions(side, n):
    """RatSpn.forward / backward of the background SPN at side x side pixels against the float64 oracle: 50 x 50 (two leaves of
    1250), 35 x 35 = 1225 (leaves of 612 and 613 pixels, three 512-lane slices with a ragged last one), 24 x 24 (ragged
    second slice), and 32 x 32 through the tuned kernels for comparison."""
    from stove_amd.spn import probabilistic_models as prob
    c, structs, params = oracle_setup(torch.float64, width=side, height=side)
    spn = fill_analytic(prob._get_bg_spn(c, 42), 'sup.bg_spn.').to(DEV)
    assert spn._kind == 'bg' and spn.num_dims == side * side
    d = side * side
    g = torch.Generator().manual_seed(side + n)
    x64 = torch.rand(n, d, generator=g, dtype=torch.float64)
    m64 = torch.rand(n, d, generator=g, dtype=torch.float64) * 1.4 - 0.2
    m64[0] = 0.0
    w64 = torch.linspace(0.5, 1.5, n, dtype=torch.float64)
    xo, mo = x64.clone().requires_grad_(), m64.clone().requires_grad_()
    out_o = O.spn_forward(structs['bg'], params, 'sup.bg_spn.', xo, mo, 6, 3, c.bg_min_var, c.bg_max_var)
    (out_o[:, 0] * w64).sum().backward()
    xd, md = x64.float().to(DEV).requires_grad_(), m64.float().to(DEV).requires_grad_()
    out_d = spn(xd, md)
    assert out_d.shape == (n, 1)
    check('bgspn_any.fwd', err(out_d, out_o), 5e-6)
    (out_d[:, 0] * w64.float().to(DEV)).sum().backward()
    check('bgspn_any.dx', err(xd.grad, xo.grad), 3e-4)
    check('bgspn_any.dmarg', err(md.grad, mo.grad), 3e-4)
    for name, p in spn.named_parameters():
        ref = params['sup.bg_spn.' + name].grad
        if name.startswith('output_vector'):
            continue
        check_grad('bgspn_any.grad', p.grad, ref, 3e-4, 3e-4, 5e-3)
    # marginalized=None and bitwise reruns
    o1, o2 = spn(xd.detach(), None), spn(xd.detach(), None)
    assert torch.equal(o1, o2)
    check('bgspn_any.fwd_nomarg', err(o1, O.spn_forward(structs['bg'], params, 'sup.bg_spn.', x64, None, 6, 3, c.bg_min_var, c.bg_max_var)), 5e-6)


WIDE = {'res50': dict(width=50, height=50), 'ac32': dict(align_corners=True)}


@pytest.mark.parametrize('composed', [False, True])
@pytest.mark.parametrize('name', list(WIDE))
def test_likelihood_beyond_the_32x32_contract(name, composed):
    """composed=False: the fused pipeline with run-time geometry (stove_scene_fwd_any / _bwd_any); True: the reference's op
    sequence on ATen's sampler + the HIP SPN operators (Supair._likelihood_general), both against the reference's own numbers."""
    from stove_amd.video_prediction.supair import Supair
    g = load_golden(f'g13_likelihood_{name}_f64')
    sup = fill_analytic(Supair(_cfg(scene_composed=composed, **WIDE[name])), 'sup.').to(DEV)
    sup.step_counter = 0
    x = t_(g['x']).float().to(DEV)
    z = t_(g['z']).float().to(DEV).requires_grad_()
    lp, prop = sup.likelihood(x, z)
    check('wide.log_p', err(lp, g['log_p']), 5e-6)
    for k in ('bg', 'patch', 'overlap'):
        check('wide.part_' + k, abs(float(prop[k]) - float(g[k])) / (abs(float(g[k])) + 1e-9), 5e-6)
    (lp * t_(g['w']).float().to(DEV)).sum().backward()
    check('wide.dz', err(z.grad, g['gz']), 3e-4)
    params = dict(sup.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            check('wide.grad_norm', abs(float(params[k[3:]].grad.norm()) - float(v)) / (float(v) + 1e-9), 3e-4)
            n += 1
        elif k.startswith('g_'):
            check('wide.grad_tensor', err(params[k[2:]].grad, v), 3e-4)
    assert n > 10


@pytest.mark.parametrize('shape', [(50, 50, False, 3), (40, 28, False, 3), (28, 40, True, 5), (64, 64, True, 8), (32, 32, True, 6),
                                   (136, 24, False, 2)])      # the last: a side past the coverage tables' 128 entries (mask image kept)
def test_fused_scene_of_any_geometry_against_the_oracle(shape):
    """Frames that are not square, more objects than the goldens hold, both conventions: the fused any-size pipeline against the
    float64 oracle on the same seeded inputs (the oracle itself is pinned on 50 x 50 / align_corners goldens of the reference,
    tests/test_oracle_goldens.py), a time-slice of longer clips handed over as a view, and two runs bit for bit."""
    from stove_amd.video_prediction.supair import Supair
    h, w, ac, n_obj = shape              # frames (.., c, width, height): the last dimension is grid_sample's x
    c = _cfg(width=h, height=w, align_corners=ac, num_obj=n_obj)
    sup = fill_analytic(Supair(c), 'sup.').to(DEV)
    sup.step_counter = 1
    gen = torch.Generator().manual_seed(100 + h + w)
    n, T = 3, 4
    x = torch.rand(n, T + 1, 1, h, w, generator=gen)
    z = torch.empty(n * T * n_obj, 4)
    z[:, 0] = 0.12 + 0.3 * torch.rand(z.shape[0], generator=gen)
    z[:, 1] = z[:, 0] * (0.8 + 0.4 * torch.rand(z.shape[0], generator=gen))
    z[:, 2:] = 1.9 * torch.rand(z.shape[0], 2, generator=gen) - 0.95            # some glimpses hang over the frame's edge
    wgt = torch.randn(n * T, generator=gen)
    xs = x.to(DEV)[:, 1:]                                                           # a view with a sequence stride
    zd = z.to(DEV).requires_grad_()
    lp, _ = sup.likelihood(xs, zd)
    (lp * wgt.to(DEV)).sum().backward()
    cfg, structs, params = oracle_setup(torch.float64, num_obj=n_obj, width=h, height=w, align_corners=ac)
    params = {k: v for k, v in params.items() if k.startswith('sup.')}
    z64 = z.double().requires_grad_()
    ref = O.scene_likelihood(cfg, params, structs, x[:, 1:].double(), z64)
    (ref * wgt.double()).sum().backward()
    check('any.log_p', err(lp, ref), 6e-6)
    check_grad('any.dz', zd.grad, z64.grad, 3e-4, 3e-4, 5e-3)
    got = dict(sup.named_parameters())
    for k, v in params.items():
        if v.grad is not None and float(v.grad.abs().max()) > 0:
            check_grad('any.grad', got[k[4:]].grad, v.grad, 3e-4, 3.5e-4, 1.5e-2)
    g1 = zd.grad.clone()
    zd.grad = None
    lp2, _ = sup.likelihood(xs, zd)
    (lp2 * wgt.to(DEV)).sum().backward()
    assert torch.equal(lp, lp2) and torch.equal(g1, zd.grad)


@pytest.mark.parametrize('name', list(WIDE))
@pytest.mark.parametrize('arena', [False, True])
def test_stove_forward_beyond_the_32x32_contract(name, arena):
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.stove import Stove
    g = load_golden(f'g13_stove_{name}_f64')
    st = fill_analytic(Stove(_cfg(**WIDE[name]))).to(DEV)
    if arena:
        ar = ParamArena(st)
        assert ar.has_gnn and ar.has_spn == (name == 'ac32')  # the SPN tables of other frame sizes are baked per tensor (RatSpn.tables + autograd)
    lat, sd = t_(g['eps_lat'])[..., 0].float(), t_(g['eps_std'])[..., 0].float()
    steps = t_(g['eps_steps']).float().permute(1, 0, 2, 3).contiguous()
    table = {'latent': lat, 'std': sd, 'steps': steps}
    st.noise_fn = lambda kind, shape: table[kind].reshape(shape)
    elbo, prop, _ = st(t_(g['x']).float().to(DEV), 0, None)
    check('wide.elbo_rel', abs(float(elbo) - float(g['elbo'])) / abs(float(g['elbo'])), 1.5e-6)
    for k in ('z', 'z_dyn', 'z_sup', 'log_q', 'translik', 'bg', 'patch', 'overlap'):
        check('wide.prop_' + k, err(prop[k], g['p_' + k]), 8e-6)
    (-elbo).backward()
    params = dict(st.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            check('wide.stove_grad_norm', abs(float(params[k[3:]].grad.norm()) - float(v)) / (float(v) + 1e-9), 2e-4)
            n += 1
        elif k.startswith('g_'):
            check_grad('wide.stove_grad_tensor', params[k[2:]].grad, v, 3e-4, 3.5e-4, 1.5e-2)      # entry-wise p99: fp32 sums over 2 500 pixels against float64
    assert n > 50
    with torch.no_grad():
        rec = st.reconstruct_from_z(prop['z'])
    check('wide.recon', float((rec.cpu().double() - t_(g['recon'])).abs().max()), 2e-5)


def test_training_on_the_reference_default_gravity_data_50x50(tmp_path):
    """run_stove.py's path (model.main.main -> Trainer.train) on gravity data rendered as the reference's stock generator does
    (res = 50, envs.py:841-844): eager logging steps, replayed steps, the test pass and a rollout with rendered clips."""
    import pickle
    import model.main as M
    from stove_amd.envs import envs
    d = envs.synth_sequences('gravity', 6, 20, res=50)
    assert d['X'].shape[-2:] == (50, 50)
    data = {'X': np.transpose(d['X'], (0, 1, 3, 4, 2)).astype(np.float64), 'y': d['y'], 'coord_lim': 30, 'r': 2}
    path = str(tmp_path / 'gravity50.pkl')
    with open(path, 'wb') as f:
        pickle.dump(data, f)
    args = {'traindata': path, 'testdata': path, 'nolog': 'True', 'experiment_dir': str(tmp_path), 'batch_size': '4',
            'num_visible': '6', 'num_rollout': '4', 'num_workers': '0', 'dtype': 'torch.float', 'random_seed': '42',
            'print_every': '4', 'num_epochs': '1', 'long_rollout_every': '1000000', 'save_every': '1000000'}
    trainer = M.main(sh_args=args)
    assert (trainer.c.width, trainer.c.height) == (50, 50) and trainer.stove.sup.bg_spn.num_dims == 2500
    before = trainer.bucket.data.clone()
    trainer.train()
    assert trainer.optimizer._steps == len(trainer.dataloader) and trainer.optimizer._steps >= 8
    assert trainer._graphed is not None and trainer._graphed.graphs is not None          # non-logging steps were replayed
    assert torch.isfinite(trainer.bucket.data).all() and not torch.equal(before, trainer.bucket.data)
    out = trainer.long_rollout(idx=[0, 1], num=5)
    assert out['frames_rollout'].shape[-2:] == (50, 50) and np.isfinite(out['z_pred']).all()


# ---------------------------------------------------------------------------------------------------------------------
# object SPNs of other shapes (VERDICT r03 "what's missing" 2): glimpse sizes and vector widths from the config
# ---------------------------------------------------------------------------------------------------------------------
SHAPE_8x12 = dict(patch_width=8, patch_height=12, obj_spn_num_gauss=7, obj_spn_num_sums=5)


def test_object_spn_operator_of_another_shape_against_the_reference():
    """csrc/spn_obj_generic.hip through RatSpn.forward against the reference's own numbers (g14): 96 dimensions in leaves of 24,
    7 Gaussians, 5 sums; out-of-range marginalisation; every parameter gradient; two runs bit for bit."""
    from stove_amd.spn import probabilistic_models as prob
    g = load_golden('g14_objspn_8x12_f64')
    c = _cfg(**SHAPE_8x12)
    spn = fill_analytic(prob._get_obj_spn(c, 42), 'sup.obj_spn.').to(DEV)
    assert spn._kind == 'obj_any' and spn.num_dims == 96
    x = t_(g['x']).float().to(DEV).requires_grad_()
    m = t_(g['marg']).float().to(DEV).requires_grad_()
    out = spn(x, m)
    check('objany.fwd', err(out, g['out']), 5e-6)
    (out[:, 0] * t_(g['w']).float().to(DEV)).sum().backward()
    check_grad('objany.dx', x.grad, g['gx'], 3e-4, 3e-4, 5e-3)
    check_grad('objany.dmarg', m.grad, g['gmarg'], 3e-4, 3e-4, 5e-3)
    params = dict(spn.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith('ogn_'):
            check('objany.grad_norm', abs(float(params[k[4:]].grad.norm()) - float(v)) / (float(v) + 1e-9), 3e-4)
            n += 1
        elif k.startswith('og_'):
            check_grad('objany.grad', params[k[3:]].grad, v, 3e-4, 3e-4, 1e-2)
    assert n > 20
    o1, o2 = spn(x.detach(), None), spn(x.detach(), None)
    assert torch.equal(o1, o2)


@pytest.mark.parametrize('shape', [(10, 10, 10, 10, 300), (6, 6, 3, 2, 65), (12, 14, 16, 16, 33), (9, 7, 5, 9, 130)])
def test_object_spn_operator_of_any_shape_against_the_oracle(shape):
    """Glimpse sizes whose leaves have different lengths (9 x 7 = 63 dimensions), the widest vectors the kernels take (16 / 16) and the
    default shape through the general operator, against the float64 oracle; with and without marginalisation."""
    from stove_amd.spn import probabilistic_models as prob
    pw, ph, G, S, n = shape
    kw = dict(patch_width=pw, patch_height=ph, obj_spn_num_gauss=G, obj_spn_num_sums=S)
    c, structs, params = oracle_setup(torch.float64, **kw)
    spn = fill_analytic(prob._get_obj_spn(_cfg(**kw), 42), 'sup.obj_spn.').to(DEV)
    if (pw, ph, G, S) == (10, 10, 10, 10):
        assert spn._kind == 'obj'                        # the default shape keeps the tuned kernels ...
        spn._force_general_plan()                        # ... and also runs through the general operator here
    assert spn._kind == 'obj_any'
    d = pw * ph
    gen = torch.Generator().manual_seed(pw * 100 + ph)
    x64 = torch.rand(n, d, generator=gen, dtype=torch.float64)
    m64 = torch.rand(n, d, generator=gen, dtype=torch.float64) * 1.4 - 0.2
    w64 = torch.linspace(0.5, 1.5, n, dtype=torch.float64)
    xo, mo = x64.clone().requires_grad_(), m64.clone().requires_grad_()
    ref = O.spn_forward(structs['obj'], params, 'sup.obj_spn.', xo, mo, G, S, c.obj_min_var, c.obj_max_var)
    (ref[:, 0] * w64).sum().backward()
    xd, md = x64.float().to(DEV).requires_grad_(), m64.float().to(DEV).requires_grad_()
    out = spn(xd, md)
    check('objany.oracle_fwd', err(out, ref), 5e-6)
    (out[:, 0] * w64.float().to(DEV)).sum().backward()
    check_grad('objany.oracle_dx', xd.grad, xo.grad, 3e-4, 3e-4, 5e-3)
    check_grad('objany.oracle_dmarg', md.grad, mo.grad, 3e-4, 3e-4, 5e-3)
    for k, p in spn.named_parameters():
        r = params['sup.obj_spn.' + k].grad
        if r is not None and float(r.abs().max()) > 0:
            check_grad('objany.oracle_grad', p.grad, r, 3e-4, 3.5e-4, 1e-2)
    check('objany.oracle_nomarg', err(spn(xd.detach(), None), O.spn_forward(structs['obj'], params, 'sup.obj_spn.', x64, None, G, S, c.obj_min_var, c.obj_max_var)), 5e-6)
    # MPE reconstruction (Supair.spn_mpe's inner step): the oracle's argmax walk per sample; a fp32 / fp64 tie may flip a branch
    with torch.no_grad():
        rec = spn.mpe(xd.detach()[:24]).cpu().double()
        _, kept = O.spn_forward(structs['obj'], params, 'sup.obj_spn.', x64[:24], None, G, S, c.obj_min_var, c.obj_max_var, child_values=True)
    want = np.stack([np.clip(O.spn_decode(structs['obj'], params, 'sup.obj_spn.', {k: np.argmax(v[j].detach().numpy(), 0) for k, v in kept.items()}), 0, 1)
                     for j in range(24)])
    same = (np.abs(rec.numpy() - want).max(1) < 1e-5).mean()
    assert same >= 0.9, same


def test_likelihood_and_training_with_an_object_spn_of_another_shape(tmp_path):
    """Supair.likelihood at 8 x 12 glimpses / 7 Gaussians / 5 sums against the reference's numbers (g14), then a few optimiser
    steps of the whole model with that shape (eager and replayed): finite, and the ELBO moves."""
    from stove_amd.video_prediction.supair import Supair
    g = load_golden('g14_likelihood_8x12_f64')
    sup = fill_analytic(Supair(_cfg(**SHAPE_8x12)), 'sup.').to(DEV)
    sup.step_counter = 0
    x = t_(g['x']).float().to(DEV)
    z = t_(g['z']).float().to(DEV).requires_grad_()
    lp, prop = sup.likelihood(x, z)
    check('shape.log_p', err(lp, g['log_p']), 5e-6)
    for k in ('bg', 'patch', 'overlap'):
        check('shape.part_' + k, abs(float(prop[k]) - float(g[k])) / (abs(float(g[k])) + 1e-9), 5e-6)
    (lp * t_(g['w']).float().to(DEV)).sum().backward()
    check('shape.dz', err(z.grad, g['gz']), 3e-4)
    params = dict(sup.named_parameters())
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            check('shape.grad_norm', abs(float(params[k[3:]].grad.norm()) - float(v)) / (float(v) + 1e-9), 3e-4)
            n += 1
        elif k.startswith('g_'):
            check('shape.grad_tensor', err(params[k[2:]].grad, v), 3e-4)
    assert n > 10
    # the whole model trains with it
    from stove_amd.envs import envs
    from stove_amd.video_prediction.stove import Stove
    torch.manual_seed(3)
    st = Stove(_cfg(**SHAPE_8x12)).to(DEV)
    opt = torch.optim.Adam(st.parameters(), lr=2e-3)
    xs = torch.from_numpy(envs.synth_sequences('billiards', 8, 8, seed0=2)['X']).to(DEV)
    vals = []
    for _ in range(4):
        opt.zero_grad()
        elbo, _, _ = st(xs, 1)
        (-elbo).backward()
        opt.step()
        vals.append(float(elbo))
    assert all(np.isfinite(v) for v in vals) and vals[-1] != vals[0]


@pytest.mark.parametrize('name', ['both', 'bg', 'obj'])
def test_fixed_gaussian_debug_models_against_the_reference(name):
    """config.debug_bg_model / debug_obj_spn (SimpleBG / SimpleObj of the reference, probabilistic_models.py:42-90) on the HIP row
    kernels (csrc/spn_obj_generic.hip gauss_ll_*), through Supair.likelihood, against the reference's numbers (g15)."""
    from stove_amd.video_prediction.supair import Supair
    kw = {'both': dict(debug_bg_model=True, debug_obj_spn=True), 'bg': dict(debug_bg_model=True), 'obj': dict(debug_obj_spn=True)}[name]
    g = load_golden(f'g15_likelihood_simple_{name}_f64')
    sup = fill_analytic(Supair(_cfg(**kw)), 'sup.').to(DEV)
    sup.step_counter = 0
    x = t_(g['x']).float().to(DEV)
    z = t_(g['z']).float().to(DEV).requires_grad_()
    lp, prop = sup.likelihood(x, z)
    check('simple.log_p', err(lp, g['log_p']), 5e-6)
    for k in ('bg', 'patch', 'overlap'):
        check('simple.part_' + k, abs(float(prop[k]) - float(g[k])) / (abs(float(g[k])) + 1e-9), 5e-6)
    (lp * t_(g['w']).float().to(DEV)).sum().backward()
    check('simple.dz', err(z.grad, g['gz']), 3e-4)
    params = dict(sup.named_parameters())
    for k, v in g.items():
        if k.startswith('gn_') and float(v) > 0:
            check('simple.grad_norm', abs(float(params[k[3:]].grad.norm()) - float(v)) / (float(v) + 1e-9), 3e-4)


@pytest.mark.parametrize('n_obj', [3, 6])
def test_recognition_network_in_row_chunks_equals_the_unchunked_chain(n_obj):
    """ops._encoder_lstm_fwd_chunked (reference encoder.py:43-57): the recognition network's forward chain over two row chunks on
    two streams: bit-identical to the unchunked chain (row-wise the same kernels), codes and every gradient."""
    from stove_amd import ops
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.encoder import RnnStates
    from test_gpu_dynamics import make_cfg
    n = 8192
    g = torch.Generator().manual_seed(21)
    x = torch.rand(n, 1, 32, 32, generator=g).to(DEV)
    w = torch.randn(n, n_obj, 8, generator=g).to(DEV)
    saved = ops.ENC_CHUNKS
    res = {}
    try:
        for mode, cf in (('plain', 1), ('chunked', 2)):
            ops.ENC_CHUNKS = cf
            torch.manual_seed(5)
            enc = RnnStates(make_cfg(num_obj=n_obj)).to(DEV)
            arena = ParamArena(enc)
            assert (ops._enc_chunks(n, 1024, 256) is not None) == (cf > 1)
            arena.zero()
            out = enc(x)
            (out * w).sum().backward()
            torch.cuda.synchronize()
            res[mode] = (out.detach().clone(), {k: p.grad.detach().clone() for k, p in enc.named_parameters()})
    finally:
        ops.ENC_CHUNKS = saved
    assert torch.equal(res['chunked'][0], res['plain'][0])
    for k, gp in res['plain'][1].items():
        assert torch.equal(res['chunked'][1][k], gp), k


def test_profile_report_covered_time_of_overlapping_launches():
    """stove_profile_report's fourth column (bench.py roofline.frac_covered): the time a kernel's launches COVER -- equal to their
    summed time for launches in a row on one stream, less than it when two streams run them side by side.  (Whether two streams
    do run side by side depends on the hardware queues the runtime maps them to: several second streams are tried.)"""
    from stove_amd import _lib, ops
    lib = _lib.load()
    n = 25600
    x = torch.rand(n, 1024, device=DEV)
    w = torch.randn(1024, 1024, device=DEV) * 0.03
    outs = [torch.empty(n // 2, 1024, device=DEV) for _ in range(2)]

    def half(i):
        _lib.check(lib.stove_gemm_bf16(_lib.ptr(x[i * (n // 2):]), _lib.ptr(w), None, None, _lib.ptr(outs[i]), n // 2, 1024, 1024, 1024, 1024, 1024,
                                       0, 0, 2, 1, 3, None, _lib.stream()), 'stove_gemm_bf16')

    def measure(side):
        half(0); half(1)
        torch.cuda.synchronize()
        lib.stove_profile_enable(1)
        try:
            if side is not None:
                side.wait_stream(torch.cuda.current_stream(DEV))
            half(0)
            if side is not None:
                with torch.cuda.stream(side):
                    half(1)
                torch.cuda.current_stream(DEV).wait_stream(side)
            else:
                half(1)
            torch.cuda.synchronize()
            rep = _lib.profile_report(wall=True)
        finally:
            lib.stove_profile_enable(0)
        total, count, covered = rep['gemm_bf16_k']
        assert count == 2 and 0.0 < covered <= 1.02 * total, (total, covered)
        return total, covered

    total, covered = measure(None)
    assert abs(covered - total) <= 0.02 * total, (total, covered)          # in a row: the spans do not overlap
    ratios = []
    for _ in range(8):
        total, covered = measure(torch.cuda.Stream(device=DEV))
        ratios.append(covered / total)
        if ratios[-1] < 0.9:
            break
    if min(ratios) >= 0.9:
        pytest.skip('no second stream ran beside the first one (streams share a hardware queue): %s' % ratios)
