"""Pin the CPU oracle (oracle/stove_oracle.py) against outputs of the reference itself.

The fixtures under tests/golden/ were produced by oracle/make_goldens.py importing
/root/reference; nothing here reads the reference at run time.
"""
import json
import os

import numpy as np
import pytest
import torch

import stove_oracle as O
from helpers import GOLDEN, load_golden, oracle_setup, rel_err, source_index, t_

F64 = torch.float64


# ---------------------------------------------------------------- G1 structure
@pytest.mark.parametrize('seed', [7, 42])
def test_spn_structure_matches_reference(seed):
    with open(os.path.join(GOLDEN, 'g1_spn_structure.json')) as f:
        gold = json.load(f)
    c = O.default_config(random_seed=seed)
    for kind, st in (('obj', O.obj_spn_structure(c)), ('bg', O.bg_spn_structure(c))):
        g = gold[f'{kind}_{seed}']
        assert list(st['root']) == g['root']
        assert len(st['layers']) == len(g['layers'])
        for mine, ref in zip(st['layers'], g['layers']):
            assert [list(map(int, m)) for m in mine] == ref


# ---------------------------------------------------------------- G2 RatSpn.forward
@pytest.mark.parametrize('kind', ['obj', 'bg'])
@pytest.mark.parametrize('tag,dtype,tol', [('f64', torch.float64, 1e-10), ('f32', torch.float32, 2e-5)])
def test_ratspn_forward_backward(kind, tag, dtype, tol):
    g = load_golden(f'g2_ratspn_{kind}_{tag}')
    c, structs, params = oracle_setup(dtype)
    x = t_(g['x'], dtype).requires_grad_()
    m = t_(g['marg'], dtype).requires_grad_()
    if kind == 'obj':
        args = (c.obj_spn_num_gauss, c.obj_spn_num_sums, c.obj_min_var, c.obj_max_var)
    else:
        args = (6, 3, c.bg_min_var, c.bg_max_var)
    out = O.spn_forward(structs[kind], params, f'sup.{kind}_spn.', x, m, *args)
    assert rel_err(out.detach(), g['out']) < tol
    out_nm = O.spn_forward(structs[kind], params, f'sup.{kind}_spn.', x.detach(), None, *args)
    assert rel_err(out_nm.detach(), g['out_nomarg']) < tol
    (out[:, 0] * t_(g['wsum'], dtype)).sum().backward()
    assert rel_err(x.grad, g['gx']) < tol * 10
    assert rel_err(m.grad, g['gmarg']) < tol * 10
    n_checked = 0
    for k, v in g.items():
        if k.startswith('g_'):
            name = f'sup.{kind}_spn.' + k[2:]
            assert rel_err(params[name].grad, v) < tol * 50, name
            n_checked += 1
    assert n_checked > 10


# ---------------------------------------------------------------- G3 masks / glimpses
@pytest.mark.parametrize('n_obj', [3, 6])
def test_masks_and_glimpses(n_obj):
    g = load_golden(f'g3_scene_n{n_obj}')
    c = O.default_config(num_obj=n_obj)
    z = t_(g['z']).requires_grad_()
    x = t_(g['x'])
    mp, bg, ov = O.masks_from_z(c, z)
    pat = O.glimpses(c, x, z.flatten(0, 1))
    for a, k in ((mp, 'marg_patch'), (bg, 'bg_mask'), (ov, 'overlap'), (pat, 'patches')):
        assert rel_err(a.detach(), g[k]) < 1e-12, k
    ((mp * t_(g['wm'])).sum() + (bg * t_(g['wb'])).sum() + (ov * t_(g['wo'])).sum()
     + (pat * t_(g['wp'])).sum()).backward()
    assert rel_err(z.grad, g['gz']) < 1e-10


# ---------------------------------------------------------------- G4 likelihood
# weight regimes (tests/golden/analytic_weights.py): the round-1 fixtures ('analytic', fp64 + fp32) and round 5's 'init' / 'stress' (fp64)
REGIME_TAGS = [('analytic', 'f64', torch.float64), ('analytic', 'f32', torch.float32), ('init', 'f64', torch.float64), ('stress', 'f64', torch.float64)]


def _gname(stem, regime, tag):
    return f'{stem}_{tag}' if regime == 'analytic' else f'{stem}_{regime}_{tag}'


@pytest.mark.parametrize('n_obj,extra', [(3, {}), (6, {'overlap_beta': 100.0, 'max_obj_scale': 0.22})])
@pytest.mark.parametrize('regime,tag,dtype', REGIME_TAGS)
def test_scene_likelihood(n_obj, extra, regime, tag, dtype):
    tol = 1e-10 if dtype == torch.float64 else 2e-5
    g = load_golden(_gname(f'g4_likelihood_n{n_obj}', regime, tag))
    c, structs, params = oracle_setup(dtype, regime=regime, num_obj=n_obj, **extra)
    x = t_(g['x'], dtype)
    z = t_(g['z'], dtype).requires_grad_()
    lp, bg, pl, ol = O.scene_likelihood(c, params, structs, x, z, parts=True)
    assert rel_err(lp.detach(), g['log_p']) < tol
    assert abs(float(bg.mean().detach()) - float(g['bg'])) < tol * abs(float(g['bg'])) + 1e-9
    assert abs(float(pl.mean().detach()) - float(g['patch'])) < tol * abs(float(g['patch'])) + 1e-9
    assert abs(float(ol.mean().detach()) - float(g['overlap'])) < tol * abs(float(g['overlap'])) + 1e-9
    (lp * t_(g['w'], dtype)).sum().backward()
    assert rel_err(z.grad, g['gz']) < tol * 100
    for k, v in g.items():
        if k.startswith('g_'):
            assert rel_err(params['sup.' + k[2:]].grad, v) < tol * 100, k


# ---------------------------------------------------------------- G5 dynamics
@pytest.mark.parametrize('name,cfg', [
    ('plain3', dict(num_obj=3)), ('plain6', dict(num_obj=6)),
    ('ac3', dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True)),
    ('lim4', dict(num_obj=3))])
@pytest.mark.parametrize('regime,tag,dtype', REGIME_TAGS)
def test_dynamics(name, cfg, regime, tag, dtype):
    tol = 1e-10 if dtype == torch.float64 else 1e-4
    g = load_golden(_gname(f'g5_dynamics_{name}', regime, tag))
    c, structs, params = oracle_setup(dtype, regime=regime, **cfg)
    s = t_(g['s'], dtype).requires_grad_()
    act = t_(g['actions'], dtype) if 'actions' in g else None
    app = t_(g['app'], dtype).requires_grad_() if 'app' in g else None
    res, rew = O.dynamics_forward(c, params, s, act, app, lim_enc=int(g['lim_enc']))
    assert rel_err(res.detach(), g['result']) < tol
    loss = (res * t_(g['w'], dtype)).sum()
    if c.action_conditioned:
        assert rel_err(rew.detach(), g['reward']) < tol
        loss = loss + (rew * torch.linspace(1, 2, s.shape[0], dtype=dtype).view(-1, 1)).sum()
    loss.backward()
    assert rel_err(s.grad, g['gs']) < tol * 10
    if app is not None:
        assert rel_err(app.grad, g['gapp']) < tol * 10
    for k, v in g.items():
        if k.startswith('g_'):
            assert rel_err(params['dyn.' + k[2:]].grad, v) < tol * 10, k


# ---------------------------------------------------------------- G6 matchers / smoothing
def test_match_3only():
    g = load_golden('g6_match_3only')
    c = O.default_config()
    zm, zs, _ = O.match_3only(c, t_(g['z']), t_(g['zstd']), None)
    assert np.array_equal(zm.numpy(), g['z_matched']) or rel_err(zm, g['z_matched']) < 1e-15
    assert rel_err(zs, g['zstd_matched']) < 1e-15
    assert np.array_equal(source_index(g['zstd'], zs, 1e-12).numpy(), g['idx'])          # the permutation itself, exact
    c.debug_match_appearance = True
    zm, zs, am = O.match_3only(c, t_(g['z']), t_(g['zstd']), t_(g['app']))
    assert np.array_equal(source_index(g['zstd'], zs, 1e-12).numpy(), g['idx_app'])
    assert rel_err(zm, g['z_matched_app']) < 1e-15
    assert rel_err(zs, g['zstd_matched_app']) < 1e-15
    assert rel_err(am, g['app_matched']) < 1e-15


def test_match_greedy():
    g = load_golden('g6_match_greedy')
    c = O.default_config(num_obj=6, debug_match_objects='greedy')
    zm, zs, _ = O.match_greedy(c, t_(g['z']), t_(g['zstd']), None)
    assert rel_err(zm, g['z_matched']) < 1e-15
    assert rel_err(zs, g['zstd_matched']) < 1e-15
    assert np.array_equal(source_index(g['zstd'], zs, 1e-12).numpy(), g['idx'])


def test_fix_supair():
    g = load_golden('g6_fix_supair')
    a, b = O.fix_supair(t_(g['z']), t_(g['zstd']))
    assert rel_err(a, g['z_fixed']) < 1e-15
    assert rel_err(b, g['zstd_fixed']) < 1e-15
    assert np.abs(g['z_fixed'] - g['z']).max() > 0.05      # the fixture does exercise the branch


# ---------------------------------------------------------------- G10 units
def test_units():
    g = load_golden('g10_units')
    c = O.default_config()
    m, s = O.constrain_zp(c, t_(g['zp']))
    assert rel_err(m, g['zp_mean']) < 1e-14 and rel_err(s, g['zp_std']) < 1e-14
    mc, sc = O.constrain_z_dyn(c, t_(g['zd']), t_(g['zds']))
    assert rel_err(mc, g['zd_c']) < 1e-14 and rel_err(sc, g['zds_c']) < 1e-14
    assert rel_err(O.v_from_state(t_(g['zsup'])), g['v_full']) < 1e-14
    assert rel_err(O.v_std_from_pos(t_(g['zsups'])), g['vstd_full']) < 1e-14
    assert rel_err(O.bw_transform(t_(g['xc'])), g['bw']) < 1e-15
    tstd = torch.tensor(O.transition_std(c), dtype=F64).view(1, 1, -1)
    tl = O.normal_log_prob(t_(g['fs_z'])[..., 2:], t_(g['zdyn']), tstd)
    assert rel_err(tl, g['translik']) < 1e-6        # the reference's std tensor is float32 (dynamics.py:119)


# ---------------------------------------------------------------- G7/G8 full forward + rollout
CASES = {
    'n3': dict(num_obj=3),
    'n6': dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22),
    'ac3': dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True),
    'grav3': dict(num_obj=3),          # BASELINE.json configs[2]: gravity frames (envs.py:841-844 at res 32)
}


@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('regime,tag,dtype', REGIME_TAGS)
def test_stove_forward_and_rollout(name, regime, tag, dtype):
    _full_model_against(load_golden(_gname(f'g7_stove_{name}', regime, tag)), name, regime, dtype)


@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('regime', ['analytic', 'init', 'stress'])
def test_stove_forward_full_length(name, regime):
    """Round 6 (g17): B = 2, T = 100 -- the 98 dependent steps of the inference recursion every BASELINE config runs
    (stove.py:696-713), backward through all of them, then a 92-step rollout (stove.py:823-846), for the four workloads."""
    g = load_golden(_gname(f'g17_stove_T100_{name}', regime, 'f64'))
    assert g['x'].shape[:2] == (2, 100) and g['p_z'].shape[1] == 98 and g['roll_z'].shape[1] == 92
    _full_model_against(g, name, regime, torch.float64)


ABLATIONS = {          # full_state ablations (stove.py:140-160): g19
    'novel': dict(num_obj=3, debug_no_velocity=True),
    'nolat': dict(num_obj=3, debug_no_latents=True),
    'noreuse': dict(num_obj=3, debug_no_reuse=True),
}


@pytest.mark.parametrize('name', list(ABLATIONS))
def test_stove_forward_ablations(name):
    g = load_golden(f'g19_stove_{name}_f64')
    assert g['eps_steps'].shape[-1] == (6 if name == 'nolat' else 18)
    _full_model_against(g, name, 'analytic', torch.float64, ABLATIONS[name])
    if name == 'noreuse':          # the reference's full_state overwrites what debug_no_reuse sets (stove.py:151-163): the default model
        c, structs, params = oracle_setup(torch.float64, num_obj=3)
        eps = {'latent': t_(g['eps_lat']), 'std': t_(g['eps_std']), 'steps': [t_(e) for e in g['eps_steps']]}
        elbo, _ = O.stove_forward(c, params, structs, t_(g['x']), eps)
        assert abs(float(elbo) - float(g['elbo'])) < 1e-9 * abs(float(g['elbo']))


def _full_model_against(g, name, regime, dtype, cfg=None):
    tol = 1e-9 if dtype == torch.float64 else 1e-4
    cfg = CASES[name] if cfg is None else cfg
    c, structs, params = oracle_setup(dtype, regime=regime, **cfg)
    x = t_(g['x'], dtype)
    eps = {'latent': t_(g['eps_lat'], dtype), 'std': t_(g['eps_std'], dtype),
           'steps': [t_(e, dtype) for e in g['eps_steps']]}
    actions = t_(g['actions'], dtype) if 'actions' in g else None
    elbo, rewards, info = O.stove_forward(c, params, structs, x, eps, actions, detail=True)
    assert abs(float(elbo) - float(g['elbo'])) < tol * abs(float(g['elbo']))
    for k in ('z', 'z_dyn', 'z_sup', 'z_std', 'z_sup_std', 'log_q', 'translik'):
        assert rel_err(info[k].detach(), g['p_' + k]) < tol * 10, k
    assert rel_err(info['z_dyn_std'].detach(), g['p_z_dyn_std'][2:]) < tol * 10
    loss = -elbo
    if actions is not None:
        assert rel_err(rewards.detach(), g['rewards']) < tol * 10
        loss = loss + 3.0 * (rewards ** 2).sum()
    loss.backward()
    gtol = tol * 1e3 if dtype == torch.float64 else 5e-3
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            p = params[k[3:]]
            assert p.grad is not None, k
            assert abs(float(p.grad.norm()) - float(v)) <= gtol * float(v) + 1e-12, k
            n += 1
        elif k.startswith('g_'):
            assert rel_err(params[k[2:]].grad, v) < gtol, k
    assert n > 50
    with torch.no_grad():
        z_last = info['z'][:, -1]
        fut = actions[:, :5] if actions is not None else None
        app = info['obj_appearances'][:, -1] if actions is not None else None
        zp, rp = O.rollout(c, params, z_last, g['roll_z'].shape[1], fut, app)
    assert rel_err(zp, g['roll_z']) < tol * 100
    if actions is not None:
        assert rel_err(rp, g['roll_rewards']) < tol * 100
    if 'eps_roll' in g:                 # sampling rollout under the reference's draws
        with torch.no_grad():
            zs, lq, _ = O.rollout(c, params, z_last, g['roll_s_z'].shape[1], eps=[t_(e, dtype) for e in g['eps_roll']])
        assert rel_err(zs, g['roll_s_z']) < tol * 100
        assert rel_err(lq, g['roll_s_logq']) < tol * 100


def test_match_volatile():
    g = load_golden('g6_match_volatile')
    c = O.default_config(debug_match_objects='volatile')
    zm, zs, _ = O.match_volatile(c, t_(g['z']), t_(g['zstd']), None)
    assert rel_err(zm, g['z_matched']) < 1e-14 and rel_err(zs, g['zstd_matched']) < 1e-14
    assert (g['perm'].sum(-1) == 1).all()                                               # one source per slot: a gather
    assert np.array_equal(source_index(g['zstd'], zs, 1e-12).numpy(), g['perm'].argmax(-1))


def test_supair_only_elbo():
    g = load_golden('g11_supair_only_f64')
    c, structs, params = oracle_setup(torch.float64)
    elbo, z, log_q = O.supair_forward(c, params, structs, O.bw_transform(t_(g['x'])), t_(g['eps']))
    # the fixture stores the frames as float32, the reference ran on their float64 originals
    assert abs(float(elbo) - float(g['elbo'])) < 1e-6 * abs(float(g['elbo']))
    assert rel_err(z.detach(), g['z']) < 1e-6
    assert abs(float(log_q) - float(g['log_q'])) < 1e-6 * abs(float(g['log_q']))
    (-elbo).backward()
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            assert abs(float(params[k[3:]].grad.norm()) - float(v)) <= 1e-4 * float(v) + 1e-12, k
            n += 1
    assert n > 60


# ---------------------------------------------------------------- G12 MPE rendering
@pytest.mark.parametrize('tag,dtype,tol', [('f64', torch.float64, 1e-12), ('f32', torch.float32, 1e-6)])
def test_reconstruct_from_z(tag, dtype, tol):
    g = load_golden(f'g12_reconstruct_{tag}')
    c, structs, params = oracle_setup(dtype, requires_grad=False)
    z, x = t_(g['z'], dtype), t_(g['x'], dtype)
    assert np.abs(O.spn_max_activation(structs['bg'], params, 'sup.bg_spn.', dtype).numpy() - g['bg_max']).max() < tol
    assert np.abs(O.spn_max_activation(structs['obj'], params, 'sup.obj_spn.', dtype).numpy() - g['obj_max']).max() < tol
    mpe = O.spn_mpe(c, params, structs, z.flatten(0, 1), x.flatten(0, 1))
    assert mpe.shape == g['mpe_patches'].shape
    assert np.abs(mpe.numpy() - g['mpe_patches']).max() < tol
    assert len(np.unique(g['mpe_patches'].reshape(-1, 100), axis=0)) > 3        # the walks do depend on the glimpse
    for key, kw in (('recon_max', {}), ('recon_mpe', dict(x=x, max_activation=False, single_image=False)),
                    ('recon_mpe_single', dict(x=x[:, 0], max_activation=False, single_image=True))):
        r = O.reconstruct_from_z(c, params, structs, z, **kw)
        assert r.shape == g[key].shape and r.dtype == dtype
        assert np.abs(r.numpy() - g[key]).max() < tol * 10, key


WIDE = {'res50': dict(width=50, height=50), 'ac32': dict(align_corners=True)}


@pytest.mark.parametrize('name', list(WIDE))
def test_likelihood_and_forward_beyond_the_32x32_contract(name):
    """g13: the reference on its stock 50 x 50 gravity frames (envs.py:841-844) and under the torch-1.0.1 sampling convention
    (align_corners=True), float64: Supair.likelihood with its gradients and the whole Stove.forward."""
    dtype, tol = torch.float64, 1e-9
    g = load_golden(f'g13_likelihood_{name}_f64')
    c, structs, params = oracle_setup(dtype, **WIDE[name])
    x, z, w = t_(g['x'], dtype), t_(g['z'], dtype).requires_grad_(), t_(g['w'], dtype)
    lp, bg, pl, ol = O.scene_likelihood(c, params, structs, x, z, parts=True)
    assert rel_err(lp.detach(), g['log_p']) < tol
    assert abs(float(bg.mean().detach()) - float(g['bg'])) < tol * abs(float(g['bg'])) + 1e-9
    (lp * w).sum().backward()
    assert rel_err(z.grad, g['gz']) < tol * 100
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            p = params['sup.' + k[3:]] if not k[3:].startswith('sup.') else params[k[3:]]
            assert abs(float(p.grad.norm()) - float(v)) <= 1e-6 * float(v) + 1e-12, k
            n += 1
    assert n > 10
    g = load_golden(f'g13_stove_{name}_f64')
    c, structs, params = oracle_setup(dtype, **WIDE[name])
    eps = {'latent': t_(g['eps_lat'], dtype), 'std': t_(g['eps_std'], dtype), 'steps': [t_(e, dtype) for e in g['eps_steps']]}
    elbo, _, info = O.stove_forward(c, params, structs, t_(g['x'], dtype), eps, None, detail=True)
    assert abs(float(elbo) - float(g['elbo'])) < tol * abs(float(g['elbo']))
    for k in ('z', 'z_dyn', 'z_sup', 'log_q', 'translik'):
        assert rel_err(info[k].detach(), g['p_' + k]) < tol * 10, k
    (-elbo).backward()
    n = 0
    for k, v in g.items():
        if k.startswith('gn_'):
            assert abs(float(params[k[3:]].grad.norm()) - float(v)) <= 1e-6 * float(v) + 1e-12, k
            n += 1
    assert n > 50
    with torch.no_grad():
        rec = O.reconstruct_from_z(c, params, structs, info['z'])
    assert rel_err(rec, g['recon']) < 1e-6


SHAPE_8x12 = dict(patch_width=8, patch_height=12, obj_spn_num_gauss=7, obj_spn_num_sums=5)


def test_object_spn_of_another_shape():
    """g14: the reference with 8 x 12 glimpses, 7 Gaussians per leaf and 5 sums per region (config.py:99-100, 119-120), float64:
    RatSpn.forward with out-of-range marginalisation and its gradients, then Supair.likelihood with that object SPN."""
    dtype, tol = torch.float64, 1e-9
    g = load_golden('g14_objspn_8x12_f64')
    c, structs, params = oracle_setup(dtype, **SHAPE_8x12)
    x, m, w = t_(g['x'], dtype).requires_grad_(), t_(g['marg'], dtype).requires_grad_(), t_(g['w'], dtype)
    out = O.spn_forward(structs['obj'], params, 'sup.obj_spn.', x, m, 7, 5, c.obj_min_var, c.obj_max_var)
    assert rel_err(out.detach(), g['out']) < tol
    (out[:, 0] * w).sum().backward()
    assert rel_err(x.grad, g['gx']) < tol * 100 and rel_err(m.grad, g['gmarg']) < tol * 100
    n = 0
    for k, v in g.items():
        if k.startswith('ogn_'):
            assert abs(float(params['sup.obj_spn.' + k[4:]].grad.norm()) - float(v)) <= 1e-6 * float(v) + 1e-12, k
            n += 1
    assert n > 20
    g = load_golden('g14_likelihood_8x12_f64')
    c, structs, params = oracle_setup(dtype, **SHAPE_8x12)
    x, z, w = t_(g['x'], dtype), t_(g['z'], dtype).requires_grad_(), t_(g['w'], dtype)
    lp, bg, pl, ol = O.scene_likelihood(c, params, structs, x, z, parts=True)
    assert rel_err(lp.detach(), g['log_p']) < tol
    (lp * w).sum().backward()
    assert rel_err(z.grad, g['gz']) < tol * 100


@pytest.mark.parametrize('name', ['both', 'bg', 'obj'])
def test_fixed_gaussian_debug_models(name):
    """g15: the reference with config.debug_bg_model / debug_obj_spn (SimpleBG / SimpleObj, probabilistic_models.py:42-90) in
    place of one or both SPNs, float64: Supair.likelihood, its parts and the gradient with respect to z."""
    dtype, tol = torch.float64, 1e-9
    kw = {'both': dict(debug_bg_model=True, debug_obj_spn=True), 'bg': dict(debug_bg_model=True), 'obj': dict(debug_obj_spn=True)}[name]
    g = load_golden(f'g15_likelihood_simple_{name}_f64')
    c, structs, params = oracle_setup(dtype, **kw)
    x, z, w = t_(g['x'], dtype), t_(g['z'], dtype).requires_grad_(), t_(g['w'], dtype)
    lp, bg, pl, ol = O.scene_likelihood(c, params, structs, x, z, parts=True)
    assert rel_err(lp.detach(), g['log_p']) < tol
    assert abs(float(bg.mean().detach()) - float(g['bg'])) < tol * abs(float(g['bg'])) + 1e-9
    assert abs(float(pl.mean().detach()) - float(g['patch'])) < tol * abs(float(g['patch'])) + 1e-9
    (lp * w).sum().backward()
    assert rel_err(z.grad, g['gz']) < tol * 100
