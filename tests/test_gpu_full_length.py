"""Round 6: the inference recursion at the length every BASELINE config runs (T = 100: 98 dependent steps, reference
stove.py:696-713, backward through all of them, 92-step rollout stove.py:823-846) against reference-generated goldens (g17,
oracle/make_goldens.py g17: B = 2, T = 100, four workloads x three weight regimes) and against the pinned oracle on the box."""
import numpy as np
import pytest
import torch

import stove_oracle as O
from gpu_helpers import check, check_grad, err, fill_analytic
from helpers import load_golden, oracle_setup, t_
from test_gpu_dynamics import CASES, DEV, _golden_noise, full_model_against_golden, gname, make_cfg

pytestmark = pytest.mark.gpu

# (regime, fused, arena): the fused kernels with and without the flat parameter arena in every regime, and once the op-by-op chain
# (host time loop over stove_gnn_fwd/bwd, PyTorch state chain and ELBO assembly) as the third leg
T100_VARIANTS = [('analytic', True, True), ('analytic', True, False), ('analytic', False, False),
                 ('init', True, True), ('init', True, False), ('stress', True, True), ('stress', True, False)]


@pytest.mark.parametrize('name', list(CASES))
@pytest.mark.parametrize('regime,fused,arena', T100_VARIANTS)
def test_stove_forward_T100_vs_reference_golden(name, regime, fused, arena):
    gold = load_golden(gname(f'g17_stove_T100_{name}', regime))
    assert gold['x'].shape[:2] == (2, 100) and gold['p_z'].shape[1] == 98 and gold['roll_z'].shape[1] == 92
    full_model_against_golden(gold, name, regime, fused, arena, f'g17_{name}_{regime}', 'stoveT100')


PRESET = {'n3': 'billiards', 'grav3': 'gravity', 'n6': 'multibilliards', 'ac3': 'avoidance'}


@pytest.mark.parametrize('name', list(CASES))
def test_T100_golden_inside_the_full_size_batch(name):
    """BASELINE's own shape (256 sequences x 100 frames: 25 600 encoder rows in two row chunks, one workgroup per sequence in the
    recursion, the streamed per-step gradient workspace) with the two golden sequences as rows 0-1 and 77-78 of the batch: every
    per-sequence output of those rows must be the reference's -- sequences are independent, so a long-T or large-B
    indexing error anywhere in the path shows up here against the reference itself, not only against a property."""
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.stove import Stove
    gold = load_golden(gname(f'g17_stove_T100_{name}', 'analytic'))
    st = fill_analytic(Stove(make_cfg(**CASES[name])), '', 'analytic').to(DEV)
    ParamArena(st)
    B, T, N = 256, 100, CASES[name]['num_obj']
    # (the other 252 rows: 32 simulated sequences, repeated -- rows are independent, only their count and placement matter here)
    data = envs.synth_sequences(PRESET[name], 32, T, seed0=11)
    data = {k: np.concatenate([v] * (B // 32), 0) for k, v in data.items()}
    x = torch.from_numpy(data['X']).float()
    gx = t_(gold['x']).float()
    rows = [0, 1, 77, 78]
    g = torch.Generator().manual_seed(5)
    noise = {'latent': torch.randn(B, N, 12, generator=g), 'std': torch.randn(B, N, 12, generator=g), 'steps': torch.randn(B, T - 2, N, 18, generator=g)}
    gn = {'latent': t_(gold['eps_lat'])[..., 0].float(), 'std': t_(gold['eps_std'])[..., 0].float(),
          'steps': t_(gold['eps_steps']).float().permute(1, 0, 2, 3)}
    actions = torch.from_numpy(data['action']).float() if 'action' in data else None
    for j, r in enumerate(rows):
        x[r] = gx[j % 2]
        for k in noise:
            noise[k][r] = gn[k][j % 2]
        if actions is not None:
            actions[r] = t_(gold['actions']).float()[j % 2]
    noise = {k: v.to(DEV) for k, v in noise.items()}
    st.noise_fn = lambda kind, shape: noise[kind].reshape(shape)
    elbo, prop, rewards = st(x.to(DEV), 0, actions.to(DEV) if actions is not None else None)
    (-elbo).backward()
    assert np.isfinite(float(elbo))
    for k, bar in (('z', 3e-6), ('z_dyn', 3e-6), ('z_sup', 8e-6)):
        got = prop[k][rows]
        ref = np.concatenate([gold['p_' + k], gold['p_' + k]], 0)
        check('stoveT100.full_batch.' + k, err(got, ref), bar)
        assert torch.equal(got[0], got[2]) and torch.equal(got[1], got[3])          # same sequence, another row: same bits
    if rewards is not None and actions is not None:
        check('stoveT100.full_batch.rewards', err(rewards[rows], np.concatenate([gold['rewards'], gold['rewards']], 0)), 1e-6)
    with torch.no_grad():
        zp, _ = st.rollout(prop['z'][:, -1], num=92, actions=actions[:, :5].to(DEV) if actions is not None else None,
                           appearance=prop['obj_appearances'][:, -1] if actions is not None else None)
    check('stoveT100.full_batch.rollout_z', err(zp[rows], np.concatenate([gold['roll_z'], gold['roll_z']], 0)), 3e-6)


@pytest.mark.parametrize('name', ['n3', 'n6'])
def test_T100_against_the_oracle_on_the_box(name):
    """B = 8, T = 100 on inputs no fixture holds: the float64 oracle (pinned to the reference at this length by
    tests/test_oracle_goldens.py::test_stove_forward_full_length) on the host against the fused kernels and against the op-by-op
    chain -- ELBO, z, and EVERY parameter gradient (the fixtures store full tensors only up to 2 100 entries)."""
    from stove_amd.arena import ParamArena
    from stove_amd.envs import envs
    from stove_amd.video_prediction.stove import Stove
    B, T, N = 8, 100, CASES[name]['num_obj']
    x = torch.from_numpy(envs.synth_sequences(PRESET[name], B, T, seed0=900)['X']).float()
    g = torch.Generator().manual_seed(77)
    eps = O.draw_eps(B, N, T, generator=g, dtype=torch.float64)
    c, structs, params = oracle_setup(torch.float64, **CASES[name])
    elbo_o, _, info = O.stove_forward(c, params, structs, x.double(), eps, None, detail=True)
    (-elbo_o).backward()
    table = {'latent': eps['latent'][..., 0].float().to(DEV), 'std': eps['std'][..., 0].float().to(DEV),
             'steps': torch.stack(eps['steps'], 1).float().to(DEV)}
    for fused in (True, False):
        st = fill_analytic(Stove(make_cfg(fused_dynamics=fused, fused_state=fused, fused_elbo=fused, **CASES[name]))).to(DEV)
        if fused:
            ParamArena(st)
        st.noise_fn = lambda kind, shape: table[kind].reshape(shape)
        elbo, prop, _ = st(x.to(DEV), 0, None)
        (-elbo).backward()
        tag = 'stoveT100.oracle.' + ('fused' if fused else 'chain')
        check(tag + '.elbo_rel', abs(float(elbo) - float(elbo_o)) / abs(float(elbo_o)), 1.5e-6)
        for k in ('z', 'z_dyn', 'z_sup'):
            check(tag + '.' + k, err(prop[k], info[k].detach()), 8e-6 if k == 'z_sup' else 3e-6)
        n = 0
        for k, p in st.named_parameters():
            ref = params[k].grad
            if ref is None:
                continue
            check_grad(tag + '.grad', p.grad, ref, 3e-4, 3.5e-4, 4e-3)
            n += 1
        assert n > 100
