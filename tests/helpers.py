"""Shared helpers for the parity tests (oracle side)."""
import os

import numpy as np
import torch

import stove_oracle as O
from analytic_weights import analytic_tensor, fan_ins

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name + '.npz')))


def oracle_setup(dtype=torch.float64, requires_grad=True, regime='analytic', **cfg):
    """config + SPN structures + analytic parameters (weight regime `regime`) keyed by reference state-dict names."""
    c = O.default_config(**cfg)
    structs = O.build_structs(c)
    shapes = O.param_shapes(c, structs)
    fi = fan_ins({k: tuple(v) for k, v in shapes.items()})
    params = {}
    for k, shp in shapes.items():
        t = analytic_tensor(k, shp, dtype, regime, fi.get(k))
        params[k] = t.requires_grad_() if requires_grad else t
    return c, structs, params


def t_(a, dtype=torch.float64):
    return torch.from_numpy(np.asarray(a)).to(dtype)


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def reference_at_codes(gold, regime, cfg, codes):
    """The float64 oracle (pinned to the reference on this very fixture, tests/test_oracle_goldens.py) evaluated AT the codes another
    implementation's recognition network produced: a dict with the golden fixture's keys (elbo, p_*, gn_*, g_*, roll_*), everything
    the oracle does not restate (bg / patch / overlap means) copied from `gold`.  For the 'stress' regime, whose six-step recursion and
    its gradients amplify a 1e-5 difference of the codes to 1e-4 in z and 1e-2 in the dynamics' gradients: the recognition network is
    held to the fixture's codes by its own test, everything behind it to the reference at the same codes."""
    dtype = torch.float64
    c, structs, params = oracle_setup(dtype, regime=regime, **cfg)
    x = t_(gold['x'], dtype)
    eps = {'latent': t_(gold['eps_lat'], dtype), 'std': t_(gold['eps_std'], dtype), 'steps': [t_(e, dtype) for e in gold['eps_steps']]}
    actions = t_(gold['actions'], dtype) if 'actions' in gold else None
    elbo, rewards, info = O.stove_forward(c, params, structs, x, eps, actions, detail=True, code_values=codes.detach().double().cpu())
    loss = -elbo
    if actions is not None:
        loss = loss + 3.0 * (rewards ** 2).sum()
    loss.backward()
    out = dict(gold)
    out['elbo'] = elbo.detach().numpy()
    for k in ('z', 'z_dyn', 'z_sup', 'z_std', 'z_sup_std', 'log_q', 'translik'):
        out['p_' + k] = info[k].detach().numpy()
    nan2 = np.full(2, np.nan)
    out['p_z_dyn_std'] = np.concatenate([nan2, info['z_dyn_std'].detach().numpy()])
    if actions is not None:
        out['rewards'] = rewards.detach().numpy()
        out['p_obj_appearances'] = info['obj_appearances'].detach().numpy()
    for k in list(gold):
        if k.startswith('gn_'):
            out[k] = params[k[3:]].grad.norm().numpy()
        elif k.startswith('g_'):
            out[k] = params[k[2:]].grad.numpy()
    with torch.no_grad():
        z_last = info['z'].detach()[:, -1]
        fut = actions[:, :5] if actions is not None else None
        app = info['obj_appearances'].detach()[:, -1] if actions is not None else None
        zp, rp = O.rollout(c, params, z_last, gold['roll_z'].shape[1], fut, app)
        out['roll_z'] = zp.numpy()
        if actions is not None:
            out['roll_rewards'] = rp.numpy()
        if 'eps_roll' in gold:
            zs, lq, _ = O.rollout(c, params, z_last, gold['roll_s_z'].shape[1], eps=[t_(e, dtype) for e in gold['eps_roll']])
            out['roll_s_z'], out['roll_s_logq'] = zs.numpy(), lq.numpy()
    return out


def source_index(z, zm, tol=1e-6):
    """idx[b, t, slot] = k with zm[b, t, slot] == z[b, t, k] (to `tol`): the permutation a matcher applied, recovered from its
    output; every matched row must be exactly one source row (rows of the test tracks differ in their first two columns)."""
    z, zm = torch.as_tensor(z).double().cpu(), torch.as_tensor(zm).double().cpu()
    eq = ((zm.unsqueeze(3) - z.unsqueeze(2)).abs() < tol).all(-1)
    assert bool((eq.sum(-1) == 1).all()), 'a matched row is not exactly one source row'
    return eq.long().argmax(-1)
