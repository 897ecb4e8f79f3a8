"""Generate golden fixtures by importing the reference (/root/reference) in THIS container.

TEST INFRASTRUCTURE.  Run once here (`python oracle/make_goldens.py`); the outputs
under tests/golden/ are committed, the reference's Python never travels.  Each
fixture holds inputs + the reference's outputs (and gradients) for one piece of
the hot path.  Model weights are NOT stored: they
are the analytic fill of oracle/analytic_weights.py applied through
`named_parameters()`.
"""
import json
import os
import sys
import types

sys.dont_write_bytecode = True
REF = os.environ.get('STOVE_REFERENCE', '/root/reference')
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), 'tests', 'golden')
sys.path.insert(0, REF)
sys.path.insert(0, HERE)

import numpy as np
import scipy
import torch

# ---- import-time stubs for packages the reference imports but the path never uses
for name in ['spriteworld', 'spriteworld.renderers', 'spriteworld.sprite', 'imageio', 'setproctitle']:
    if name not in sys.modules:
        sys.modules[name] = types.ModuleType(name)
sys.modules['spriteworld.sprite'].Sprite = object
sys.modules['spriteworld'].renderers = sys.modules['spriteworld.renderers']
sys.modules['setproctitle'].setproctitle = lambda *_: None
if not hasattr(scipy, 'rand'):
    scipy.rand = np.random.rand
    scipy.randn = np.random.randn

import warnings
warnings.filterwarnings('ignore')

from analytic_weights import analytic_tensor, fan_ins  # noqa: E402

from model.video_prediction.config import StoveConfig  # noqa: E402
from model.video_prediction.stove import Stove  # noqa: E402
from model.video_prediction.supair import Supair  # noqa: E402
from model.video_prediction.dynamics import Dynamics  # noqa: E402
from model.spn import probabilistic_models as prob  # noqa: E402
from model.spn import rat_torch  # noqa: E402
from model.envs import envs as ref_envs  # noqa: E402
import torch.distributions.normal as tdn  # noqa: E402


def ref_config(dtype=torch.float64, **kw):
    c = StoveConfig()
    c.num_obj, c.width, c.height = 3, 32, 32
    c.device = torch.device('cpu')
    c.dtype = dtype
    c.random_seed = 42
    c.action_conditioned = False
    c.action_space = None
    c.skip = 2
    c.r, c.coord_lim, c.num_frames = 1.2, 10, 100
    for k, v in kw.items():
        setattr(c, k, v)
    torch.set_default_dtype(dtype)
    return c


def fill(module, prefix='', regime='analytic'):
    with torch.no_grad():
        named = {prefix + name: p for name, p in module.named_parameters()}
        fi = fan_ins({k: tuple(p.shape) for k, p in named.items()})
        for name, p in named.items():
            p.copy_(analytic_tensor(name, p.shape, p.dtype, regime, fi.get(name)))


def np_(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def save(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **{k: np_(v) for k, v in arrays.items()})
    print('wrote', path, os.path.getsize(path) // 1024, 'KiB')


class EpsFeeder:
    """Replaces torch.distributions.normal._standard_normal by a replay of a fixed list."""

    def __init__(self, tensors):
        self.tensors = list(tensors)
        self.i = 0

    def __call__(self, shape, dtype, device):
        t = self.tensors[self.i]
        self.i += 1
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return t.to(dtype)


# ------------------------------------------------------------------ G1 structure
def spn_struct_json(spn):
    layers = []
    pos = {}
    for li, layer in enumerate(spn.vector_list):
        cur = []
        for i, vec in enumerate(layer):
            pos[id(vec)] = (li, i)
            if isinstance(vec, rat_torch.GaussVector):
                cur.append([int(s) for s in vec.scope])
            elif isinstance(vec, rat_torch.ProductVector):
                a, b = vec.inputs
                cur.append([*pos[id(a)], *pos[id(b)]])
            else:
                cur.append([pos[id(p)][1] for p in vec.inputs])
        layers.append(cur)
    return {'layers': layers, 'root': list(pos[id(spn.output_vector)])}


def g1_structures():
    out = {}
    for seed in (7, 42):
        c = ref_config(random_seed=seed)
        out[f'obj_{seed}'] = spn_struct_json(prob._get_obj_spn(c, seed))
        out[f'bg_{seed}'] = spn_struct_json(prob._get_bg_spn(c, seed))
    with open(os.path.join(OUT, 'g1_spn_structure.json'), 'w') as f:
        json.dump(out, f)
    print('wrote g1_spn_structure.json')


# ------------------------------------------------------------------ G2 RatSpn.forward
def g2_ratspn():
    for dtype, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        c = ref_config(dtype)
        g = torch.Generator().manual_seed(2)
        for kind in ('obj', 'bg'):
            spn = (prob._get_obj_spn if kind == 'obj' else prob._get_bg_spn)(c, 42)
            fill(spn, f'sup.{kind}_spn.')
            d = spn.num_dims
            x = torch.rand(8, d, generator=g, dtype=torch.float64).to(dtype).requires_grad_()
            m = torch.rand(8, d, generator=g, dtype=torch.float64).to(dtype)
            m[0] = 0.0
            m[1] = 1.0
            m[2, ::3] = 1.3       # out of range on purpose (clamped by the leaf)
            m[2, 1::3] = -0.2
            m[3, : d // 2] = 0.0
            m[3, d // 2:] = 1.0
            m.requires_grad_()
            out = spn.forward(x, m)
            wsum = torch.linspace(0.5, 1.5, 8, dtype=dtype)
            (out[:, 0] * wsum).sum().backward()
            grads = {f'g_{n}': p.grad for n, p in spn.named_parameters()}
            out_nomarg = spn.forward(x.detach(), None)
            save(f'g2_ratspn_{kind}_{tag}', x=x, marg=m, out=out, out_nomarg=out_nomarg,
                 gx=x.grad, gmarg=m.grad, wsum=wsum, **grads)


# ------------------------------------------------------------------ G3/G4 scene pieces
def crafted_z(n, n_obj, g, dtype):
    z = torch.zeros(n, n_obj, 4, dtype=torch.float64)
    z[..., 0] = 0.1 + 0.5 * torch.rand(n, n_obj, generator=g, dtype=torch.float64)
    z[..., 1] = z[..., 0] * (0.75 + 0.5 * torch.rand(n, n_obj, generator=g, dtype=torch.float64))
    z[..., 2:] = 1.8 * torch.rand(n, n_obj, 2, generator=g, dtype=torch.float64) - 0.9
    z[0, 1, 2:] = z[0, 0, 2:] + 0.05               # overlapping pair
    z[1, 0, 2:] = torch.tensor([0.95, -0.97])      # partly out of frame
    z[2, :, 0] = 0.1                               # tiny boxes
    z[2, :, 1] = 0.075
    if n > 3:
        z[3, :, 2:] = z[3, 0:1, 2:]                # all objects stacked
    return z.to(dtype)


def g3_masks_glimpses():
    for n_obj in (3, 6):
        c = ref_config(num_obj=n_obj)
        sup = Supair(c)
        g = torch.Generator().manual_seed(3)
        n = 5
        z = crafted_z(n, n_obj, g, c.dtype).requires_grad_()
        x = torch.rand(n, 1, 32, 32, generator=g, dtype=torch.float64)
        mp, bg, ov = sup.masks_from_z(z)
        pat = sup.patches_from_z(x, z.flatten(end_dim=1))
        wm = torch.rand(mp.shape, generator=g, dtype=torch.float64)
        wb = torch.rand(bg.shape, generator=g, dtype=torch.float64)
        wo = torch.rand(ov.shape, generator=g, dtype=torch.float64)
        wp = torch.rand(pat.shape, generator=g, dtype=torch.float64)
        ((mp * wm).sum() + (bg * wb).sum() + (ov * wo).sum() + (pat * wp).sum()).backward()
        save(f'g3_scene_n{n_obj}', z=z, x=x, marg_patch=mp, bg_mask=bg, overlap=ov, patches=pat,
             wm=wm, wb=wb, wo=wo, wp=wp, gz=z.grad)


def _rs(regime):
    """file-name infix of a weight regime (analytic_weights.REGIMES): the round-1 fixtures carry none"""
    return '' if regime == 'analytic' else '_' + regime


def g4_likelihood(regime='analytic', dtypes=((torch.float64, 'f64'), (torch.float32, 'f32'))):
    for dtype, tag in dtypes:
        for n_obj, extra in ((3, {}), (6, {'overlap_beta': 100.0, 'max_obj_scale': 0.22})):
            c = ref_config(dtype, num_obj=n_obj, **extra)
            c.debug = True
            sup = Supair(c)
            fill(sup, 'sup.', regime)
            g = torch.Generator().manual_seed(4)
            n, t = 2, 4
            x = (torch.rand(n, t, 1, 32, 32, generator=g, dtype=torch.float64) ** 3).to(dtype)
            z = crafted_z(n * t, n_obj, g, dtype).flatten(end_dim=1).requires_grad_()
            sup.step_counter = 0
            lp, prop = sup.likelihood(x, z)
            w = torch.linspace(0.5, 1.5, n * t, dtype=dtype)
            (lp * w).sum().backward()
            grads = {f'g_{k}': p.grad for k, p in sup.named_parameters() if p.grad is not None}
            save(f'g4_likelihood_n{n_obj}{_rs(regime)}_{tag}', x=x, z=z, log_p=lp, w=w, gz=z.grad,
                 bg=prop['bg'], patch=prop['patch'], overlap=prop['overlap'], **grads)


# ------------------------------------------------------------------ G5 dynamics
def g5_dynamics(regime='analytic', dtypes=((torch.float64, 'f64'), (torch.float32, 'f32'))):
    variants = [
        ('plain3', dict(num_obj=3), 2),
        ('plain6', dict(num_obj=6), 2),
        ('ac3', dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True), 2),
        ('lim4', dict(num_obj=3), 4),
    ]
    for dtype, tag in dtypes:
        for name, kw, lim in variants:
            c = ref_config(dtype, **kw)
            dyn = Dynamics(c)
            fill(dyn, 'dyn.', regime)
            g = torch.Generator().manual_seed(5)
            b, n_obj = 6, c.num_obj
            s = (torch.rand(b, n_obj, 16, generator=g, dtype=torch.float64) * 1.6 - 0.8).to(dtype).requires_grad_()
            act = app = None
            if c.action_conditioned:
                act = torch.zeros(b, 9, dtype=dtype)
                act[torch.arange(b), torch.arange(b) % 9] = 1.0
                app = torch.rand(b, n_obj, 3, generator=g, dtype=torch.float64).to(dtype).requires_grad_()
            res, rew = dyn.forward(s, 0, act, app, lim_enc=lim)
            w = torch.rand(res.shape, generator=g, dtype=torch.float64).to(dtype)
            loss = (res * w).sum()
            if c.action_conditioned:
                loss = loss + (rew * torch.linspace(1, 2, b, dtype=dtype).view(-1, 1)).sum()
            loss.backward()
            grads = {f'g_{k}': p.grad for k, p in dyn.named_parameters() if p.grad is not None}
            extra = {}
            if c.action_conditioned:
                extra = dict(actions=act, app=app, reward=rew, gapp=app.grad)
            save(f'g5_dynamics_{name}{_rs(regime)}_{tag}', s=s, result=res, w=w, gs=s.grad, lim_enc=np.array(lim), **extra, **grads)


# ------------------------------------------------------------------ G6 matchers + fix_supair, G10 units
def source_index(z, zm):
    """The permutation the reference's matcher applied, recovered from its output: idx[b, t, slot] = k with zm[b, t, slot] == z[b, t, k]
    to 1e-12 (the matchers gather rows or multiply by a 0/1 matrix; the reference maps positions to [0, 1] and back around it, stove.py:
    216-219 / 322-325, so values return one ulp off; the crafted rows differ by O(0.1))."""
    eq = ((zm.unsqueeze(3) - z.unsqueeze(2)).abs() < 1e-12).all(-1)   # (n, t, slot, k)
    assert bool((eq.sum(-1) == 1).all()), 'a matched row is not exactly one source row'
    return eq.long().argmax(-1)


def source_matrix(z, zm):
    """As source_index for the 'volatile' matcher, whose 0/1 matrix need not be a permutation: perm[b, t, slot, k] in {0, 1} with
    zm[b, t, slot] == sum_k perm * z[b, t, k] to 1e-12."""
    n, t, o, _ = z.shape
    perm = torch.zeros(n, t, o, o, dtype=torch.long)
    for b in range(n):
        for tt in range(t):
            for slot in range(o):
                hits = []
                for mask in range(1 << o):
                    acc = torch.zeros_like(z[b, tt, 0])
                    for k in range(o):
                        if mask >> k & 1:
                            acc = acc + z[b, tt, k]
                    if float((acc - zm[b, tt, slot]).abs().max()) < 1e-12:
                        hits.append(mask)
                assert len(hits) == 1, (b, tt, slot, hits)
                for k in range(o):
                    perm[b, tt, slot, k] = hits[0] >> k & 1
    return perm


def g6_matchers():
    g = torch.Generator().manual_seed(6)
    # 3_only: mostly smooth tracks + injected swaps + an ambiguous (fault) case
    c = ref_config()
    st = Stove(c)
    n, t = 6, 7
    base = torch.rand(n, 1, 3, 4, generator=g, dtype=torch.float64) * 1.6 - 0.8
    drift = torch.cumsum(0.03 * torch.randn(n, t, 3, 4, generator=g, dtype=torch.float64), 1)
    z = base + drift
    z[1, 3] = z[1, 3][[1, 0, 2]]
    z[2, 2] = z[2, 2][[2, 0, 1]]
    z[2, 5] = z[2, 5][[0, 2, 1]]
    z[3, 1:, 0, 2:] = z[3, 1:, 1, 2:] + 1e-3       # two objects nearly on top of each other -> non-unique argmin
    z[4, 4, :, 2:] = z[4, 3, 0:1, 2:]              # all current objects at prev object 0 -> fault repair
    zstd = torch.rand(n, t, 3, 4, generator=g, dtype=torch.float64) * 0.3
    zm, zsm, _ = st._3_only_match_objects(z.clone(), zstd.clone(), None)
    app = torch.rand(n, t, 3, 3, generator=g, dtype=torch.float64)
    c.debug_match_appearance = True
    zm_a, zsm_a, app_m = st._3_only_match_objects(z.clone(), zstd.clone(), app.clone())
    c.debug_match_appearance = False
    save('g6_match_3only', z=z, zstd=zstd, z_matched=zm, zstd_matched=zsm,
         app=app, z_matched_app=zm_a, zstd_matched_app=zsm_a, app_matched=app_m,
         idx=source_index(z, zm), idx_app=source_index(z, zm_a))

    c6 = ref_config(num_obj=6, debug_match_objects='greedy')
    st6 = Stove(c6)
    base = torch.rand(n, 1, 6, 4, generator=g, dtype=torch.float64) * 1.6 - 0.8
    z6 = base + torch.cumsum(0.03 * torch.randn(n, t, 6, 4, generator=g, dtype=torch.float64), 1)
    for b in range(n):
        for tt in range(1, t):
            perm = torch.randperm(6, generator=g)
            z6[b, tt] = z6[b, tt][perm]
    z6std = torch.rand(n, t, 6, 4, generator=g, dtype=torch.float64) * 0.3
    zm6, zsm6, _ = st6._greedy_match_objects(z6.clone(), z6std.clone(), None)
    save('g6_match_greedy', z=z6, zstd=z6std, z_matched=zm6, zstd_matched=zsm6, idx=source_index(z6, zm6))

    # fix_supair: glitches on scale dims (0,1) trigger, on position dims do not
    zf = base[:, :, :3] + torch.cumsum(0.01 * torch.randn(n, t, 3, 4, generator=g, dtype=torch.float64), 1)
    zf[0, 3, 1, 0] += 0.3
    zf[1, 2, 0, 1] -= 0.25
    zf[2, 4, 2, 2] += 0.5            # position glitch only: must not fire
    zf[3, 0, 0, 0] += 0.4            # t=0 never fires
    zf[3, t - 1, 1, 1] += 0.4        # t=T-1 never fires
    zf[4, 2, 1, 0] += 0.3
    zf[4, 3, 1, 0] += 0.3            # two in a row
    zfs = torch.rand(n, t, 3, 4, generator=g, dtype=torch.float64) * 0.3
    c = ref_config()
    st = Stove(c)
    a, b_ = st.fix_supair(zf.clone(), zfs.clone())
    save('g6_fix_supair', z=zf, zstd=zfs, z_fixed=a, zstd_fixed=b_)


def g10_units():
    c = ref_config()
    st = Stove(c)
    g = torch.Generator().manual_seed(10)
    zp = torch.randn(20, 8, generator=g, dtype=torch.float64) * 2
    m, s = st.sup.constrain_zp(zp)
    zd = torch.randn(4, 3, 16, generator=g, dtype=torch.float64) * 2
    zds = torch.randn(4, 3, 16, generator=g, dtype=torch.float64) * 2
    mc, sc = st.dyn.constrain_z_dyn(zd, zds)
    zsup = torch.rand(4, 5, 3, 4, generator=g, dtype=torch.float64)
    zsups = torch.rand(4, 5, 3, 4, generator=g, dtype=torch.float64) * 0.3
    vf = st.v_from_state(zsup)
    vs = st.v_std_from_pos(zsups)
    xc = torch.rand(2, 3, 3, 32, 32, generator=g, dtype=torch.float64)
    from model.utils.utils import bw_transform
    bw = bw_transform(xc)
    # full_state + transition_lik
    eps = torch.randn(4, 3, 18, generator=g, dtype=torch.float64)
    saved = tdn._standard_normal
    tdn._standard_normal = EpsFeeder([eps])
    zdyn = torch.rand(4, 3, 16, generator=g, dtype=torch.float64) - 0.5
    sdyn = torch.rand(4, 3, 16, generator=g, dtype=torch.float64) * 0.3 + 0.01
    zs6 = torch.rand(4, 3, 6, generator=g, dtype=torch.float64) - 0.5
    ss6 = torch.rand(4, 3, 6, generator=g, dtype=torch.float64) * 0.3 + 0.01
    z_s, log_q, mean, std = st.full_state(zdyn, sdyn, zs6, ss6)
    tdn._standard_normal = saved
    tl = st.transition_lik(zdyn, z_s[..., 2:])
    save('g10_units', zp=zp, zp_mean=m, zp_std=s, zd=zd, zds=zds, zd_c=mc, zds_c=sc,
         zsup=zsup, zsups=zsups, v_full=vf, vstd_full=vs, xc=xc, bw=bw,
         eps=eps, zdyn=zdyn, sdyn=sdyn, zs6=zs6, ss6=ss6, fs_z=z_s, fs_logq=log_q, fs_mean=mean, fs_std=std,
         translik=tl)


# ------------------------------------------------------------------ G0 environments
def g0_envs():
    arrs = {}
    for seed in range(4):
        env = ref_envs.BillardsEnv(n=3, r=1.2, m=1., hw=10, granularity=10, res=32, t=1.,
                                   friction_coefficient=0., seed=seed)
        imgs, states = [], []
        for _ in range(100):
            img, st_, _ = env.step()
            imgs.append(img.copy())
            states.append(st_.copy())
        arrs[f'bill3_img_{seed}'] = np.stack(imgs).astype(np.float32)
        arrs[f'bill3_state_{seed}'] = np.stack(states)
    for seed in range(2):
        env = ref_envs.BillardsEnv(n=6, r=1., m=1., hw=10, granularity=10, res=32, t=1.,
                                   friction_coefficient=0., seed=seed, use_colors=False)
        imgs, states = [], []
        for _ in range(30):
            img, st_, _ = env.step()
            imgs.append(img.copy())
            states.append(st_.copy())
        arrs[f'bill6_img_{seed}'] = np.stack(imgs).astype(np.float32)
        arrs[f'bill6_state_{seed}'] = np.stack(states)
    for seed in range(2):
        env = ref_envs.GravityEnv(n=3, r=2, m=4., hw=30, granularity=50, res=32, t=1.,
                                  init_v_factor=0.55, friction_coefficient=0., seed=seed)
        imgs, states = [], []
        for _ in range(30):
            img, st_, _ = env.step()
            imgs.append(img.copy())
            states.append(st_.copy())
        arrs[f'grav3_img_{seed}'] = np.stack(imgs).astype(np.float32)
        arrs[f'grav3_state_{seed}'] = np.stack(states)
    for seed in range(2):
        base = ref_envs.BillardsEnv(n=3, r=1., m=1., hw=10, granularity=50, res=32, t=1.,
                                    friction_coefficient=0., seed=seed)
        task = ref_envs.AvoidanceTask(base, 4, greyscale=False, action_force=0.6)
        p = np.random.uniform(0.2, 0.3)
        pol = ref_envs.MonteCarloActionPolicy(action_space=9, prob_change=p)
        imgs, states, acts, rews = [], [], [], []
        for _ in range(30):
            a = pol.next()
            img, st_, rew, done = task.step(a)
            imgs.append(np.asarray(img).copy())
            states.append(np.asarray(st_).copy())
            acts.append(a)
            rews.append(rew)
        arrs[f'avoid_img_{seed}'] = np.stack(imgs).astype(np.float32)
        arrs[f'avoid_state_{seed}'] = np.stack(states)
        arrs[f'avoid_action_{seed}'] = np.array(acts)
        arrs[f'avoid_reward_{seed}'] = np.array(rews, dtype=np.float64)
    save('g0_envs', **arrs)


def gravity_frames(n_seq, t_len):
    """cfg 3 of BASELINE.json: GravityEnv (reference envs.py:841-844) at res=32, one env per sequence, seed = i."""
    xs = []
    for seed in range(n_seq):
        env = ref_envs.GravityEnv(n=3, r=2, m=4., hw=30, granularity=50, res=32, t=1.,
                                  init_v_factor=0.55, friction_coefficient=0., seed=seed)
        xs.append(np.stack([env.step()[0].copy() for _ in range(t_len)]))
    return np.transpose(np.stack(xs), (0, 1, 4, 2, 3))


def billiards_frames(n_seq, t_len, n=3, r=1.2):
    xs = []
    for seed in range(n_seq):
        env = ref_envs.BillardsEnv(n=n, r=r, m=1., hw=10, granularity=10, res=32, t=1.,
                                   friction_coefficient=0., seed=seed, use_colors=None if n == 3 else False)
        imgs = [env.step()[0].copy() for _ in range(t_len)]
        xs.append(np.stack(imgs))
    x = np.stack(xs)                                   # (B,T,res,res,3)
    return np.transpose(x, (0, 1, 4, 2, 3))            # load_data.py:64


# ------------------------------------------------------------------ G7/G8 full model
def g7_g8_full(regime='analytic', dtypes=((torch.float64, 'f64'), (torch.float32, 'f32')), shape=None, prefix='g7_stove', roll_all=None, cases=None):
    """`shape` = (B, T) for every case instead of the short per-case ones; `roll_all`: rollout length for every case (g17);
    `cases`: another list of (name, config overrides, B, T) (g19)."""
    cases = cases or [
        ('n3', dict(num_obj=3), 4, 8),
        ('n6', dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22), 2, 6),
        ('ac3', dict(num_obj=3, action_conditioned=True, action_space=9, debug_core_appearance=True), 3, 6),
        ('grav3', dict(num_obj=3), 4, 8),
    ]
    if shape is not None:
        cases = [(n, kw, shape[0], shape[1]) for n, kw, _, _ in cases]
    only = os.environ.get('G7_ONLY')          # regenerate a subset without touching the other fixtures
    if only:
        cases = [c for c in cases if c[0] in only.split(',')]
    for dtype, tag in dtypes:
        for name, kw, B, T in cases:
            c = ref_config(dtype, **kw)
            c.debug = True
            N = c.num_obj
            st = Stove(c)
            fill(st, '', regime)
            if name == 'grav3':
                x = torch.from_numpy(gravity_frames(B, T)).to(dtype)
            else:
                x = torch.from_numpy(billiards_frames(B, T, n=N, r=1.2 if N == 3 else 1.0)).to(dtype)
            if regime != 'analytic':
                # the fixture stores float32 frames: run the reference ON the stored values.  (The round-1 'analytic' fixtures ran it
                # on the simulator's float64 frames and stored their float32 rounding: a 1e-8 input difference, 4e-10 in z there --
                # harmless at the analytic weights' gains, 3.5e-7 in z under the 'stress' weights.)
                x = x.to(torch.float32).to(dtype)
            g = torch.Generator().manual_seed(123)
            lat = torch.randn(B, N, 12, 1, generator=g, dtype=torch.float64)
            sd = torch.randn(B, N, 12, 1, generator=g, dtype=torch.float64)
            steps = [torch.randn(B, N, 6 if kw.get('debug_no_latents') else 18, generator=g, dtype=torch.float64) for _ in range(2, T)]
            actions = None
            if c.action_conditioned:
                actions = torch.zeros(B, T, 9, dtype=dtype)
                ai = torch.randint(0, 9, (B, T), generator=g)
                actions.scatter_(2, ai.unsqueeze(-1), 1.0)
            saved = tdn._standard_normal
            tdn._standard_normal = EpsFeeder([lat, sd] + steps)
            elbo, prop, rewards = st(x, 0, actions)
            tdn._standard_normal = saved
            loss = -elbo
            if c.action_conditioned:
                loss = loss + 3.0 * (rewards ** 2).sum()
            loss.backward()
            gnorm = {f'gn_{k}': p.grad.norm() for k, p in st.named_parameters() if p.grad is not None}
            small = {f'g_{k}': p.grad for k, p in st.named_parameters()
                     if p.grad is not None and p.numel() <= 2100}
            props = {f'p_{k}': v for k, v in prop.items() if v is not None and torch.is_tensor(v)}
            extra = {}
            if actions is not None:
                extra['actions'] = actions
                extra['rewards'] = rewards
            # rollout from the last inferred state (G8)
            with torch.no_grad():
                z_last = prop['z'][:, -1]
                fut = None
                app = None
                if c.action_conditioned:
                    fut = actions[:, :5]
                    app = prop['obj_appearances'][:, -1]
                z_pred, r_pred = st.rollout(z_last, num=roll_all or (92 if name == 'n3' else 12), actions=fut, appearance=app)
            extra['roll_z'] = z_pred
            if name in ('n3', 'grav3', 'novel', 'nolat', 'noreuse'):
                # sampling rollout (stove.py:833-838; the reference only runs it with return_std=True, see z_dyn_stds)
                eps_roll = [torch.randn(B, N, 16, generator=g, dtype=torch.float64) for _ in range(10)]
                tdn._standard_normal = EpsFeeder(eps_roll)
                with torch.no_grad():
                    zs, lq, _ = st.rollout(z_last, num=10, sample=True, return_std=True)
                tdn._standard_normal = saved
                extra.update(eps_roll=torch.stack(eps_roll, 0), roll_s_z=zs, roll_s_logq=lq)
            if c.action_conditioned:
                extra['roll_rewards'] = r_pred
            save(f'{prefix}_{name}{_rs(regime)}_{tag}', x=x.to(torch.float32), eps_lat=lat, eps_std=sd, eps_steps=torch.stack(steps, 0),
                 elbo=elbo, **props, **gnorm, **small, **extra)


# ------------------------------------------------------------------ G6b volatile matcher, G11 SuPAIR-only ELBO, G9 optimiser steps
def g6b_volatile():
    g = torch.Generator().manual_seed(66)
    c = ref_config(debug_match_objects='volatile')
    st = Stove(c)
    n, t = 6, 7
    base = torch.rand(n, 1, 3, 4, generator=g, dtype=torch.float64) * 1.6 - 0.8
    z = base + torch.cumsum(0.03 * torch.randn(n, t, 3, 4, generator=g, dtype=torch.float64), 1)
    z[1, 3] = z[1, 3][[1, 0, 2]]
    z[2, 2] = z[2, 2][[2, 0, 1]]
    z[4, 4, :, 2:] = z[4, 3, 0:1, 2:]              # all current objects collapse onto one previous slot
    zstd = torch.rand(n, t, 3, 4, generator=g, dtype=torch.float64) * 0.3
    zm, zsm, _ = st._volatile_match_objects(z.clone(), zstd.clone(), None)
    save('g6_match_volatile', z=z, zstd=zstd, z_matched=zm, zstd_matched=zsm, perm=source_matrix(z, zm))


def g11_supair_only():
    for dtype, tag in ((torch.float64, 'f64'),):
        c = ref_config(dtype)
        c.debug = True
        st = Stove(c)
        fill(st)
        B, T = 3, 4
        x = torch.from_numpy(billiards_frames(B, T)).to(dtype)
        g = torch.Generator().manual_seed(11)
        eps = torch.randn(B * T * 3, 4, generator=g, dtype=torch.float64)
        saved = tdn._standard_normal
        tdn._standard_normal = EpsFeeder([eps])
        elbo, prop, _ = st(x, 0, None, True)
        tdn._standard_normal = saved
        (-elbo).backward()
        gn = {f'gn_{k}': p.grad.norm() for k, p in st.named_parameters() if p.grad is not None}
        save(f'g11_supair_only_{tag}', x=x.to(torch.float32), eps=eps, elbo=elbo, z=prop['z'], log_q=prop['log_q'], **gn)


def g12_reconstruct():
    """MPE rendering, supair.py:357-498: max-activation images, per-glimpse MPE patches, rendered frames."""
    for dtype, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
        c = ref_config(dtype)
        sup = Supair(c)
        fill(sup, 'sup.')
        g = torch.Generator().manual_seed(12)
        n, t = 2, 3
        x = (torch.rand(n, t, 1, 32, 32, generator=g, dtype=torch.float64) ** 3).to(dtype)
        z = crafted_z(n * t, 3, g, dtype).view(n, t, 3, 4)
        with torch.no_grad():
            save(f'g12_reconstruct_{tag}', x=x, z=z,
                 bg_max=sup.spn_max_activation(sup.bg_spn), obj_max=sup.spn_max_activation(sup.obj_spn),
                 mpe_patches=sup.spn_mpe(z.flatten(end_dim=1), x.flatten(end_dim=1)),
                 recon_max=sup.reconstruct_from_z(z),
                 recon_mpe=sup.reconstruct_from_z(z, x, max_activation=False, single_image=False),
                 recon_mpe_single=sup.reconstruct_from_z(z, x[:, 0], max_activation=False, single_image=True))


def g9_optimiser_steps():
    """Three steps of the reference's optimisation recipe (train.py:431-473) on one fixed batch."""
    c = ref_config(torch.float32)
    st = Stove(c)
    fill(st)
    B, T = 4, 8
    x = torch.from_numpy(billiards_frames(B, T)).to(torch.float32)
    opt = torch.optim.Adam(st.parameters(), lr=c.learning_rate, amsgrad=c.debug_amsgrad)
    g = torch.Generator().manual_seed(9)
    elbos, all_eps = [], []
    for step in range(1, 4):
        lat = torch.randn(B, 3, 12, 1, generator=g, dtype=torch.float64)
        sd = torch.randn(B, 3, 12, 1, generator=g, dtype=torch.float64)
        steps = [torch.randn(B, 3, 18, generator=g, dtype=torch.float64) for _ in range(2, T)]
        all_eps.append((lat, sd, torch.stack(steps, 0)))
        lr = max(c.learning_rate * np.exp(-step / c.debug_anneal_lr), c.min_learning_rate)
        for grp in opt.param_groups:
            grp['lr'] = lr
        opt.zero_grad()
        saved = tdn._standard_normal
        tdn._standard_normal = EpsFeeder([lat, sd] + steps)
        elbo, _, _ = st(x, step, None)
        tdn._standard_normal = saved
        (-elbo).backward()
        torch.nn.utils.clip_grad_norm_(st.parameters(), 1)
        opt.step()
        elbos.append(float(elbo))
    checks = {f'p_{k}': p.detach().double().sum() for k, p in st.named_parameters()
              if k in ('dyn.out.0.1.weight', 'sup.encoder.fc2.weight', 'sup.obj_spn.vector_list.4.0.params',
                       'sup.bg_spn.vector_list.0.0.means', 'dyn.rel_cores.0.0.weight')}
    save('g9_optimiser_steps', x=x, elbos=np.array(elbos),
         eps_lat=torch.stack([e[0] for e in all_eps]), eps_std=torch.stack([e[1] for e in all_eps]),
         eps_steps=torch.stack([e[2] for e in all_eps]), **checks)


# ------------------------------------------------------------------ G13 beyond the 32 x 32 / align_corners=False contract
def gravity_frames_res(n_seq, t_len, res):
    """the reference's stock gravity generator (envs.py:841-844: res = 50)"""
    xs = []
    for seed in range(n_seq):
        env = ref_envs.GravityEnv(n=3, r=2, m=4., hw=30, granularity=50, res=res, t=1.,
                                  init_v_factor=0.55, friction_coefficient=0., seed=seed)
        xs.append(np.stack([env.step()[0].copy() for _ in range(t_len)]))
    return np.transpose(np.stack(xs), (0, 1, 4, 2, 3))


class _AlignCorners:
    """The torch 1.0.1 sampling convention the reference was written for (requirements.txt:103): its calls pass no flag
    (supair.py:272-275, 321-341), torch >= 1.3 then samples with align_corners=False; this context makes them sample with True."""

    def __enter__(self):
        import torch.nn.functional as Fn
        self.Fn, self.ag, self.gs = Fn, Fn.affine_grid, Fn.grid_sample
        Fn.affine_grid = lambda theta, size, align_corners=None: self.ag(theta, size, align_corners=True)
        Fn.grid_sample = lambda inp, grid, mode='bilinear', padding_mode='zeros', align_corners=None: self.gs(
            inp, grid, mode=mode, padding_mode=padding_mode, align_corners=True)

    def __exit__(self, *exc):
        self.Fn.affine_grid, self.Fn.grid_sample = self.ag, self.gs
        return False


class _Nothing:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


def g13_wide():
    """Supair.likelihood and the full Stove.forward (a) on 50 x 50 frames (the reference's stock gravity data) and (b) on 32 x 32
    frames under align_corners=True.  float64 only (the GPU tests compare against float64)."""
    dtype, tag = torch.float64, 'f64'
    for name, res, ac in (('res50', 50, False), ('ac32', 32, True)):
        ctx = _AlignCorners() if ac else _Nothing()
        with ctx:
            # likelihood (as g4)
            c = ref_config(dtype, num_obj=3, width=res, height=res)
            c.debug = True
            sup = Supair(c)
            fill(sup, 'sup.')
            g = torch.Generator().manual_seed(4)
            n, t = 2, 3
            x = (torch.rand(n, t, 1, res, res, generator=g, dtype=torch.float64) ** 3).to(dtype)
            z = crafted_z(n * t, 3, g, dtype).flatten(end_dim=1).requires_grad_()
            sup.step_counter = 0
            lp, prop = sup.likelihood(x, z)
            w = torch.linspace(0.5, 1.5, n * t, dtype=dtype)
            (lp * w).sum().backward()
            grads = {f'gn_{k}': p.grad.norm() for k, p in sup.named_parameters() if p.grad is not None}
            small = {f'g_{k}': p.grad for k, p in sup.named_parameters() if p.grad is not None and p.numel() <= 700}
            save(f'g13_likelihood_{name}_{tag}', x=x, z=z, log_p=lp, w=w, gz=z.grad, bg=prop['bg'], patch=prop['patch'],
                 overlap=prop['overlap'], **grads, **small)
            # full model (as g7)
            c = ref_config(dtype, num_obj=3, width=res, height=res)
            c.debug = True
            st = Stove(c)
            fill(st)
            B, T, N = 2, 6, 3
            x = torch.from_numpy(gravity_frames_res(B, T, res)).to(dtype)
            g = torch.Generator().manual_seed(123)
            lat = torch.randn(B, N, 12, 1, generator=g, dtype=torch.float64)
            sd = torch.randn(B, N, 12, 1, generator=g, dtype=torch.float64)
            steps = [torch.randn(B, N, 6 if kw.get('debug_no_latents') else 18, generator=g, dtype=torch.float64) for _ in range(2, T)]
            saved = tdn._standard_normal
            tdn._standard_normal = EpsFeeder([lat, sd] + steps)
            elbo, prop, _ = st(x, 0, None)
            tdn._standard_normal = saved
            (-elbo).backward()
            gnorm = {f'gn_{k}': p.grad.norm() for k, p in st.named_parameters() if p.grad is not None}
            small = {f'g_{k}': p.grad for k, p in st.named_parameters() if p.grad is not None and p.numel() <= 700}
            props = {f'p_{k}': v for k, v in prop.items() if v is not None and torch.is_tensor(v)}
            with torch.no_grad():
                rec = st.reconstruct_from_z(prop['z'])
            save(f'g13_stove_{name}_{tag}', x=x.to(torch.float32), eps_lat=lat, eps_std=sd, eps_steps=torch.stack(steps, 0), elbo=elbo,
                 recon=rec.to(torch.float32), **props, **gnorm, **small)


def g14_spn_shapes():
    """Supair.likelihood with an object SPN of another shape: 8 x 12 glimpses, 7 Gaussians per leaf, 5 sums per inner region
    (config.patch_width / patch_height / obj_spn_num_gauss / obj_spn_num_sums, reference config.py:99-100, 119-120), and the
    RatSpn operator itself on random inputs with out-of-range marginalisation.  float64."""
    dtype, tag = torch.float64, 'f64'
    kw = dict(patch_width=8, patch_height=12, obj_spn_num_gauss=7, obj_spn_num_sums=5)
    c = ref_config(dtype, num_obj=3, **kw)
    c.debug = True
    sup = Supair(c)
    fill(sup, 'sup.')
    g = torch.Generator().manual_seed(14)
    # the operator
    n, d = 37, 96
    xin = torch.rand(n, d, generator=g, dtype=torch.float64).to(dtype).requires_grad_()
    marg = (torch.rand(n, d, generator=g, dtype=torch.float64) * 1.4 - 0.2).to(dtype).requires_grad_()
    out = sup.obj_spn.forward(xin, marg)
    w = torch.linspace(0.5, 1.5, n, dtype=dtype)
    (out[:, 0] * w).sum().backward()
    og = {f'og_{k}': p.grad.clone() for k, p in sup.obj_spn.named_parameters() if p.grad is not None and p.numel() <= 2000}
    ogn = {f'ogn_{k}': p.grad.norm() for k, p in sup.obj_spn.named_parameters() if p.grad is not None}
    save(f'g14_objspn_8x12_{tag}', x=xin, marg=marg, out=out, w=w, gx=xin.grad, gmarg=marg.grad, **og, **ogn)
    sup.zero_grad()
    # the likelihood
    n, t = 2, 3
    x = (torch.rand(n, t, 1, 32, 32, generator=g, dtype=torch.float64) ** 3).to(dtype)
    z = crafted_z(n * t, 3, g, dtype).flatten(end_dim=1).requires_grad_()
    sup.step_counter = 0
    lp, prop = sup.likelihood(x, z)
    w = torch.linspace(0.5, 1.5, n * t, dtype=dtype)
    (lp * w).sum().backward()
    grads = {f'gn_{k}': p.grad.norm() for k, p in sup.named_parameters() if p.grad is not None}
    small = {f'g_{k}': p.grad for k, p in sup.named_parameters() if p.grad is not None and p.numel() <= 700}
    save(f'g14_likelihood_8x12_{tag}', x=x, z=z, log_p=lp, w=w, gz=z.grad, bg=prop['bg'], patch=prop['patch'],
         overlap=prop['overlap'], **grads, **small)


def g15_simple_models():
    """Supair.likelihood with the reference's fixed-Gaussian debug models (config.debug_bg_model / debug_obj_spn,
    supair.py:33-42, probabilistic_models.py:42-90): both on, and each on beside the other SPN.  float64."""
    dtype, tag = torch.float64, 'f64'
    for name, kw in (('both', dict(debug_bg_model=True, debug_obj_spn=True)), ('bg', dict(debug_bg_model=True)), ('obj', dict(debug_obj_spn=True))):
        c = ref_config(dtype, num_obj=3, **kw)
        c.debug = True
        sup = Supair(c)
        fill(sup, 'sup.')
        g = torch.Generator().manual_seed(15)
        n, t = 2, 3
        x = (torch.rand(n, t, 1, 32, 32, generator=g, dtype=torch.float64) ** 3).to(dtype)
        z = crafted_z(n * t, 3, g, dtype).flatten(end_dim=1).requires_grad_()
        sup.step_counter = 0
        lp, prop = sup.likelihood(x, z)
        w = torch.linspace(0.5, 1.5, n * t, dtype=dtype)
        (lp * w).sum().backward()
        grads = {f'gn_{k}': p.grad.norm() for k, p in sup.named_parameters() if p.grad is not None}
        save(f'g15_likelihood_simple_{name}_{tag}', x=x, z=z, log_p=lp, w=w, gz=z.grad, bg=prop['bg'], patch=prop['patch'],
             overlap=prop['overlap'], **grads)


def g16_regimes():
    """Round 5: Supair.likelihood (g4), Dynamics.forward (g5) and the full Stove.forward + rollout (g7) once more, with the
    model in the two other weight regimes of tests/golden/analytic_weights.py -- 'init' (the reference's initial statistics) and
    'stress' (saturated: variances at their bounds, near one-hot sums, constrain_zp / constrain_z_dyn at both ends of their
    sigmoids, glimpses leaving the frame).  float64 fixtures; the float32 runs are kept as well because the reference's own
    fp32-vs-fp64 gap in a saturated model is what a fp32 implementation can be held to."""
    only = os.environ.get('G16_ONLY')
    for regime in ('init', 'stress'):
        if only and regime not in only.split(','):
            continue
        g4_likelihood(regime)
        g5_dynamics(regime)
        g7_g8_full(regime)


def g19_ablations():
    """Round 6: the full_state ablations (stove.py:140-160; config.py:79,132,134) -- debug_no_velocity, debug_no_latents (q(z) over six
    dimensions, six noise values per object and step) and debug_no_reuse (overwritten inside the reference's full_state: == the
    default model) -- Stove.forward + backward + rollout, B = 2, T = 6, float64."""
    g7_g8_full('analytic', dtypes=((torch.float64, 'f64'),), prefix='g19_stove', cases=[
        ('novel', dict(num_obj=3, debug_no_velocity=True), 2, 6),
        ('nolat', dict(num_obj=3, debug_no_latents=True), 2, 6),
        ('noreuse', dict(num_obj=3, debug_no_reuse=True), 2, 6)])


def g17_full_length():
    """Round 6: Stove.forward + backward at the length every BASELINE config runs -- B = 2, T = 100 (98 dependent steps of the
    inference recursion, stove.py:696-713) -- for the four workloads in the three weight regimes, each followed by a 92-step
    rollout (stove.py:823-846).  float64 fixtures are committed; the float32 runs feed oracle/fp32_gap.py and are deleted."""
    only = os.environ.get('G17_ONLY')
    for regime in ('analytic', 'init', 'stress'):
        if only and regime not in only.split(','):
            continue
        g7_g8_full(regime, shape=(2, 100), prefix='g17_stove_T100', roll_all=92)


if __name__ == '__main__':
    which = sys.argv[1:] or ['g0', 'g1', 'g2', 'g3', 'g4', 'g5', 'g6', 'g10', 'g7', 'g6b', 'g11', 'g9', 'g12', 'g13', 'g14', 'g15', 'g16', 'g17', 'g19']
    os.makedirs(OUT, exist_ok=True)
    table = {'g0': g0_envs, 'g1': g1_structures, 'g2': g2_ratspn, 'g3': g3_masks_glimpses,
             'g4': g4_likelihood, 'g5': g5_dynamics, 'g6': g6_matchers, 'g10': g10_units, 'g7': g7_g8_full,
             'g6b': g6b_volatile, 'g11': g11_supair_only, 'g9': g9_optimiser_steps, 'g12': g12_reconstruct, 'g13': g13_wide, 'g14': g14_spn_shapes, 'g15': g15_simple_models, 'g16': g16_regimes, 'g17': g17_full_length, 'g19': g19_ablations}
    for k in which:
        torch.manual_seed(0)
        np.random.seed(0)
        table[k]()
