"""ParamArena host logic on CPU: storage moves into one flat buffer without changing names / values, autograd
accumulates into the flat gradient buffer, the index plans of the bake / gather kernels address the right
parameters (emulated here with torch indexing; the kernels themselves are covered by the -m gpu tests)."""
import math

import torch

import stove_oracle as O  # noqa: F401  (conftest puts oracle/ on the path; only used for the config defaults)
from stove_amd import ops
from stove_amd.arena import ParamArena


def _cpu_stove(**kw):
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.stove import Stove
    cfg = StoveConfig()
    cfg.num_obj, cfg.width, cfg.height, cfg.random_seed = 3, 32, 32, 42
    cfg.device, cfg.dtype = torch.device('cpu'), torch.float32
    for k, v in kw.items():
        setattr(cfg, k, v)
    return Stove(cfg)


def test_arena_moves_storage_only():
    st = _cpu_stove()
    before = {k: v.clone() for k, v in st.state_dict().items()}
    n_named = len(list(st.named_parameters()))
    arena = ParamArena(st)
    after = st.state_dict()
    assert list(after) == list(before) and len(list(st.named_parameters())) == n_named == len(arena.params)
    for k in before:
        assert torch.equal(before[k], after[k]), k
    lo, hi = arena.data.data_ptr(), arena.data.data_ptr() + 4 * arena.numel
    for p in st.parameters():
        assert lo <= p.data_ptr() < hi and p.data_ptr() % 16 == 0
        assert p.grad is not None and arena.grad.data_ptr() <= p.grad.data_ptr() < arena.grad.data_ptr() + 4 * arena.numel
    arena.check()
    # writes through the flat buffer are visible through the module (what Adam / load_state_dict rely on)
    arena.data.add_(1.0)
    for k in before:
        assert torch.allclose(st.state_dict()[k], before[k] + 1.0)
    st.load_state_dict(before)
    arena.check()
    assert torch.equal(arena.view_of(st.dyn.state_enc.weight), before['dyn.state_enc.weight'])


def test_autograd_accumulates_into_flat_gradient_and_clip_matches_torch():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3)).double()
    ref = torch.nn.Sequential(torch.nn.Linear(5, 7), torch.nn.Tanh(), torch.nn.Linear(7, 3)).double()
    ref.load_state_dict(net.state_dict())
    arena = ParamArena(net, world_size=1)
    x = torch.randn(11, 5, dtype=torch.float64)
    for _ in range(2):
        arena.zero_grad()
        ref.zero_grad()
        (net(x) ** 2).sum().backward()
        (ref(x) ** 2).sum().backward()
        arena.check()
        for p, q in zip(net.parameters(), ref.parameters()):
            assert torch.allclose(p.grad, q.grad, rtol=1e-12, atol=0)
            assert torch.equal(p.grad, arena.view_of(p, arena.grad))
    n1 = arena.clip_grad_norm_(1.0)
    n2 = torch.nn.utils.clip_grad_norm_(ref.parameters(), 1.0)
    assert abs(float(n1) - float(n2)) < 1e-12 * float(n2)
    for p, q in zip(net.parameters(), ref.parameters()):
        assert torch.allclose(p.grad, q.grad, rtol=1e-12, atol=0)
    for p in net.parameters():
        p.grad = None
    try:
        arena.check()
        raise AssertionError('detached gradient views must be refused')
    except RuntimeError as e:
        assert 'zero_grad' in str(e)


def test_gnn_gather_table_addresses_the_param_image():
    for kw in ({}, dict(action_conditioned=True, action_space=9, debug_core_appearance=True)):
        st = _cpu_stove(**kw)
        arena = ParamArena(st)
        for k in range(3):
            w, v, wt = st.dyn.param_image(k)
            pf, pt = ops.gnn_pack_perms(torch.device('cpu'))      # + both weight sections again in the small-graph kernels' LDS order
            want = torch.cat([w, wt, v, w[pf], wt[pt]]).detach()
            assert torch.equal(want, ops._gnn_image(w, v, wt).detach())      # what the non-arena path builds
            src, src_g = arena._gnn[k]
            got = torch.where(src >= 0, arena.data[src.clamp(min=0).long()], torch.zeros(()))
            assert torch.equal(got, want)
            live = src_g[src_g >= 0]
            assert live.numel() == live.unique().numel()          # the scatter never hits an address twice
            assert src_g.numel() == w.numel() + v.numel()


def test_spn_plan_addresses_the_tables():
    st = _cpu_stove()
    arena = ParamArena(st)
    t = arena._spn['keep']
    obj, bg = st.sup.obj_spn, st.sup.bg_spn
    oc, ow, orr, _, _ = obj.tables()
    bc, bw, _ = bg.tables()
    d = arena.data

    def coef(mu, rho, a):
        var = a.gauss_min_sigma + (a.gauss_max_sigma - a.gauss_min_sigma) * torch.sigmoid(rho)
        return torch.stack([-0.5 / var, mu / var, -0.5 * mu * mu / var - 0.5 * torch.log(2 * math.pi * var)], -1)
    r250 = torch.arange(250)
    got = coef(d[t['obj_mu'].long()[:, None] + r250], d[t['obj_rho'].long()[:, None] + r250], obj.args).view(24, 25, 10, 3)
    assert torch.allclose(got, oc.detach(), rtol=1e-6, atol=1e-6)
    r1000 = torch.arange(1000)
    got = torch.softmax(d[t['obj_sum'].long()[:, None] + r1000].view(12, 100, 10), 1)
    assert torch.allclose(got, ow.detach(), rtol=1e-6, atol=1e-7)
    plan = arena._spn['plan']
    assert torch.allclose(torch.softmax(d[plan.obj_root:plan.obj_root + 600], 0).view(6, 100), orr.detach(), atol=1e-7)
    assert torch.allclose(torch.softmax(d[plan.bg_root:plan.bg_root + 108], 0).view(3, 36), bw.detach(), atol=1e-7)
    gi = t['bg_gidx'].long()
    base_mu, base_rho = t['bg_mu'].long()[gi >> 9] + (gi & 511) * 6, t['bg_rho'].long()[gi >> 9] + (gi & 511) * 6
    g6 = torch.arange(6)
    got = coef(d[base_mu[:, None] + g6], d[base_rho[:, None] + g6], bg.args).view(3, 1024, 6, 3)
    assert torch.allclose(got, bc.detach(), rtol=1e-6, atol=1e-6)
