"""FlatAdam (csrc/arena.hip: grad_scan_k / flat_adam_k / adam_tick_k through stove_flat_adam) against torch.optim.Adam,
including the per-parameter semantics the reference's optimiser has (train.py:46-49): parameters without a gradient are
skipped, every parameter has its own step count."""
import copy

import pytest
import torch
import torch.nn as nn

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


class _Net(nn.Module):
    def __init__(self):
        super().__init__()
        self.a = nn.Linear(8, 16)
        self.b = nn.Linear(16, 5)          # odd sizes: segments are padded to float4 boundaries
        self.unused = nn.Linear(3, 3)      # never takes part (the dynamics cores 1-2 of the reference, stove.py:698-699)

    def forward(self, x, full):
        h = torch.tanh(self.a(x))
        return self.b(h) if full else h


def _loss(net, x, full):
    return (net(x, full) ** 2).sum(1).mean()


@pytest.mark.parametrize('amsgrad', [True, False])
def test_flat_adam_matches_torch_adam_with_late_and_frozen_parameters(amsgrad):
    from stove_amd.arena import ParamArena
    from stove_amd.optim import FlatAdam
    torch.manual_seed(0)
    ref = _Net().to(DEV)
    net = copy.deepcopy(ref)
    arena = ParamArena(net, 1)
    opt = FlatAdam(arena, lr=1e-2, amsgrad=amsgrad)
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, amsgrad=amsgrad)
    g = torch.Generator().manual_seed(1)
    step = 0

    def both(full, clip):
        nonlocal step
        step += 1
        x = torch.randn(32, 8, generator=g).to(DEV)
        for grp in list(opt.param_groups) + list(ropt.param_groups):
            grp['lr'] = 1e-2 * 0.9 ** step
        arena.zero_grad()
        _loss(net, x, full).backward()
        norm = opt.step(max_norm=clip)
        ropt.zero_grad(set_to_none=True)
        _loss(ref, x, full).backward()
        if clip is not None:
            rnorm = torch.nn.utils.clip_grad_norm_(ref.parameters(), clip)
            assert abs(float(norm) - float(rnorm)) < 1e-5 * float(rnorm)
        ropt.step()
        for (n, p), q in zip(net.named_parameters(), ref.parameters()):
            assert float((p - q).abs().max()) < 2e-6, (step, n, float((p - q).abs().max()))

    # 1. only `a` is in the graph: `b` has no gradient (torch: grad None -> skipped, no state)
    for _ in range(3):
        both(False, 0.05)
    # 2. `b` joins late: its bias corrections start from ITS first step, not from the global count
    for _ in range(3):
        both(True, 0.05)
    sd, rsd = opt.state_dict(), ropt.state_dict()
    names = [n for n, _ in net.named_parameters()]
    for i, n in enumerate(names):
        want = float(rsd['state'][i]['step']) if i in rsd['state'] else 0.0
        assert float(sd['state'][i]['step']) == want, (n, float(sd['state'][i]['step']), want)
    assert float(sd['state'][names.index('a.weight')]['step']) == 6 and float(sd['state'][names.index('b.weight')]['step']) == 3
    # 3. freeze `a` (Trainer.disable_supair_grad): its non-zero moments must not keep moving it
    for m in (net, ref):
        for p in m.a.parameters():
            p.requires_grad = False
            p.grad = None
    frozen = net.a.weight.detach().clone()
    for _ in range(3):
        both(True, None)
    assert torch.equal(net.a.weight, frozen)
    assert float(net.unused.weight.grad.abs().max()) == 0.0 and torch.equal(net.unused.weight, ref.unused.weight)
    # 4. a checkpoint of either optimiser restores the other's state (per-parameter steps included)
    opt2 = FlatAdam(arena, lr=1e-2, amsgrad=amsgrad)
    opt2.load_state_dict(ropt.state_dict())
    assert opt2._seg_steps.tolist() == opt._seg_steps.tolist()
    for k in opt._flat:
        assert float((opt2._flat[k] - opt._flat[k]).abs().max()) < 1e-6


def test_flat_adam_strict_zero_gradients_matches_torch_on_zero_filled_grads():
    """FlatAdam(strict_zero_grad=True) (config.strict_adam_zero_grad): a tensor that has received a gradient keeps stepping on its
    momentum while its gradient is all zero -- torch.optim.Adam with zero_grad(set_to_none=False), the zero_grad() of the torch 1.0.1
    the reference pins; a tensor that never received one stays untouched in both.  Default mode: it stands still."""
    from stove_amd.arena import ParamArena
    from stove_amd.optim import FlatAdam
    torch.manual_seed(0)
    ref = _Net().to(DEV)
    nets = {False: copy.deepcopy(ref), True: copy.deepcopy(ref)}
    arenas = {k: ParamArena(n, 1) for k, n in nets.items()}
    opts = {k: FlatAdam(arenas[k], lr=1e-2, amsgrad=True, strict_zero_grad=k) for k in nets}
    ropt = torch.optim.Adam(ref.parameters(), lr=1e-2, amsgrad=True)
    g = torch.Generator().manual_seed(2)
    for step in range(6):
        full = step < 3                                        # `b` trains for three steps, then drops out of the graph
        x = torch.randn(32, 8, generator=g).to(DEV)
        for k in nets:
            arenas[k].zero_grad()
            _loss(nets[k], x, full).backward()
            opts[k].step()
        ropt.zero_grad(set_to_none=False)                      # zero-filled .grad tensors
        _loss(ref, x, full).backward()
        ropt.step()
        for (n, p), q in zip(nets[True].named_parameters(), ref.parameters()):
            if q.grad is not None:                             # `unused` never had a gradient: None in torch, skipped in both
                assert float((p - q).abs().max()) < 2e-6, (step, n)
        if step == 2:
            kept = nets[False].b.weight.detach().clone()
    assert torch.equal(nets[False].b.weight, kept)             # default mode: no gradient, no step
    assert not torch.equal(nets[True].b.weight, kept)          # strict mode: the momentum kept moving it
    assert torch.equal(nets[True].unused.weight, ref.unused.weight)


@pytest.mark.parametrize('rows,M,N', [(25088, 12, 9), (3000, 1, 8), (2048, 16, 16), (5, 3, 4), (0, 2, 2)])
def test_small_tn_weight_gradient(rows, M, N):
    """a^T b for narrow operands over many rows (the action embedding's weight gradient): against float64, bit-reproducible."""
    from stove_amd import ops
    g = torch.Generator().manual_seed(rows + M)
    a, b = torch.randn(rows, M, generator=g), torch.randn(rows, N, generator=g)
    got = ops.small_tn(a.to(DEV), b.to(DEV))
    ref = a.double().t() @ b.double()
    assert got.shape == (M, N)
    if rows:
        assert float((got.cpu().double() - ref).abs().max()) <= 2e-6 * max(1.0, float(ref.abs().max()))
    assert torch.equal(got, ops.small_tn(a.to(DEV), b.to(DEV)))
