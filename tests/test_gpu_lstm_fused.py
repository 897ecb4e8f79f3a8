"""The LSTM of the recognition network with its cells inside the products (ops._EncoderLstmFusedFn; csrc/gemm_bf16.hip EPI 1 / 2,
csrc/lstm.hip) against (a) torch.nn.LSTM's arithmetic in float64 on the CPU (reference encoder.py:43-51: the same frame fed for
num_obj steps) and (b) the round-3 chain of products and stand-alone cell kernels, for ragged row counts (edge tiles, rows below
one tile, the split-K tail launch of the input projection), 1 / 2 / 3 / 6 steps, both operand precisions, and through the arena's
gradient views.  Tolerances are pinned at ~4x what an MI355X achieves (tests/gpu_helpers.check records the worst error)."""
import pytest
import torch

from gpu_helpers import check, check_grad, err

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _ref_lstm(x, w_ih, w_hh, b_ih, b_hh, steps):
    """float64, op by op (torch.nn.LSTM cell equations, gate order i, f, g, o)"""
    H = w_hh.shape[1]
    h = torch.zeros(x.shape[0], H, dtype=torch.float64)
    c = torch.zeros_like(h)
    gx = x @ w_ih.t() + b_ih + b_hh
    hs = []
    for _ in range(steps):
        g = gx + h @ w_hh.t()
        i, f, gg, o = torch.sigmoid(g[:, :H]), torch.sigmoid(g[:, H:2 * H]), torch.tanh(g[:, 2 * H:3 * H]), torch.sigmoid(g[:, 3 * H:])
        c = f * c + i * gg
        h = o * torch.tanh(c)
        hs.append(h)
    return torch.stack(hs, 1)                       # (n, steps, H)


def _params(D, H, seed):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(4 * H, D, generator=g, dtype=torch.float64) * (1.0 / D ** 0.5),
            torch.randn(4 * H, H, generator=g, dtype=torch.float64) * (1.5 / H ** 0.5),
            torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.3, torch.randn(4 * H, generator=g, dtype=torch.float64) * 0.3)


@pytest.mark.parametrize('n,D,H,steps', [(37, 1024, 256, 3), (1, 1024, 256, 2), (300, 1024, 256, 1), (300, 1024, 256, 6), (129, 64, 32, 3),
                                         (2048, 1024, 256, 3), (25600 + 1024 + 5, 1024, 256, 2)])
@pytest.mark.parametrize('gemm', ['bf16x3', 'bf16'])
def test_fused_lstm_against_float64_and_the_unfused_chain(n, D, H, steps, gemm):
    from stove_amd import ops
    if gemm == 'bf16' and n > 4000:
        pytest.skip('one large case per precision is enough')
    p64 = _params(D, H, n + steps)
    g = torch.Generator().manual_seed(3)
    x64 = torch.rand(n, D, generator=g, dtype=torch.float64)
    w64 = torch.randn(n, steps, H, generator=g, dtype=torch.float64)
    ref_in = [t.clone().requires_grad_() for t in p64]
    out64 = _ref_lstm(x64, *ref_in, steps)
    (out64 * w64).sum().backward()
    res = {}
    for fused in (True, False):
        prm = [t.float().to(DEV).requires_grad_() for t in p64]
        assert ops.encoder_lstm_fused_ok(x64.float().to(DEV), prm[1], gemm, steps)
        out = ops.encoder_lstm(x64.float().to(DEV), *prm, steps, time_major=False, gemm=gemm, fused=fused)
        assert out.shape == (n, steps, H)
        (out * w64.float().to(DEV)).sum().backward()
        res[fused] = (out.detach(), [p.grad.clone() for p in prm])
    lo = gemm == 'bf16'
    check('lstm_fused.h' + ('.bf16' if lo else ''), err(res[True][0], out64), 2e-3 if lo else 4e-6)
    check('lstm_fused.h_vs_unfused' + ('.bf16' if lo else ''), err(res[True][0], res[False][0]), 2e-3 if lo else 3e-6)
    for name, a, b, r in zip(('w_ih', 'w_hh', 'b_ih', 'b_hh'), res[True][1], res[False][1], ref_in):
        if lo:
            check('lstm_fused.grad.bf16', err(a, r.grad), 1e-2)
        else:
            check_grad('lstm_fused.grad', a, r.grad, 2e-5, 2e-5, 1e-3)
            check('lstm_fused.grad_vs_unfused', err(a, b), 1e-5)
    # bitwise reproducible
    prm = [t.float().to(DEV).requires_grad_() for t in p64]
    out2 = ops.encoder_lstm(x64.float().to(DEV), *prm, steps, time_major=False, gemm=gemm, fused=True)
    (out2 * w64.float().to(DEV)).sum().backward()
    assert torch.equal(out2.detach(), res[True][0]) and all(torch.equal(p.grad, q) for p, q in zip(prm, res[True][1]))


def test_fused_lstm_input_gradient_and_time_major():
    from stove_amd import ops
    n, D, H, steps = 70, 128, 64, 3
    p64 = _params(D, H, 5)
    g = torch.Generator().manual_seed(4)
    x64 = torch.rand(n, D, generator=g, dtype=torch.float64).requires_grad_()
    w64 = torch.randn(n, steps, H, generator=g, dtype=torch.float64)
    (_ref_lstm(x64, *p64, steps) * w64).sum().backward()
    x = x64.detach().float().to(DEV).requires_grad_()
    prm = [t.float().to(DEV).requires_grad_() for t in p64]
    out = ops.encoder_lstm(x, *prm, steps, time_major=True, gemm='bf16x3', fused=True)
    assert out.shape == (steps, n, H)
    (out.transpose(0, 1) * w64.float().to(DEV)).sum().backward()
    check('lstm_fused.dx', err(x.grad, x64.grad), 2e-5)


def test_fused_lstm_through_the_arena_views_bitwise():
    """Inside the model the parameter gradients are ADDED into the flat arena's views by the producing kernels (split-K slice sums,
    the gate-ordered column sums): the same numbers, bit for bit, as the gradients returned to autograd."""
    from stove_amd import ops
    from stove_amd.arena import ParamArena
    from stove_amd.video_prediction.config import StoveConfig
    from stove_amd.video_prediction.encoder import RnnStates
    cfg = StoveConfig()
    cfg.width, cfg.height, cfg.channels, cfg.num_obj = 32, 32, 1, 3
    torch.manual_seed(1)
    enc = RnnStates(cfg).to(DEV)
    g = torch.Generator().manual_seed(8)
    x = torch.rand(523, 1, 32, 32, generator=g).to(DEV)
    w = torch.randn(523, 3, 8, generator=g).to(DEV)
    (enc(x) * w).sum().backward()
    plain = [p.grad.clone() for p in enc.parameters()]
    arena = ParamArena(enc, 1)
    for fused in (True, False):
        cfg.encoder_fused_cell = fused
        arena.zero_grad()
        (enc(x) * w).sum().backward()
        torch.cuda.synchronize()
        for p, q in zip(enc.parameters(), plain):
            if fused:
                assert torch.equal(p.grad, q)
            else:
                assert float((p.grad - q).abs().max()) <= 2e-5 * float(q.abs().max())
