"""Probe: split-K of the LSTM dW_hh GEMM (K = frames, 1024 x 256 output) as a batched GEMM + sum."""
import time
import torch
dev = 'cuda:0'
def bench(f, name, flop):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('%-46s %7.1f us  %6.1f TFLOP/s' % (name, dt * 1e6, flop / dt / 1e12))
for n in (25600, 51200):
    dg = torch.randn(n, 1024, device=dev); hs = torch.randn(n, 256, device=dev)
    fl = 2.0 * n * 1024 * 256
    bench(lambda: torch.mm(dg.t(), hs), 'n=%d mm(dg.T, hs)' % n, fl)
    ref = torch.mm(dg.t(), hs)
    for S in (4, 8, 16, 32, 64):
        a = dg.view(S, n // S, 1024).transpose(1, 2); b = hs.view(S, n // S, 256)
        bench(lambda: torch.bmm(a, b).sum(0), 'n=%d bmm S=%d + sum' % (n, S), fl)
        a2 = hs.view(S, n // S, 256).transpose(1, 2); b2 = dg.view(S, n // S, 1024)
        bench(lambda: torch.bmm(a2, b2).sum(0), 'n=%d bmm^T S=%d + sum' % (n, S), fl)
    got = torch.bmm(dg.view(16, n // 16, 1024).transpose(1, 2), hs.view(16, n // 16, 256)).sum(0)
    print('   rel diff', float((got - ref).abs().max() / ref.abs().max()))
print('--- d_wih: (1024 x n) @ (n x 1024)')
n = 25600
dgx = torch.randn(n, 1024, device=dev); x = torch.randn(n, 1024, device=dev)
fl = 2.0 * n * 1024 * 1024
bench(lambda: torch.mm(dgx.t(), x), 'mm(dgx.T, x)', fl)
for S in (2, 4, 8, 16):
    a = dgx.view(S, n // S, 1024).transpose(1, 2); b = x.view(S, n // S, 1024)
    bench(lambda: torch.bmm(a, b).sum(0), 'bmm S=%d + sum' % S, fl)
print('--- gx = x @ W_ih^T (n x 1024) @ (1024 x 1024)')
w = torch.randn(1024, 1024, device=dev)
bench(lambda: torch.mm(x, w.t()), 'mm(x, w.T)', fl)
bench(lambda: torch.mm(x, w), 'mm(x, w)', fl)
bias = torch.randn(1024, device=dev)
bench(lambda: torch.addmm(bias, x, w.t()), 'addmm(bias, x, w.T)', fl)
print('--- dh = dg @ W_hh (n x 1024) @ (1024 x 256)')
whh = torch.randn(1024, 256, device=dev); dh0 = torch.randn(n, 256, device=dev)
bench(lambda: torch.addmm(dh0, dgx, whh), 'addmm(dh, dg, whh)', 2.0 * n * 1024 * 256)
print('--- gh = h @ W_hh^T (n x 256) @ (256 x 1024)')
h = torch.randn(n, 256, device=dev)
bench(lambda: torch.mm(h, whh.t()), 'mm(h, whh.T)', 2.0 * n * 1024 * 256)
