import sys, torch
sys.path.insert(0, '.')
from stove_amd import ops
DEV = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for name, x, w in (
    ('frames x W_ih(0.03)', torch.rand(4096, 1024, generator=g), torch.randn(1024, 1024, generator=g) * 0.03),
    ('frames(sparse) x W_ih U(1/16)', (torch.rand(4096, 1024, generator=g) > 0.9).float() * torch.rand(4096, 1024, generator=g), (torch.rand(1024, 1024, generator=g) - 0.5) / 8),
    ('h(0.3) x W_hh(0.06)', torch.tanh(torch.randn(4096, 256, generator=g) * 0.3), (torch.rand(1024, 256, generator=g) - 0.5) / 8),
    ('h x W(0.003)', torch.tanh(torch.randn(4096, 256, generator=g) * 0.3), torch.randn(1024, 256, generator=g) * 0.003),
):
    x, w = x.to(DEV), w.to(DEV)
    ref = x.double() @ w.double().t()
    sc = float(ref.abs().max())
    def e(c): return float((c.double() - ref).abs().max()) / sc
    lib = x @ w.t()
    r = {'lib': e(lib), 'bf16x2': e(ops.gemm_bf16(x, w, nsplit=2, splitk=1)), 'half': e(ops.gemm_bf16(x, w, nsplit=3, splitk=1))}
    for sa, sb in ((0, 8), (8, 8), (4, 6), (12, 8)):
        r['half 2^%d,2^%d' % (sa, sb)] = e(ops.gemm_bf16(x * 2.0 ** sa, w * 2.0 ** sb, nsplit=3, splitk=1) * 2.0 ** -(sa + sb))
    print(name, {k: '%.2e' % v for k, v in r.items()})
