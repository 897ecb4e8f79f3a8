import torch, time
dev = 'cuda:0'
def bench(f, name, flop):
    for _ in range(3): f()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
    print('%-40s %7.1f us %6.1f TFLOP/s' % (name, dt * 1e6, flop / dt / 1e12))
n = 25600
x = torch.randn(n, 1024, device=dev); w = torch.randn(1024, 1024, device=dev); b = torch.randn(1024, device=dev)
h = torch.randn(n, 256, device=dev); whh = torch.randn(1024, 256, device=dev); dg = torch.randn(n, 1024, device=dev)
for lib in ('default', 'hipblaslt', 'hipblas'):
    try:
        if lib != 'default':
            torch.backends.cuda.preferred_blas_library(lib)
    except Exception as e:
        print(lib, 'unavailable', e); continue
    print('---', lib, torch.backends.cuda.preferred_blas_library())
    bench(lambda: torch.addmm(b, x, w.t()), 'addmm(b, x, W_ih^T)', 2.0 * n * 1024 * 1024)
    bench(lambda: torch.mm(dg.t(), x), 'mm(dgx^T, x)', 2.0 * n * 1024 * 1024)
    bench(lambda: torch.mm(h, whh.t()), 'mm(h, W_hh^T)', 2.0 * n * 1024 * 256)
    bench(lambda: torch.addmm(h, dg, whh), 'addmm(dh, dg, W_hh)', 2.0 * n * 1024 * 256)
