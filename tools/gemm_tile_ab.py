"""A/B of the GEMM tiles on the recognition network's shapes (25 600 frames): 256 x 128 (eight waves, two per SIMD) against
256 x 256 (four waves of 128 x 128, one per SIMD).  Usage: python tools/gemm_tile_ab.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stove_amd import ops

dev = torch.device('cuda:0')
n = 25600


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(reps))
    return ts[len(ts) // 2] * 1e3


COLD = os.environ.get('GEMM_COLD', '0') == '1'      # every timed launch behind a 1 GiB fill: operands come from HBM, as in the step
flush = torch.empty(1 << 28, dtype=torch.float32, device=dev) if COLD else None


def timeit_cold(fn, reps=12):
    fn()
    ts = []
    for _ in range(reps):
        flush.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2] * 1e3


if COLD:
    timeit = timeit_cold

x = torch.rand(n, 1024, device=dev)
w_ih = torch.randn(1024, 1024, device=dev) * 0.03
w_hh = torch.randn(1024, 256, device=dev) * 0.06
h = torch.randn(n, 256, device=dev)
gx = torch.randn(n, 1024, device=dev)
dg = torch.randn(n, 1024, device=dev)
dg2 = torch.randn(2 * n, 1024, device=dev)
h2 = torch.randn(2 * n, 256, device=dev)
cases = [
    ('gx = x W_ih^T  25600x1024x1024', lambda t, sk: ops.gemm_bf16(x, w_ih, None, False, False, 2, sk, tile=t), 2 * n * 1024 * 1024, (1,), (1,)),
    ('gh = h W_hh^T + gx  25600x1024x256', lambda t, sk: ops.gemm_bf16(h, w_hh, None, False, False, 2, sk, add=gx, tile=t), 2 * n * 1024 * 256, (1,), (1,)),
    ('dh = dg W_hh  25600x256x1024', lambda t, sk: ops.gemm_bf16(dg, w_hh, None, False, True, 2, sk, tile=t), 2 * n * 1024 * 256, (1, 2), (1, 2, 4)),
    ('dW_ih = dgx^T x  1024x1024x25600', lambda t, sk: ops.gemm_bf16(dg, x, None, True, True, 2, sk, tile=t), 2 * n * 1024 * 1024, (8,), (8, 16)),
    ('dW_hh = dg^T h  1024x256x51200', lambda t, sk: ops.gemm_bf16(dg2, h2, None, True, True, 2, sk, tile=t), 4 * n * 1024 * 256, (32,), (32, 64)),
]
for name, fn, flops, sk1, sk3 in cases:
    ref = fn(1, sk1[0])
    for tile, sks in ((1, sk1), (2, sk1), (3, sk3)):
        for sk in sks:
            got = fn(tile, sk)
            err = float((got - ref).abs().max() / ref.abs().max())
            t = timeit(lambda: fn(tile, sk))
            print('%-38s tile %d splitk %2d  %7.1f us  %6.1f TF fp32-equiv  %6.1f TF on the pipe   (max diff to tile 1: %.1e)' % (
                name, tile, sk, t, flops / t / 1e6, 3 * flops / t / 1e6, err), flush=True)
