"""Probe: what the library's bf16 GEMM (fp32 output) does on the recognition network's shapes with K tripled (hi/lo split
stacked along K: [Ahi | Ahi | Alo] x [Bhi | Blo | Bhi]^T), next to stove_gemm_bf16."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stove_amd import ops
dev = torch.device('cuda:0')


def t_us(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for (M, N, K, name) in ((25600, 1024, 1024, 'x W_ih^T'), (25600, 1024, 256, 'h W_hh^T'), (25600, 256, 1024, 'dg W_hh')):
    a = torch.randn(M, K, device=dev)
    b = torch.randn(N, K, device=dev)
    a3 = torch.cat([a, a, a], 1).bfloat16().contiguous()
    b3 = torch.cat([b, b, b], 1).bfloat16().contiguous()
    flops = 2.0 * M * N * K
    try:
        us = t_us(lambda: torch.mm(a3, b3.t(), out_dtype=torch.float32))
        print('%-10s lib bf16 K x3, fp32 out : %7.1f us  (%.0f TFLOP/s on the pipe)' % (name, us, 3 * flops / us * 1e-6))
    except Exception as e:
        print(name, 'out_dtype mm failed:', str(e)[:100])
    us = t_us(lambda: torch.mm(a3, b3.t()))
    print('%-10s lib bf16 K x3, bf16 out : %7.1f us' % (name, us))
    us = t_us(lambda: ops.gemm_bf16(a, b))
    print('%-10s stove_gemm_bf16         : %7.1f us' % (name, us))
# weight gradient: dgx^T x, K = frames
M, N, K = 1024, 1024, 25600
a = torch.randn(K, M, device=dev)
b = torch.randn(K, N, device=dev)
a3 = torch.cat([a, a, a], 0).bfloat16().contiguous()
b3 = torch.cat([b, b, b], 0).bfloat16().contiguous()
try:
    us = t_us(lambda: torch.mm(a3.t(), b3, out_dtype=torch.float32))
    print('dgx^T x    lib bf16 K x3, fp32 out : %7.1f us  (%.0f TFLOP/s on the pipe)' % (us, 3 * 2.0 * M * N * K / us * 1e-6))
except Exception as e:
    print('wgrad out_dtype mm failed:', str(e)[:100])
print('dgx^T x    stove_gemm_bf16         : %7.1f us' % t_us(lambda: ops.gemm_bf16(a, b, None, True, True, 2)))
