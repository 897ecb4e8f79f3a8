"""One GEMM shape in a loop, for rocprofv3 --pmc passes.  Usage: python3 tools/gemm_pmc.py [tile]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from stove_amd import ops

tile = int(sys.argv[1]) if len(sys.argv) > 1 else 0
dev = torch.device('cuda:0')
x = torch.rand(25600, 1024, device=dev)
w = torch.randn(1024, 1024, device=dev) * 0.03
for _ in range(5):
    c = ops.gemm_bf16(x, w, None, False, False, 2, 1, tile=tile)
torch.cuda.synchronize()
