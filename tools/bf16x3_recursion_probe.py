"""Would the recursion survive split-bf16 (hi/lo, 3-product) matrix products?  CPU emulation on the oracle: the GNN core's linear layers
with x W^T replaced by hi W_hi^T + hi W_lo^T + lo W_hi^T (fp32 accumulate), full Stove forward at T = 100, against fp64."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle')); sys.path.insert(0, ROOT)
import torch
import stove_oracle as O
from stove_amd.envs import envs

B, T = 8, 100
c = O.default_config()
structs = O.build_structs(c)
torch.manual_seed(0)
P64 = {}
for k, shp in O.param_shapes(c, structs).items():
    scale = 0.1 if k.endswith(('means', 'sigma_params', 'params')) else 1.0 / max(1.0, float(shp[-1])) ** 0.5
    P64[k] = (torch.randn(*shp, dtype=torch.float64) * scale)
x = torch.from_numpy(envs.synth_sequences('billiards', B, T, seed0=0)['X'])
g = torch.Generator().manual_seed(1)
eps64 = O.draw_eps(B, c.num_obj, T, generator=g, dtype=torch.float64)


def run(dtype, mode):
    params = {k: v.to(dtype) for k, v in P64.items()}
    eps = {k: ([e.to(dtype) for e in v] if isinstance(v, list) else v.to(dtype)) for k, v in eps64.items()}
    orig = O._lin

    def split(t):
        hi = t.bfloat16().float()
        lo = (t - hi).bfloat16().float()
        return hi, lo

    def lin(params_, name, xx):
        if mode == 'fp' or not name.startswith('dyn.'):
            return orig(params_, name, xx)
        w, b = params_[name + '.weight'], params_[name + '.bias']
        xh, xl = split(xx)
        wh, wl = split(w)
        if mode == 'bf16':
            return xh @ wh.t() + b
        return (xh @ wh.t() + xh @ wl.t() + xl @ wh.t()) + b
    O._lin = lin
    try:
        with torch.no_grad():
            elbo, _, prop = O.stove_forward(c, params, structs, x.to(dtype), eps, detail=True)
    finally:
        O._lin = orig
    return float(elbo), prop


e64, p64 = run(torch.float64, 'fp')
for dtype, mode in ((torch.float32, 'fp'), (torch.float32, 'bf16x3'), (torch.float32, 'bf16')):
    e, p = run(dtype, mode)
    dz = float((p['z'].double() - p64['z']).abs().max()) if 'z' in p else float('nan')
    print('%-8s ELBO %.6f  rel err %.2e   max |z - z64| %.2e' % (mode if mode != 'fp' else 'fp32', e, abs(e - e64) / abs(e64), dz))
print('fp64 ELBO %.6f' % e64)
