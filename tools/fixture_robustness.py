"""Are the g7 fixtures' matching decisions robust at float32 resolution?  (TEST INFRASTRUCTURE; CPU, uses the oracle.)  The matchers
(reference stove.py:200-514) take argmins over distances between objects; where two candidates tie to 1e-7 the choice is made
below float32 resolution and no float32 implementation can be held to the float64 fixture.  Adds Gaussian noise of 3e-5 to the
recognition network's codes (an fp32 implementation's error in the 'stress' regime is ~1e-5) and reports the largest change of
the matched states over 12 draws: ~noise = every decision unchanged, >> noise = a decision flipped.
    python tools/fixture_robustness.py stress|analytic|init"""
import os
import sys
_R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(_R, 'oracle'), os.path.join(_R, 'tests'), os.path.join(_R, 'tests', 'golden')]
import torch, numpy as np
import stove_oracle as O
import analytic_weights as AW
from helpers import load_golden, oracle_setup, t_
N6=dict(num_obj=6, debug_match_objects='greedy', overlap_beta=100.0, max_obj_scale=0.22)
CASES={'n3':dict(num_obj=3),'n6':N6,'grav3':dict(num_obj=3)}
def matched(c,codes,B,T):
    N=c.num_obj
    zm,zs=O.constrain_zp(c,codes.reshape(-1,8)); zm=zm.view(B,T,N,4); zs=zs.view(B,T,N,4)
    f=O.match_greedy if c.debug_match_objects=='greedy' else O.match_3only
    a,b,_=f(c,zm,zs)
    return a,b,zm,zs
def flips(c,params,x,noise,trials=12,seed=0):
    B,T=x.shape[:2]; N=c.num_obj
    codes=O.encoder_forward(c,params,O.bw_transform(x).flatten(end_dim=1)).view(B,T,N,8)
    a0,b0,zm,zs=matched(c,codes,B,T)
    g=torch.Generator().manual_seed(seed); worst=0.0
    for i in range(trials):
        a,b,_,_=matched(c,codes+noise*torch.randn(codes.shape,generator=g,dtype=codes.dtype),B,T)
        worst=max(worst,(a-a0).abs().max().item(),(b-b0).abs().max().item())
    return worst, zm[...,0].min().item(), zm[...,0].max().item(), zm[...,2:4].min().item(), zm[...,2:4].max().item(), zs.min().item(), zs.max().item()
if __name__=='__main__':
    regime=sys.argv[1]
    sets=[None]
    for st in sets:
        for name,kw in CASES.items():
            g=load_golden(f'g7_stove_{name}_f64' if regime=='analytic' else f'g7_stove_{name}_{regime}_f64')
            c,structs,params=oracle_setup(torch.float64,requires_grad=False,regime=regime,**kw)
            if st:
                for k,amp in zip(('sup.encoder.rnn.weight_ih_l0','sup.encoder.rnn.weight_hh_l0','sup.encoder.fc1.weight','sup.encoder.fc2.weight','sup.encoder.fc2.bias'),st):
                    params[k]=torch.from_numpy(AW._m._sine(k,tuple(params[k].shape),amp,0.0).reshape(tuple(params[k].shape)))
            r=flips(c,params,t_(g['x']),3e-5)
            print(regime,st,name,'max change of matched states under 3e-5 code noise %.2e | sx %.3f..%.3f pos %.3f..%.3f std %.1e..%.2f'%r)
