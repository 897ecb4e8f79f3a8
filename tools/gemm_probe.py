import torch, time
dev='cuda:0'
n=25600
dg=torch.randn(n,1024,device=dev); w_hh=torch.randn(1024,256,device=dev); dhs=torch.randn(n,256,device=dev)
w_hh_t=w_hh.t().contiguous()
x=torch.randn(n,1024,device=dev); w_ih=torch.randn(1024,1024,device=dev); hs=torch.randn(n,256,device=dev)
def bench(f,name):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print('%-40s %.1f us' % (name,(time.perf_counter()-t)/20*1e6))
bench(lambda: torch.addmm(dhs, dg, w_hh), 'dh = addmm(dhs, dg, w_hh) [NN]')
bench(lambda: torch.addmm(dhs, dg, w_hh_t.t()), 'dh = addmm(dhs, dg, w_hh_t.T) [NT]')
bench(lambda: torch.mm(dg, w_hh), 'mm(dg, w_hh)')
bench(lambda: torch.nn.functional.linear(dg, w_hh_t, None), 'linear(dg, w_hh_t)')
bench(lambda: torch.mm(w_hh_t, dg.t()).t(), '(w_hh_t @ dg.T).T')
bench(lambda: torch.mm(dg.t(), hs), 'dWhh: mm(dg.T, hs)')
bench(lambda: torch.mm(hs, w_hh.t()), 'gh: mm(hs, w_hh.T)')
bench(lambda: torch.mm(x, w_ih.t()), 'gx: mm(x, w_ih.T)')
bench(lambda: torch.mm(dg.t(), x), 'dWih: mm(dg.T, x)')
bench(lambda: torch.mm(x.t(), dg).t(), 'dWih: mm(x.T, dg).T')
bench(lambda: dg.sum(0), 'dg.sum(0)')
ones=torch.ones(n,device=dev)
bench(lambda: torch.mv(dg.t(), ones), 'mv(dg.T, ones)')
