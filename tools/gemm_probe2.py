import torch, time
dev='cuda:0'
n=25600
dg=torch.randn(n,1024,device=dev); hs=torch.randn(n,256,device=dev)
dg2=torch.randn(2*n,1024,device=dev); hs2=torch.randn(2*n,256,device=dev)
def bench(f,name):
    for _ in range(3): f()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(20): f()
    torch.cuda.synchronize(); print('%-40s %.1f us' % (name,(time.perf_counter()-t)/20*1e6))
bench(lambda: torch.mm(dg.t(), hs), 'mm(dg.T, hs)  (1024x256)')
bench(lambda: torch.mm(hs.t(), dg), 'mm(hs.T, dg)  (256x1024)')
bench(lambda: torch.mm(dg2.t(), hs2), 'mm(dg2.T, hs2) both steps')
bench(lambda: torch.mm(hs2.t(), dg2), 'mm(hs2.T, dg2) both steps')
dgT=dg.t().contiguous()
bench(lambda: torch.mm(dgT, hs), 'mm(dgT_contig, hs)')
bench(lambda: dg.t().contiguous(), 'transpose copy dg')
