import torch, time
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e3
N=273744
a=torch.empty(N,576,device='cuda'); b=torch.randn(N,576,device='cuda')
print('fill 630MB ms', t(lambda: a.fill_(1.0)), 'GB/s', 0.63/ (t(lambda: a.fill_(1.0))/1e3))
print('copy 630MB ms', t(lambda: a.copy_(b)))
x=torch.randn(N,64,device='cuda'); w=torch.randn(64,576,device='cuda')
print('torch mm [N,64]x[64,576] ms', t(lambda: torch.mm(x,w,out=a)))
w2=torch.randn(64,16,device='cuda'); o=torch.empty(N,16,device='cuda')
bv=b.view(N,9,64)
print('torch 9x mm strided [N,64]x[64,16] ms', t(lambda: [torch.mm(bv[:,p,:],w2,out=o) for p in range(9)]))
c=torch.empty(N,64,device='cuda')
print('copy strided slice 70MB ms', t(lambda: c.copy_(bv[:,3,:])))
