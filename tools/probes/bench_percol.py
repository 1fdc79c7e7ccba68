import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from qsparse_amd import _hip
def us(f, n=30):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for dtype in (torch.bfloat16, torch.float32):
    for N,T,C in ((256,197,3072),(64,1024,4096),(256,197,768)):
        post=T*C
        xs=[torch.randn(N*post, device="cuda").to(dtype) for _ in range(3)]
        acc=torch.zeros(C,32,device="cuda")
        fl=_hip.mean_flags(True, False)
        i=[0]
        def plain():
            i[0]+=1; _hip.mean_dim(xs[i[0]%3],1,N,post,dtype,fl)
        def rider():
            i[0]+=1; _hip.mean_dim(xs[i[0]%3],1,N,post,dtype,fl,absmax_out=acc,chan_div=1,C=C)
        nb=N*post*xs[0].element_size()
        a=us(plain); b=us(rider)
        print(os.environ.get("QS_MEAN_PERCOL","1"), str(dtype)[6:], (N,T,C), f"plain {a:7.1f} us ({nb/a/1e3:5.0f} GB/s)  rider {b:7.1f} us ({nb/b/1e3:5.0f} GB/s)", flush=True)
