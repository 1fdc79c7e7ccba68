import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from qsparse_amd import _hip
def us(f, n=50):
    for _ in range(5): f()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/n*1e6
for dtype in (torch.bfloat16, torch.float32):
    for T,C in ((197,3072),(1024,4096),(197,768),(50,1024)):
        x=torch.randn(T*C, device="cuda").to(dtype)
        a=us(lambda: _hip.mean_dim(x,1,T,C,dtype,0))
        print(os.environ.get("QS_MEAN_SPLIT","0"), os.environ.get("QS_MEAN_NARROW","1"), str(dtype)[6:], (T,C), f"{a:7.1f} us", flush=True)
