import os, sys, torch, time
sys.path.insert(0, os.getcwd())
from qsparse_amd import _hip
def us(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    evs=[]
    for _ in range(n):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); evs.append((a,b))
    torch.cuda.synchronize()
    ts=sorted(a.elapsed_time(b) for a,b in evs); return ts[len(ts)//2]*1e3
for shape in ((256,197,3072),(3072,197,256),(256*197,3072),(256,197,4096),(256,197,2048),(256,788,768)):
    C=shape[-1]
    xs=[torch.randn(shape, device="cuda").bfloat16() for _ in range(2)]
    mask=(torch.rand(C, device="cuda")>0.75)
    scale=torch.tensor([[0.37]], device="cuda")
    i=[0]
    def f():
        i[0]+=1
        _hip.quant_fwd("scaler", xs[i[0]%2], scale, -1, torch.float32, chan_mask=mask, mask_channel_index=len(shape)-1, pre_relu=True, want_gate=True)
    t=us(f); n=xs[0].numel()
    print(shape, f"fwd(gate) {t:7.1f} us  {n*6.125/t/1e3:6.0f} GB/s", flush=True)
