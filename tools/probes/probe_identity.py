import os, sys, torch
sys.path.insert(0, os.getcwd())
from qsparse_amd import _hip
torch.manual_seed(0)
ident = _hip.try_activation(_hip.ACT_LEAKY, 1.0)
def bits(t): return t.contiguous().view(torch.int16 if t.element_size()==2 else torch.int32)
for dtype in (torch.float16, torch.bfloat16, torch.float32):
    for cl in (False, True):
        shape=(8,16,14,14)
        x=(torch.randn(shape)*2).to(dtype)
        x.view(-1)[:6]=torch.tensor([0.0,-0.0,1e-7,-1e-7,6e-8,-3e-5]).to(dtype)
        x=x.cuda()
        g=(torch.randn(shape)*3).cuda()
        g.view(-1)[:4]=torch.tensor([0.0,-0.0,1e-9,-1e-9]).cuda()
        if cl: x=x.contiguous(memory_format=torch.channels_last); g=g.contiguous(memory_format=torch.channels_last)
        mask=(torch.rand(16)>0.4).cuda()
        scale=torch.tensor([[0.37]]).cuda()
        y0,_=_hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1)
        y1,_,gate=_hip.quant_fwd("scaler", x, scale, -1, torch.float32, chan_mask=mask, mask_channel_index=1, pre_relu=ident, want_gate=True)
        print(str(dtype)[6:], cl, "fwd equal", torch.equal(bits(y0),bits(y1)), int((bits(y0)!=bits(y1)).sum()))
        for gdt in (torch.float32, dtype):
            gg=g.to(gdt)
            a=_hip.ste_bwd(gg, scale, False, -1, -8.0, 7.0, False, dtype if gdt==torch.float32 else gdt, chan_mask=mask, mask_channel_index=1)
            b=_hip.ste_relu_bwd(gg, None, scale, False, -8.0, 7.0, mask, mask_channel_index=1, gate=gate, act=ident)
            c=_hip.ste_relu_bwd(gg, x, scale, False, -8.0, 7.0, mask, mask_channel_index=1, act=ident)
            print("   bwd g", str(gdt)[6:], "gate==plain", torch.equal(bits(a),bits(b)), int((bits(a)!=bits(b)).sum()), " x==plain", torch.equal(bits(a),bits(c)), int((bits(a)!=bits(c)).sum()))
            if not torch.equal(bits(a),bits(b)):
                idx=(bits(a)!=bits(b)).flatten().nonzero().flatten()[:5]
                print("      ", [(float(a.flatten()[i]), float(b.flatten()[i]), float(gg.flatten()[i]), float(x.flatten()[i])) for i in idx])
