import torch
torch.manual_seed(0)
x=(torch.randn(8,16,14,14)*2).half().cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
h=torch.nn.functional.gelu(x)
g=(torch.randn(8,16,14,14)*0.01).half().cuda()
g.view(-1)[:200]=0.0; g.view(-1)[200:400]=-0.0
g_cl=g.contiguous(memory_format=torch.channels_last)
(a,)=torch.autograd.grad(h,x,g,retain_graph=True)
(b,)=torch.autograd.grad(h,x,g_cl,retain_graph=True)
print("strides", a.stride(), b.stride())
ai,bi=a.contiguous().view(torch.int16),b.contiguous().view(torch.int16)
print("bit diffs", int((ai!=bi).sum()), "numeric diffs", int((a.float()!=b.float()).sum()))
