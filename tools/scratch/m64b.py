import torch
g = torch.Generator().manual_seed(4000 + 774)
x = (torch.randn(256, 96, 3, 3, generator=g) * 1.5).to(torch.float64)
def imp(x):
    a = x.abs()
    a = a.mean(2, keepdim=True)
    a = a.mean(3, keepdim=True)
    return a
c = imp(x); d = imp(x.cuda()).cpu()
print("n diff fp64", int((c != d).sum()), "max rel", ((c - d).abs() / c.abs()).max().item())
print("fp32 flips", int((c.float() != d.float()).sum()))
# where do they flip: is the exact mean a float32 midpoint?
idx = (c.float() != d.float()).view(-1).nonzero().view(-1)[:5]
for i in idx.tolist():
    print(i, repr(c.view(-1)[i].item()), repr(d.view(-1)[i].item()), repr(c.view(-1)[i].float().item()), repr(d.view(-1)[i].float().item()))
