import torch
g = torch.Generator().manual_seed(0)
x = (torch.randn(256, 96, 3, 3, generator=g, dtype=torch.float64) * 1.5)
def imp(x):
    a = x.abs()
    a = a.mean(2, keepdim=True)
    a = a.mean(3, keepdim=True)
    return a
c = imp(x); d = imp(x.cuda()).cpu()
print("fp64 imp equal:", torch.equal(c, d), "max rel diff", ((c - d).abs() / c.abs()).max().item(), "n diff", int((c != d).sum()))
mag = torch.rand(256, 96, 1, 1)
for t in (0, 1, 2):
    r1 = ((t * mag + c) / (t + 1)).float()
    r2 = ((t * mag.cuda() + d.cuda()) / (t + 1)).float().cpu()
    r3 = ((t * mag.cuda() + c.cuda()) / (t + 1)).float().cpu()
    print(t, "flips with gpu imp:", int((r1 != r2).sum()), " flips with the SAME imp (gpu arithmetic only):", int((r1 != r3).sum()))
m1 = mag.clone(); m1.data[:] = (2 * m1 + c) / 3
m2 = mag.clone().cuda(); m2.data[:] = (2 * m2 + c.cuda()) / 3
print("assign flips:", int((m1 != m2.cpu()).sum()))
