"""GPU probe: is ATen's fp16 gelu_backward the same function of (dy, x) in a full block of its vectorised kernel and in the tail
block?  The same 1,000 (dy, x) pairs evaluated (a) at the start of a tensor of 2^20 elements, (b) as a tensor of 1,000 elements."""
import torch
dev = "cuda"
for dt in (torch.float16, torch.bfloat16, torch.float32):
    g = torch.Generator().manual_seed(0)
    n_bad = 0
    for trial in range(20):
        x = (torch.randn(1 << 20, generator=g) * 2).to(dt).to(dev)
        dy = (torch.randn(1 << 20, generator=g)).to(dt).to(dev)
        full = torch.ops.aten.gelu_backward(dy, x)[:1000]
        tail = torch.ops.aten.gelu_backward(dy[:1000].clone(), x[:1000].clone())
        n_bad += int((full.view(torch.int16 if dt != torch.float32 else torch.int32) != tail.view(torch.int16 if dt != torch.float32 else torch.int32)).sum())
    print(dt, "gelu_backward: differing results between a full block and a tail block:", n_bad, "of 20000")
    n_bad = 0
    for trial in range(20):
        x = (torch.randn(1 << 20, generator=g) * 2).to(dt).to(dev)
        full = torch.nn.functional.gelu(x)[:1000]
        tail = torch.nn.functional.gelu(x[:1000].clone())
        n_bad += int((full.view(torch.int16 if dt != torch.float32 else torch.int32) != tail.view(torch.int16 if dt != torch.float32 else torch.int32)).sum())
    print(dt, "gelu: differing:", n_bad, "of 20000")
