"""GPU probe: hand-written erf-GELU (this image's hipcc, -ffp-contract=off) against ATen's GPU gelu / gelu_backward.
every bf16 and fp16 bit pattern (computed in float, rounded to the dtype as ATen does) + 2^24 random float32 values"""
import ctypes, os, sys
import torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "probe_gelu_bits.so"))
P, L, I = ctypes.c_void_p, ctypes.c_long, ctypes.c_int
lib.gelu_fwd.argtypes = [P, P, L, P]; lib.gelu_bwd.argtypes = [P, P, P, L, I, P]
dev = torch.device("cuda", 0)
st = lambda: torch.cuda.current_stream().cuda_stream
def bits(a):
    return a.view(torch.int16 if a.element_size() == 2 else torch.int32)
def same(a, b):
    return bool(((bits(a) == bits(b)) | (a.isnan() & b.isnan())).all())
def ndiff(a, b):
    return int((~((bits(a) == bits(b)) | (a.isnan() & b.isnan()))).sum())
for dt in (torch.bfloat16, torch.float16):
    allx = torch.arange(65536, dtype=torch.int32, device=dev).to(torch.int16).view(dt)
    xf = allx.float().contiguous()
    y = torch.empty_like(xf)
    lib.gelu_fwd(xf.data_ptr(), y.data_ptr(), xf.numel(), st()); torch.cuda.synchronize()
    want = torch.nn.functional.gelu(allx)
    print(dt, "fwd differing patterns:", ndiff(y.to(dt), want), "(float results vs float gelu:", ndiff(y, torch.nn.functional.gelu(xf)), ")")
    g = torch.Generator(device="cpu").manual_seed(1)
    for trial in range(4):
        dy = (torch.randn(65536, generator=g) * (10.0 ** (trial - 2))).to(dt).to(dev)
        if trial == 0:
            dy = torch.ones(65536, dtype=dt, device=dev)
        want = torch.ops.aten.gelu_backward(dy, allx)
        for variant in (0, 1, 2):
            dx = torch.empty_like(xf)
            lib.gelu_bwd(dy.float().contiguous().data_ptr(), xf.data_ptr(), dx.data_ptr(), xf.numel(), variant, st()); torch.cuda.synchronize()
            print(dt, "bwd trial", trial, "variant", variant, "differing:", ndiff(dx.to(dt), want))
g = torch.Generator(device="cpu").manual_seed(2)
xf = (torch.randn(1 << 24, generator=g) * 3).to(dev)
y = torch.empty_like(xf)
lib.gelu_fwd(xf.data_ptr(), y.data_ptr(), xf.numel(), st()); torch.cuda.synchronize()
print("float32 fwd differing:", ndiff(y, torch.nn.functional.gelu(xf)), "of", xf.numel())
dy = torch.randn(1 << 24, generator=g).to(dev)
want = torch.ops.aten.gelu_backward(dy, xf)
for variant in (0, 1, 2):
    dx = torch.empty_like(xf)
    lib.gelu_bwd(dy.data_ptr(), xf.data_ptr(), dx.data_ptr(), xf.numel(), variant, st()); torch.cuda.synchronize()
    print("float32 bwd variant", variant, "differing:", ndiff(dx, want))
