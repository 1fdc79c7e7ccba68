"""Out-of-bounds WRITE check for the HIP kernels (there is no GPU address sanitizer on this pool): every buffer the
ctypes wrappers in qsparse_amd/_hip.py allocate -- kernel outputs, staging buffers, workspaces, accumulators -- is
carved out of a larger allocation whose margins hold a byte pattern, the randomised differential tests of tests/fuzz/
(odd shapes, ragged rows, channels_last, every dtype) run on top of it, and after every case the margins must be intact.
The results themselves are checked by the fuzz cases as usual."""
import importlib.util
import math
import os
import random

import pytest
import torch

import qsparse_amd as qs
from qsparse_amd import _hip

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PAD = 256          # elements on either side (keeps 16-byte alignment for every dtype)
PATTERN = 0xA5


class _Guarded:
    """stands in for the `torch` module inside qsparse_amd._hip: allocation functions hand out guarded buffers"""

    def __init__(self):
        self.live = []
        self.count = 0

    def __getattr__(self, name):
        return getattr(torch, name)

    def _alloc(self, shape, strides, dtype, device):
        device = torch.device(device) if device is not None else torch.device("cpu")
        shape = tuple(int(s) for s in shape)
        if device.type != "cuda":
            return torch.empty_strided(shape, strides, dtype=dtype, device=device)
        numel, item = math.prod(shape), torch.empty(0, dtype=dtype).element_size()
        raw = torch.full(((numel + 2 * PAD) * item,), PATTERN, dtype=torch.uint8, device=device)
        self.live.append((raw, PAD * item, numel * item))
        self.count += 1
        return raw.view(dtype).as_strided(shape, strides, PAD)

    @staticmethod
    def _contiguous_strides(shape):
        strides, acc = [], 1
        for s in reversed(shape):
            strides.append(acc)
            acc *= max(int(s), 1)
        return tuple(reversed(strides))

    def empty(self, *size, dtype=None, device=None, memory_format=None):
        shape = tuple(size[0]) if len(size) == 1 and isinstance(size[0], (tuple, list, torch.Size)) else size
        strides = self._contiguous_strides(shape)
        if memory_format in (torch.channels_last, torch.channels_last_3d):
            strides = torch.empty(shape, device="meta").contiguous(memory_format=memory_format).stride()
        return self._alloc(shape, strides, dtype or torch.float32, device)

    def zeros(self, *size, dtype=None, device=None):
        return self.empty(*size, dtype=dtype, device=device).zero_()

    def empty_like(self, like, dtype=None):
        dense = (like.is_contiguous() or (like.dim() == 4 and like.is_contiguous(memory_format=torch.channels_last))
                 or (like.dim() == 5 and like.is_contiguous(memory_format=torch.channels_last_3d)))
        strides = like.stride() if dense else self._contiguous_strides(like.shape)
        return self._alloc(like.shape, strides, dtype or like.dtype, like.device)

    def check(self):
        torch.cuda.synchronize()
        for raw, pad_bytes, body_bytes in self.live:
            head, tail = raw[:pad_bytes], raw[pad_bytes + body_bytes:]
            assert bool((head == PATTERN).all()), "a kernel wrote in front of its buffer"
            assert bool((tail == PATTERN).all()), "a kernel wrote behind its buffer"
        self.live.clear()


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "tests", "fuzz", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture
def guarded(monkeypatch):
    before = {k: qs.get_qsparse_option(k) for k in ("log_on_created", "log_during_train", "fold_relu", "preserve_dtype", "graph_safe", "elide_pruned", "relu_gate")}
    threads = torch.get_num_threads()
    qs.set_qsparse_options(log_on_created=False, log_during_train=False)
    torch.set_num_threads(1)
    g = _Guarded()
    monkeypatch.setattr(_hip, "torch", g)
    yield g
    torch.set_num_threads(threads)
    qs.set_qsparse_options(**before)


def test_guard_catches_a_stray_write(guarded):
    y = guarded.empty(10, dtype=torch.float32, device="cuda")
    guarded.check()
    y = guarded.empty(10, dtype=torch.float32, device="cuda")
    y.as_strided((11,), (1,), y.storage_offset())[10] = 1.0       # one element past the end
    with pytest.raises(AssertionError):
        guarded.check()


def test_no_kernel_writes_outside_its_buffers_sites(guarded):
    fz = _load("fuzz_parity")
    rng = random.Random(31)
    for i in range(100):
        r = fz.one_case(rng, i)
        assert r in ("ok", None), str(r)
        guarded.check()
    assert guarded.count > 500


def test_no_kernel_writes_outside_its_buffers_modules_and_functional(guarded):
    fz = _load("fuzz_cpu_gpu")
    rng = random.Random(32)
    for i in range(50):
        r = fz.one_case(rng, i)
        assert r in ("ok", None), r
        guarded.check()
    for i in range(200):
        r = fz.one_functional(rng, i)
        assert r in ("ok", None), r
        guarded.check()
    assert guarded.count > 500


def test_multi_tensor_weight_path_stays_inside_its_flat_buffer(guarded, monkeypatch):
    from qsparse_amd import batch
    monkeypatch.setattr(batch, "torch", guarded)
    torch.manual_seed(0)
    net = torch.nn.Sequential(
        qs.quantize(torch.nn.Conv2d(3, 5, 3), bits=8, timeout=1, channelwise=-1),            # 135 weights: not a multiple of 8
        torch.nn.ReLU(),
        qs.quantize(torch.nn.Conv2d(5, 7, 1), bits=4, timeout=2, channelwise=-1, callback=qs.DecimalQuantizer()),   # 35
        torch.nn.Flatten(),
        qs.quantize(torch.nn.Linear(7 * 6 * 6, 3), bits=8, timeout=1, channelwise=-1),       # 756
    ).cuda().train()
    wb = qs.WeightBatcher(net)
    assert len(wb.layers) == 3
    x = torch.randn(2, 3, 8, 8, device="cuda")
    for _ in range(4):
        net(x).sum().backward()
        guarded.check()
    net.eval()
    for _ in range(2):
        net(x)
        guarded.check()
    assert guarded.count > 3


def test_composite_sites_images_and_token_statistics_stay_inside_their_buffers(guarded, monkeypatch):
    """the buffers the HOST modules allocate for the composite calls -- outputs, gate bitmaps, autocast images, the round-6 riders'
    gradient image, the token-major statistics' per-column keys and stages -- guarded the same way: ragged maps, channel counts
    that are no multiple of 8 / 32, token-major sites on both statistics routes, a bottleneck data flow under autocast"""
    import torch.nn as nn
    import sys
    from qsparse_amd import fused
    from qsparse_amd.fused import fuse_prune_quantize_pairs
    # (`qsparse_amd.quantize` / `.prune` the ATTRIBUTES are functions: the modules come from sys.modules)
    for mod in (fused, sys.modules["qsparse_amd.quantize"], sys.modules["qsparse_amd.sparse"]):
        monkeypatch.setattr(mod, "torch", guarded)

    def pair(dim=1):
        return fuse_prune_quantize_pairs(nn.Sequential(
            nn.Sequential(nn.ReLU(), qs.prune(sparsity=0.5, dimensions={dim}, start=1, interval=1, repetition=1)),
            qs.quantize(bits=4, channelwise=-1, timeout=1))).cuda().train()

    g = torch.Generator().manual_seed(5)
    for shape, dim, dtype in (((4, 7, 40), 2, torch.bfloat16), ((3, 5, 96), 2, torch.float32), ((6, 3, 33), 2, torch.float16),
                              ((5, 24, 7, 7), 1, torch.bfloat16), ((4, 40, 3, 5), 1, torch.float32), ((9, 48), 1, torch.bfloat16)):
        site = pair(dim)
        for step in range(5):
            x = (torch.randn(shape, generator=g) * 2).to(dtype).cuda().requires_grad_(True)
            y = site(x)
            y.backward(torch.randn(shape, generator=g).to(y.dtype).cuda())
            guarded.check()
    # a bottleneck's data flow under autocast: first / second image, promoting add with the gradient image, ragged 7 x 7 maps
    for down in (False, True):
        s0, s1 = pair(), pair()
        conv1, conv2, head = (nn.Conv2d(24, 24, 1, bias=False).cuda() for _ in range(3))
        for step in range(5):
            x = (torch.randn(4, 24, 7, 7, generator=g) * 2).cuda().requires_grad_(True)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                y0 = s0(x)
                out = conv1(y0)
                y1 = s1(out + (conv2(y0) if down else y0))
                loss = head(y1).float().sum()
            loss.backward()
            guarded.check()
    assert guarded.count > 150
