"""loader for the fixtures written by tests/golden/generate.py (data only; see its docstring)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.meta = json.loads(str(self.z["meta"]))
        self.cases = self.meta["cases"]

    def has(self, key):
        return any(k in self.z for k in (key, key + "__bf16", key + "__f16"))

    def get(self, key):
        """tensor stored under `key` (bf16 restored from its bit pattern)."""
        if key + "__bf16" in self.z:
            a = self.z[key + "__bf16"].view(np.int16)
            return torch.from_numpy(a.copy()).view(torch.bfloat16)
        if key + "__f16" in self.z:
            return torch.from_numpy(self.z[key + "__f16"].copy())
        a = self.z[key]
        if a.ndim == 0:
            return a.item()
        return torch.from_numpy(a.copy())


def tdtype(name):
    return getattr(torch, name)


def same(a, b):
    """bit-for-bit equality of two tensors (same dtype, same shape, NaNs compare by payload)."""
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype == torch.bool:
        return bool((a == b).all())
    return bool(torch.equal(a.contiguous().reshape(-1).view(torch.uint8), b.contiguous().reshape(-1).view(torch.uint8)))


def same_up_to_nan_payload(a, b):
    """`same`, except that two NaNs match whatever their sign and payload (x86 and the GPU produce different default NaNs:
    Inf * 0 is 0xffc00000 on the one, 0x7fc00000 on the other)."""
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if not a.is_floating_point():
        return same(a, b)
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return bool(((a.contiguous().view(it) == b.contiguous().view(it)) | (a.isnan() & b.isnan())).all())

