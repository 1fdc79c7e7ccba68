"""The drop-in boundary and the separation rules (CPU only, no compute calls on a GPU):
  * libqsparse_hip.so loads and exports every symbol include/qsparse_hip.h declares;
  * the product package never touches oracle/ or the reference checkout;
  * GPU work without the library fails loudly instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "qsparse_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(qs_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    from qsparse_amd import _hip

    ge.build_hip()
    lib = ctypes.CDLL(_hip.lib_path())
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/qsparse_hip.h but not exported"
    assert sorted(_hip.SIGNATURES) == declared, "ctypes prototypes out of sync with the header"
    lib.qs_version.restype = ctypes.c_int
    assert lib.qs_version() == _hip.ABI_VERSION
    lib.qs_status_string.restype = ctypes.c_char_p
    assert b"aligned" in lib.qs_status_string(-3)


def test_argument_validation_without_a_gpu():
    """rejected arguments return QS_ERR_* before anything is enqueued, so this is safe on a CPU-only host"""
    from qsparse_amd import _hip

    lib = _hip.load()
    assert lib.qs_quant_scaler_fwd(None, None, None, None, 1, 0.1, None, 1, 1, 8, 0, 0, 0, 0, 0, 0, 0, 0, None, None, 0, None, None) == -2
    assert lib.qs_quant_scaler_fwd(16, 32, None, None, 1, 0.1, None, 1, 1, 8, 5, 0, 0, 0, 0, 0, 0, 0, None, None, 0, None, None) == -1
    assert lib.qs_quant_scaler_fwd(20, 32, None, None, 1, 0.1, None, 1, 1, 8, 0, 0, 0, 0, 0, 0, 0, 0, None, None, 0, None, None) == -3
    assert lib.qs_kth_value(16, 10, 10, 32, None, 0, None) == -2          # k out of range
    assert lib.qs_pq_select(16, None, 0, 70000, 0, 0, 0, 0, 32, None, 1, 0, 0, 4, None, None, None, None, None, None, None, 0, None, 1, None, None) == -2
    # the stage entry points of ABI v22 / v23: a plan that names no order, more kept dims than the kernel takes, the vectorised inner
    # order on a strided dim, a split coordinate outside the kept dims, a cascade prefix longer than the row
    import ctypes
    kept = (ctypes.c_int64 * 3)(8, 1, 1)
    base = ctypes.addressof(kept)
    assert lib.qs_mean_strided(16, 32, 8, 4, 1, base, base + 8, base + 16, 3, -1, 0, 0, 0, 0, None, None) == -2
    assert lib.qs_mean_strided(16, 32, 8, 4, 7, base, base + 8, base + 16, 1, -1, 0, 0, 0, 0, None, None) == -2
    assert lib.qs_mean_strided(16, 32, 8, 4, 1, base, base + 8, base + 16, 0, -1, 0, 0, 0, 0, None, None) == -2
    assert lib.qs_mean_strided(16, 32, 8, 4, 1, base, base + 8, base + 16, 2, 1, 4, 0, 0, 0, None, None) == -2
    assert lib.qs_mean_strided(16, 32, 8, 4, 1, base, base + 8, base + 16, 1, -1, 0, 0, 1, 0, None, None) == -1       # bf16 result of an fp32 input
    assert lib.qs_mean_dim_split(16, 32, 1, 8, 64, 65, 0, 0, 0, None, None) == -2
    assert lib.qs_mean_dim_split(16, 32, 1, 8, 1, 0, 0, 0, 0, None, None) == -2
    # the mailbox exchange (ABI v24): sizes, a world the flag header cannot hold, missing pointers, step 0 (steps count from 1)
    assert lib.qs_mailbox_bytes(2, 32) == (64 + 2 * 2 * 32) * 4 and lib.qs_mailbox_bytes(33, 32) == 0 and lib.qs_mailbox_bytes(2, 0) == 0
    assert lib.qs_mailbox_alloc(0, None) == -2 and lib.qs_mailbox_export(None, None) == -2 and lib.qs_mailbox_open(None, None) == -2
    boxes = (ctypes.c_void_p * 2)(16, None)
    assert lib.qs_mailbox_publish(16, 32, boxes, 2, 0, 1, None) == -2          # a peer that was never mapped
    boxes[1] = 32
    assert lib.qs_mailbox_publish(16, 32, boxes, 2, 2, 1, None) == -2          # rank outside the world
    assert lib.qs_mailbox_publish(16, 32, boxes, 2, 0, 0, None) == -2
    assert lib.qs_mailbox_wait(16, 2, 32, 1, None, 1000, None, None) == -2      # no status word
    assert lib.qs_mailbox_wait(16, 2, 32, 1, 48, 0, None, None) == -2           # a wait that may not poll at all


def test_product_never_imports_the_oracle_or_the_reference():
    pkg = os.path.join(ROOT, "qsparse_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if not f.endswith((".py", ".h", ".hip")):
                continue
            text = open(os.path.join(dirpath, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
            assert "qs_oracle" not in text and "/root/reference" not in text, f
    for f in ("bench.py", "__graft_entry__.py"):
        text = open(os.path.join(ROOT, f)).read()
        assert "/root/reference" not in text


def test_gpu_tensors_fail_loudly_without_the_library(monkeypatch, tmp_path):
    from qsparse_amd import _hip

    monkeypatch.setenv("QSPARSE_HIP_LIB", str(tmp_path / "missing.so"))
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(_hip.QsparseHipError):
        _hip.load()
    assert not _hip.available()
    # a host tensor can never be handed to a kernel
    with pytest.raises(_hip.QsparseHipError):
        _hip._ptr(torch.zeros(4))


@pytest.mark.gpu
def test_integration_md_binding_sketch_runs_as_written():
    """the ctypes stub INTEGRATION.md shows a maintainer (reference-side binding of qs_quant_scaler_fwd) is executed
    verbatim against the built library and checked against the oracle."""
    import re
    import torch
    from oracle import qs_oracle as O
    from qsparse_amd import _hip

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(import ctypes, torch\n.*?)```", text, re.S).group(1)
    code = code.replace('ctypes.CDLL("libqsparse_hip.so")', f'ctypes.CDLL("{_hip.lib_path()}")')
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    x = (torch.randn(4, 8, 7, 7, generator=torch.Generator().manual_seed(0)) * 2).bfloat16()
    s = torch.rand(8, 1, generator=torch.Generator().manual_seed(1)) * 0.1 + 0.01
    assert torch.equal(ns["scaler_fwd"](x.cuda(), s.cuda(), 1).cpu(), O.scaler_fwd(x, 8, s, 1))
