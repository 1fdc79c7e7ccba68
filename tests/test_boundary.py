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
    lib.qs_abi_floor.restype = ctypes.c_int
    assert lib.qs_abi_floor() <= _hip.ABI_VERSION <= lib.qs_version()        # the loader's rule (ABI compatibility, v25)
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


def test_loader_accepts_any_library_that_honours_the_bindings_version():
    """`_hip.load()` no longer hard-fails on any version difference.  From v25 on positional prototypes are frozen
    and new operands arrive through size-prefixed descriptors, so a library serves a binding written against V whenever
    qs_abi_floor() <= V <= qs_version()"""
    from qsparse_amd import _hip
    assert _hip.abi_compatible(found=25, floor=25, needed=25)
    assert _hip.abi_compatible(found=31, floor=25, needed=25)         # a NEWER library: fields were appended, nothing moved
    assert _hip.abi_compatible(found=31, floor=25, needed=28)
    assert not _hip.abi_compatible(found=24, floor=24, needed=25)     # older than the binding: symbols / fields missing
    assert not _hip.abi_compatible(found=40, floor=30, needed=25)     # a library that dropped the v25 prototypes
    lib = _hip.load()
    assert lib.qs_abi_floor() == 25 and _hip.abi_compatible(lib.qs_version(), lib.qs_abi_floor())


def test_descriptor_structs_have_the_headers_layout(tmp_path):
    """the ctypes mirrors of the descriptor structs and caller-built tables against `sizeof` / `offsetof` of the header itself
    (compiled here with gcc: the header is plain C)"""
    import subprocess
    from qsparse_amd import _hip
    pairs = {"qs_quant_fwd_args": _hip.QuantFwdArgs, "qs_pq_select_args": _hip.PqSelectArgs, "qs_site_fwd_args": _hip.SiteFwdArgs,
             "qs_quantize_step_args": _hip.QuantizeStepArgs, "qs_site_bwd_args": _hip.SiteBwdArgs,
             "qs_ste_relu_bwd_args": _hip.SteReluBwdArgs, "qs_site_plan": _hip.SitePlanStruct, "qs_multi_row": _hip.MultiRow}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "qsparse_hip.h")}"', 'int main(void) {']
    for cname, ct in pairs.items():
        lines.append(f'printf("{cname} %zu", sizeof({cname}));')
        for fname, _ in ct._fields_:
            lines.append(f'printf(" %zu", offsetof({cname}, {fname}));')
        lines.append('printf("\\n");')
    lines += ['return 0; }']
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-o", str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines()
    assert len(out) == len(pairs)
    for line, (cname, ct) in zip(out, pairs.items()):
        name, size, *offsets = line.split()
        assert name == cname and int(size) == ctypes.sizeof(ct), (cname, size, ctypes.sizeof(ct))
        assert [int(o) for o in offsets] == [getattr(ct, f).offset for f, _ in ct._fields_], cname
        assert ct._fields_[0][0] in ("struct_size", "N", "x")


def test_descriptor_entry_points_validate_without_a_gpu():
    """a descriptor that is missing, shorter than its size field, or names nothing to do is refused before anything is enqueued; a
    caller compiled against an OLDER header (smaller struct_size) is read up to what it knows"""
    from qsparse_amd import _hip
    lib = _hip.load()
    for fn in (lib.qs_quant_fwd_v, lib.qs_pq_select_v, lib.qs_quantize_step_v, lib.qs_quant_ste_relu_bwd_v):
        assert fn(None) == -2
    assert lib.qs_site_fwd_v(None, None) == -2 and lib.qs_site_bwd_v(None, None) == -2
    a = _hip.QuantFwdArgs()
    a.struct_size = 2                                   # not even the size field itself
    assert lib.qs_quant_fwd_v(ctypes.byref(a)) == -2
    a.struct_size = ctypes.sizeof(a)
    a.kind = 7
    assert lib.qs_quant_fwd_v(ctypes.byref(a)) == -2     # unknown quantizer kind
    a.kind, a.x, a.y, a.nparam, a.param_host, a.outer, a.C, a.inner, a.xdt = 0, 16, 32, 1, 0.1, 1, 1, 8, 5
    assert lib.qs_quant_fwd_v(ctypes.byref(a)) == -1     # the positional entry point's own checks: unknown dtype
    a.xdt, a.x = 0, 20
    assert lib.qs_quant_fwd_v(ctypes.byref(a)) == -3     # ... misaligned x
    b = _hip.SteReluBwdArgs()
    b.struct_size = _hip.SteReluBwdArgs.g3.offset        # a v24-era caller: the struct ends in front of the v25 fields
    b.g3 = 0xdead0                                       # (garbage behind the caller's struct must not be read)
    b.gx_image = 0xbeef0
    assert lib.qs_quant_ste_relu_bwd_v(ctypes.byref(b)) == -2      # no gradient, no gate: refused for THAT, not for the garbage
    b.g, b.gate, b.gx, b.nstep, b.step_host, b.outer, b.C, b.inner, b.gdt, b.xdt = 16, 32, 48, 1, 1.0, 0, 1, 8, 0, 1
    assert lib.qs_quant_ste_relu_bwd_v(ctypes.byref(b)) == 0       # an empty tensor: accepted (fp32 g, bf16 x -- the riders would be refused)
    b.struct_size = ctypes.sizeof(b)
    assert lib.qs_quant_ste_relu_bwd_v(ctypes.byref(b)) == -2      # the same struct read in full: g3 without g2 is an error


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
