"""The ATen-order contract on the GPU BOX.  The staged-mean kernels (qs_reduce.h) restate the summation order
of ONE torch version's SumKernel.cpp; the fixtures record which (`meta["torch"]`), the package pins it (`PINNED_TORCH`) and warns
under another.  A torch upgrade on the GPU image must be caught HERE -- by the CPU probes of that order, which need no GPU but
run in the `-m gpu` set -- and not in a mask bit three layers later."""
import glob
import json
import os
import warnings

import numpy as np
import pytest
import torch

import qsparse_amd as qs
import test_aten_contract as contract

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_running_torch_is_the_one_the_fixtures_and_the_kernels_were_pinned_to():
    running = ".".join(torch.__version__.split("+")[0].split(".")[:2])
    assert running == qs.PINNED_TORCH, f"torch {torch.__version__} on this box, summation order pinned to {qs.PINNED_TORCH}"
    for f in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "*.npz"))):
        meta = json.loads(str(np.load(f)["meta"]))
        assert ".".join(meta["torch"].split("+")[0].split(".")[:2]) == qs.PINNED_TORCH, f
        assert meta["intra_op_threads"] == 1, f


def test_sumkernel_order_probes_hold_on_this_box():
    contract.test_aten_reduces_h_of_a_channels_last_sample_like_the_batch_of_its_rows()
    contract.test_aten_reduces_w_of_a_channels_last_tensor_in_row_sum_order()
    contract.test_aten_reduce_plan_names_the_order_aten_takes_for_any_dense_layout()


def test_another_torch_warns_once_at_import(monkeypatch):
    from qsparse_amd import util
    monkeypatch.setattr(util, "_torch_pin_warned", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        util.check_torch_pin("2.12.1+rocm9")
        util.check_torch_pin("2.12.1+rocm9")
    assert len(w) == 1 and "summation order" in str(w[0].message)
    monkeypatch.setattr(util, "_torch_pin_warned", False)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        util.check_torch_pin(qs.PINNED_TORCH + ".0+rocm7.0")
    assert not w
