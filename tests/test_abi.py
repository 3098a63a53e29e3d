"""CPU-side checks of the C-ABI boundary: the library builds for gfx950, loads, and exports every
symbol include/tsgnn.h declares (no compute calls here — there is no GPU in this container)."""
import ctypes
import os

import pytest


def test_header_parses_and_library_exports_every_symbol():
    from two_stage_gnn_amd import _native as nat
    decls = nat.parse_header()
    assert len(decls) >= 20 and "tsgnn_csr_spmm_f32" in decls
    nat.build()
    L = ctypes.CDLL(nat.LIB_PATH)
    missing = [n for n in decls if not hasattr(L, n)]
    assert not missing, missing
    assert nat.lib().tsgnn_abi_version() == 1
    assert b"invalid" in nat.lib().tsgnn_strerror(-1)


def test_invalid_arguments_are_rejected_without_a_gpu():
    from two_stage_gnn_amd import _native as nat
    L = nat.lib()
    # null pointers / negative sizes must come back as TSGNN_EINVAL before any launch
    assert L.tsgnn_csr_spmm_f32(None, None, None, None, None, 4, None, 4, 10, 4, 0.0, 0, 0, None) == -1
    assert L.tsgnn_linear_l2norm_f32(None, 1, None, 1, None, None, 1, None, 1, 1, 1, 1, None) == -1
    assert L.tsgnn_exclusive_scan_i32(None, -1, None, None, None) == -1


def test_product_path_refuses_cpu_tensors():
    import torch
    from two_stage_gnn_amd import message_passing as mp
    with pytest.raises(RuntimeError, match="GPU only"):
        mp.linear_l2norm(torch.zeros(4, 4), torch.zeros(4, 4))


def test_product_never_imports_oracle():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "two-stage-gnn_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dp, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f
