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


def test_new_entry_points_validate_on_the_host():
    """round-2 entry points reject bad descriptions / shapes before any launch (no GPU needed)"""
    import numpy as np
    from two_stage_gnn_amd import _native as nat
    L = nat.lib()
    assert L.tsgnn_gat_fused_supported(4, 64) == 1 and L.tsgnn_gat_fused_supported(4, 6) == 0 and L.tsgnn_gat_fused_supported(9, 4) == 0
    assert L.tsgnn_gat_fused_supported(2, 256) == 0                                   # H * Fh / 4 lanes must fit one wave
    assert L.tsgnn_contract_dense_supported(64, 8, 192) == 1 and L.tsgnn_contract_dense_supported(512, 64, 192) == 0
    assert L.tsgnn_contract_rows_bwd_supported(64, 192) == 1 and L.tsgnn_contract_rows_bwd_supported(66, 192) == 0
    assert L.tsgnn_gat_bwd_parts(32) == 8 and L.tsgnn_gat_bwd_parts(1000) == 1
    assert L.tsgnn_gat_pack_desc_words() == 23 and L.tsgnn_sage_multi_tn_words() == 11 and L.tsgnn_sage_multi_g_words() == 20
    bad = np.zeros(64, dtype=np.int64)
    assert L.tsgnn_gat_pack_f32(bad.ctypes.data, None) == -1                          # zero layers
    assert L.tsgnn_sage_multi_f32(bad.ctypes.data, None) == -1                        # no problems
    assert L.tsgnn_wgrad_reduce_sets_f32(None, None) == -1
    ns, rps, need = np.zeros(1, np.int32), np.zeros(1, np.int64), np.zeros(1, np.int64)
    assert L.tsgnn_wgrad_blocks_plan(8518, 256, 264, 256, 264, ns.ctypes.data, rps.ctypes.data, need.ctypes.data) == 0
    assert ns[0] > 0 and ns[0] * rps[0] >= 8518 and rps[0] % 32 == 0 and need[0] == 6 * ns[0] * 129 * 128
    assert L.tsgnn_wgrad_blocks_plan(100, 600, 264, 600, 264, ns.ctypes.data, rps.ctypes.data, need.ctypes.data) == 0 and ns[0] == 0
    # late round 2: Linear-layout weight gradients, the ingest hand-shake and the riding pull
    assert L.tsgnn_wgrad_blocks_oi_f32(None, 8, None, 8, 16, 8, 8, 1, 32, None, None, 8, None, None) == -1
    assert L.tsgnn_ingest_pull_expand_ack_f32(None, None, 4, 64, 256, 1024, 16, 64, None, None, None, None, None, None, 8, None, 8, None, None) == -1
    assert L.tsgnn_ingest_pull_f32(None, None, 4, 64, 256, 1024, 64, None) == -1
    assert L.tsgnn_ingest_arm_pull_rider(None, None, 4, 64, 256, 1024, 64) == -1       # nothing armed by a rejected call:
    assert L.tsgnn_ingest_arm_expand_rider(None, 4, 64, 256, 1024, 16, 64, None, None, None, None, None, None, 8, None, 8, None) == -1
    assert L.tsgnn_ingest_flush_pull_rider(None) == 0                                   # ... so this launches nothing
    assert L.tsgnn_collate_pool_submit_ack(None, None, None, None, None, None, None, 4, 64, 256, 1024, 16, 64, None, None, None, 0, 1,
                                           None) == -1
    # late round 3: entry points whose argument checks run before anything is launched (no GPU here)
    assert L.tsgnn_sddmm_rows_f32(None, None, None, 8, None, 8, 16, 8, None, None, None) == -1
    assert L.tsgnn_triplet_embed_fwd_f32(None, 8, None, 8, None, 8, 4, 1e-6, None, None, None) == -1
    assert L.tsgnn_triplet_embed_bwd_f32(None, 8, None, 8, 8, 4, 1e-6, None, None, None, None, None, None, None, None, 8, None, 8, None, None) == -1
    assert L.tsgnn_margin_rank_fwd_f32(None, None, None, 1, 1.0, 1, None, None, None) == -1
    assert L.tsgnn_margin_rank_bwd_f32(None, None, 1, None, None, None) == -1
    assert L.tsgnn_row_post_bwd_f32(None, 3, 10, 12, None, 8, None, 8, None, 8, None, 8, 1, 1, None, None, None, None, 8, None) == -1
    assert L.tsgnn_gat_bwd_products_f32(None, 256, None, 264, 100, 256, 264, None, 264, None, 256, 4, 32, None, None) == -1
    assert L.tsgnn_wgrad_blocks_reduce_f32(None, 4, 256, 264, None, 264, None) == -1
    assert L.tsgnn_wgrad_blocks_reduce2_f32(None, 4, 256, 264, None, 264, None, 4, 92, 264, None, 264, None) == -1
    assert L.tsgnn_wgrad_blocks_slabs_f32(None, 256, None, 264, 100, 256, 264, 4, 32, None, None) == -1
    off = np.zeros(10, dtype=np.int64)
    assert L.tsgnn_ingest_compact_layout(4, 64, 256, 1024, 64, off.ctypes.data) == 0
    assert off[1] - off[0] == 8 and all(int(o) % 4 == 0 for o in off)                   # header: 4 sizes + sequence word + 3 spare


def test_paired_launch_records_merge_in_order(monkeypatch):
    """sage_stack.run_paired (host logic): two launch records are issued in lockstep, same-kind steps share a launch, everything
    else keeps its order inside its own record; an unsupported combination falls back to the two single launches"""
    import torch
    from two_stage_gnn_amd import _native as nat, sage_stack
    issued = []
    monkeypatch.setattr(nat, "run", lambda q: issued.extend(("single", n, a) for n, a in q))
    ok = {"multi": True}

    def fake_multi(tn, gs):
        if not ok["multi"]:
            return False
        issued.append(("multi", len(tn), len(gs)))
        return True
    monkeypatch.setattr(sage_stack, "_multi", fake_multi)
    monkeypatch.setattr(nat, "try_call", lambda name, *a: issued.append(("pair", name)) or True)
    gp = torch.zeros(1)

    def slab(tag):
        return ("linear_wgrad_f32", (tag,) + (0,) * 10 + (None, None))

    def gather(tag):
        return ("gather_rowgemm_f32", (tag,) + (0,) * 19)

    def bn(tag):
        return ("slot_bn_fwd_f32", (gp, gp, 4, 8, 10, 0, tag, 8, 8, 1, None, None, None, 8, None, 0))
    qa = [gather("a0"), bn("a"), ("readout_x", ("a",)), slab("a1"), gather("a2"), slab("a3")]
    qb = [gather("b0"), bn("b"), ("other_y", ("b",)), slab("b1"), gather("b2"), slab("b3"), ("tail", ("b",))]
    sage_stack.run_paired(qa, qb)
    kinds = [t[0] if t[0] != "single" else t[1] for t in issued]
    assert kinds == ["multi", "pair", "readout_x", "other_y", "multi", "multi", "tail"]
    assert issued[0] == ("multi", 0, 2) and issued[4] == ("multi", 2, 2) and issued[5] == ("multi", 2, 0)
    issued.clear()
    ok["multi"] = False
    sage_stack.run_paired([gather("a0"), slab("a1")], [gather("b0"), slab("b1")])
    assert [t[1] for t in issued] == ["gather_rowgemm_f32", "gather_rowgemm_f32", "linear_wgrad_f32", "linear_wgrad_f32"]
    assert [t[2][0] for t in issued] == ["a0", "b0", "a1", "b1"]


def test_deferred_records_instead_of_launching():
    from two_stage_gnn_amd import _native as nat
    ran = []
    with nat.deferred() as q:
        nat.call("definitely_not_an_entry_point", 1, 2)          # recorded, not looked up
        nat.defer(lambda: ran.append("torch-side"))
        with nat.deferred() as inner:
            nat.call("inner", 3)
        assert [n for n, _ in inner] == ["inner"]
        nat.call("after", 4)
    assert [n for n, _ in q] == ["definitely_not_an_entry_point", None, "after"] and ran == []
    assert nat._defer is None
    q[1][1]()
    assert ran == ["torch-side"]
    nat.defer(lambda: ran.append("now"))                          # outside a record: executed at once
    assert ran[-1] == "now"
