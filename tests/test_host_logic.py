"""Host-side plumbing that needs no GPU: the column buffer that replaces torch.cat of the levels' readouts, the deferred-loss
context, the C-ABI's workspace-size helpers (encoders.py:203,388-391; train.py:121-129)."""
import os
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_readout_columns_alias_the_buffer_and_route_gradients():
    """ReadoutColumns: every part is written in place into its column block, join() IS the concatenation (no copy), and the
    gradient of a part is the matching column slice of the buffer's gradient (read in place: same storage)"""
    from two_stage_gnn_amd import message_passing as mp

    class Write(torch.autograd.Function):                  # stands for a readout launch that writes its block in place
        @staticmethod
        def forward(ctx, x, into):
            into.t.copy_(x * 2.0)
            return into.t

        @staticmethod
        def backward(ctx, d):
            Write.seen.append((d.data_ptr(), tuple(d.shape), d.stride()))
            return d * 2.0, None
    Write.seen = []
    cols = mp.ReadoutColumns(3, 12, torch.device("cpu"))
    xs = [torch.randn(3, 4, requires_grad=True) for _ in range(3)]
    parts = [Write.apply(x, cols.take(4)) for x in xs]
    assert all(not p._is_view() for p in parts)            # aliases of the storage, not autograd views of the buffer
    out = cols.join(parts)
    assert out.data_ptr() == cols.buf.data_ptr() and out.shape == (3, 12)
    torch.testing.assert_close(out.detach(), torch.cat([x.detach() * 2.0 for x in xs], dim=1))
    w = torch.arange(36.0).reshape(3, 12)
    (out * w).sum().backward()
    for k, x in enumerate(xs):
        torch.testing.assert_close(x.grad, 2.0 * w[:, 4 * k:4 * k + 4])
    assert sorted(s[2] for s in Write.seen) == [(12, 1)] * 3            # column slices of ONE [3, 12] gradient, not copies
    # blocks that do not fit, or are not 16-byte aligned: take() says so and join() concatenates
    cols2 = mp.ReadoutColumns(2, 8, torch.device("cpu"))
    a = cols2.take(4)
    assert a is not None and cols2.take(8) is None and cols2.take(4) is None
    p0, p1 = torch.ones(2, 4), torch.zeros(2, 4)
    assert torch.equal(cols2.join([p0, p1]), torch.cat([p0, p1], dim=1))
    assert mp.ReadoutColumns(2, 6, torch.device("cpu")).take(3) is None


def test_deferred_loss_context_restores_the_flag():
    from two_stage_gnn_amd import message_passing as mp
    assert mp.CE_DEFER is False
    with mp.deferred_loss():
        assert mp.CE_DEFER is True
        with mp.deferred_loss():
            assert mp.CE_DEFER is True
        assert mp.CE_DEFER is True
    assert mp.CE_DEFER is False
    with pytest.raises(RuntimeError):
        with mp.deferred_loss():
            raise RuntimeError("x")
    assert mp.CE_DEFER is False
    # off the GPU the library's losses are torch's (nothing is deferred, nothing is launched)
    logits, label = torch.randn(5, 3, requires_grad=True), torch.tensor([0, 2, 1, 1, 0])
    with mp.deferred_loss():
        torch.testing.assert_close(mp.cross_entropy(logits, label), torch.nn.functional.cross_entropy(logits, label))
        logp = torch.log_softmax(logits, -1)
        torch.testing.assert_close(mp.nll_loss(logp, label), torch.nn.functional.nll_loss(logp, label))


def test_workspace_size_helpers():
    """host-only entry points of the C ABI: sizes the callers allocate from (include/tsgnn.h)"""
    from two_stage_gnn_amd import _native as nat
    L = nat.lib()
    assert L.tsgnn_readout_max_ws_words(16, 64, 192) == 1                      # <= 64 slots: one launch, no workspace
    assert L.tsgnn_readout_max_ws_words(16, 512, 192) == 16 * 192 + 8          # packed maxima + the graphs' ticket counters
    assert L.tsgnn_readout_max_ws_words(0, 512, 192) == 0 and L.tsgnn_readout_max_ws_words(1 << 20, 4096, 4096) == -1
    assert L.tsgnn_ragged_tn_direct_supported(64, 512) == 1 and L.tsgnn_ragged_tn_direct_supported(129, 512) == 0
    assert L.tsgnn_ragged_tn_direct_supported(64, 5000) == 0
