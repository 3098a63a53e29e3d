"""Minimal ``torch_geometric`` namespace backed by two_stage_gnn_amd — ONLY the names Code/sag imports
(network.py:2-4, layers.py:1-2).  Put ``two-stage-gnn_amd/shim`` on sys.path when the real package is not
installed and the reference's Code/sag/network.py + layers.py import and run unchanged."""
__version__ = "1.6.3+tsgnn"
