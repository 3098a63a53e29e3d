from two_stage_gnn_amd.pyg import filter_adj, topk  # noqa: F401
