from two_stage_gnn_amd.pyg import (GATConv, GCNConv, GraphConv, SAGEConv, SAGPooling, dense_diff_pool,  # noqa: F401
                                   global_max_pool, global_mean_pool)


class TopKPooling:  # imported by Code/sag/network.py:3 but never used there
    def __init__(self, *a, **k):
        raise NotImplementedError("TopKPooling is imported but unused by the reference; not provided")
