from two_stage_gnn_amd.pyg import (GATConv, GCNConv, GraphConv, SAGEConv, SAGPooling, TopKPooling, dense_diff_pool,  # noqa: F401
                                   global_max_pool, global_mean_pool)
