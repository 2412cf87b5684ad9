"""dense_to_sparse stand-in: row-major nonzero of a 2-D adjacency (PyG semantics)."""
import torch


def dense_to_sparse(adj):
    assert adj.dim() == 2
    index = adj.nonzero().t().contiguous()
    return index, adj[index[0], index[1]]
