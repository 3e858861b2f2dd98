"""Batch sharding across ranks and the packed record that is all-gathered.

Instances are independent, so GPU g of G owns the contiguous block [g*B/G, (g+1)*B/G); the only exchange is one
all-gather of the solved trajectories + torques per batch.  Used by bench.py (RCCL) and by the gloo CPU tests.
"""
import numpy as np


def shard_bounds(total, world, rank):
    """Contiguous block of rank `rank`; the first (total % world) ranks get one extra instance."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def pack_len(num_nodes):
    return (num_nodes + 1) * 30 + num_nodes * 30 + 54 + (num_nodes + 1)


def pack(X, U, wbc_out, modes):
    """[B, (N+1)*30 + N*30 + 54 + (N+1)] float64: X | U | [x(36) tau(18)] | mode (exact in fp64)."""
    xp = np if isinstance(X, np.ndarray) else None
    B = X.shape[0]
    if xp is not None:
        return np.concatenate([X.reshape(B, -1), U.reshape(B, -1), wbc_out.reshape(B, -1), modes.astype(np.float64).reshape(B, -1)], axis=1)
    import torch
    return torch.cat([X.reshape(B, -1), U.reshape(B, -1), wbc_out.reshape(B, -1), modes.to(torch.float64).reshape(B, -1)], dim=1)


def unpack(packed, num_nodes):
    N = num_nodes
    B = packed.shape[0]
    a, b, c = (N + 1) * 30, (N + 1) * 30 + N * 30, (N + 1) * 30 + N * 30 + 54
    X = packed[:, :a].reshape(B, N + 1, 30)
    U = packed[:, a:b].reshape(B, N, 30)
    wbc = packed[:, b:c]
    modes = packed[:, c:]
    return X, U, wbc, modes
