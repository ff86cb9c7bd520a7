"""Multi-GPU plumbing for the SSG loss: one process per GPU, images sharded by rank.

The reference computes the loss per rank on its LOCAL images and normalises by
the LOCAL edge-pixel count; DDP then averages parameter gradients
(base_model.py:95-98, data_sampler.py:6-48).  The SSG path therefore needs no
collective on its data path.  The only exchange is optional and tiny: three
numbers per rank to report a global-mean loss (the logging reduce of
base_model.py:367-392 does the same for its scalars).  On MI355X nodes the
"nccl" backend is RCCL over xGMI; the CPU tests use gloo.
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world_size):
    """Contiguous [lo, hi) slice of n_items owned by `rank` (sizes differ by at most 1)."""
    base, rem = divmod(n_items, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def global_mean_losses(l1, kl, n_edges, ks, group=None):
    """All-reduce [l1*M, kl*M, M] (M = n_edges*ks^2) -> (l1, kl) as if the whole job were one batch.

    l1, kl: 0-dim tensors (local means, already weighted); n_edges: int or 0-dim tensor.
    Returns local values unchanged when torch.distributed is not initialised.
    """
    if not (dist.is_available() and dist.is_initialized()):
        return l1, kl, torch.as_tensor(n_edges)
    m = torch.as_tensor(n_edges, dtype=torch.float64, device=l1.device) * float(ks * ks)
    v = torch.stack([l1.double() * m, kl.double() * m, m])
    dist.all_reduce(v, op=dist.ReduceOp.SUM, group=group)
    tot = torch.clamp(v[2], min=1.0)
    return (v[0] / tot).to(l1.dtype), (v[1] / tot).to(kl.dtype), (v[2] / float(ks * ks)).round().long()
