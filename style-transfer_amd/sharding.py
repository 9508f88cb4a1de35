"""Multi-GPU partition of the path: images are independent (run_ori.py:155-246 loops over
dataset.txt triples one at a time), so each rank stylises its own triples and the only exchange is one
gather of the output latents at the end."""
import torch


def shard_indices(n_items, rank, world):
    """Static round-robin: rank r takes items r, r + world, ...  Every item goes to exactly one rank."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_items, world))


def gather_latents(local, dist=None, dst=0):
    """local: (n_local, ...) latents of this rank, same n_local on every rank (pad with
    `pad_to_common` first).  Returns on `dst` the (world, n_local, ...) stack, elsewhere None.  One
    collective for the whole job; with dist None (single process) it is a reshape."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return local.unsqueeze(0)
    world, rank = dist.get_world_size(), dist.get_rank()
    bufs = [torch.empty_like(local) for _ in range(world)] if rank == dst else None
    dist.gather(local, bufs, dst=dst)
    return torch.stack(bufs) if rank == dst else None


def pad_to_common(local, n_items, world):
    """Ranks hold ceil or floor of n_items / world items; pad with zeros to the ceil so that the
    gather has one shape.  Returns (padded, n_valid)."""
    per = -(-n_items // world)
    n = local.shape[0]
    if n < per:
        pad = torch.zeros((per - n,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        local = torch.cat([local, pad], 0)
    return local, n


def reassemble(stacked, n_items):
    """Undo the round-robin on `dst`: (world, per, ...) -> (n_items, ...) in dataset order."""
    world, per = stacked.shape[:2]
    out = stacked.transpose(0, 1).reshape((world * per,) + tuple(stacked.shape[2:]))
    return out[:n_items]
