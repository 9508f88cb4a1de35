"""Multi-GPU partition of the path: images are independent (run_ori.py:155-246 loops over
dataset.txt triples one at a time), so each rank stylises its own triples and the only exchange is one
gather of the output latents at the end."""
import torch


def shard_indices(n_items, rank, world):
    """Static round-robin: rank r takes items r, r + world, ...  Every item goes to exactly one rank."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world of %d" % (rank, world))
    return list(range(rank, n_items, world))


def shard_by_content(content_keys, world):
    """Partition for runs that reuse what depends on the content image alone (its feature list, cached
    content blocks, filtered residuals): all items with one content key go to ONE rank, in dataset order.
    Groups are placed largest first on the rank with the fewest items so far (ties: the lower rank), so the
    assignment is the same on every rank without any exchange.  Returns one index list per rank, the items
    of a content key next to each other (a rank that walks its list meets every content image once)."""
    groups = {}
    for i, key in enumerate(content_keys):
        groups.setdefault(key, []).append(i)
    order = sorted(groups.values(), key=lambda g: (-len(g), g[0]))
    lists = [[] for _ in range(world)]
    for g in order:
        r = min(range(world), key=lambda r: (len(lists[r]), r))
        lists[r].extend(g)
    return lists


def reassemble_lists(stacked, lists):
    """Undo an arbitrary static partition on `dst`: row j of rank r's block is item lists[r][j]."""
    n = sum(len(l) for l in lists)
    out = torch.empty((n,) + tuple(stacked.shape[2:]), dtype=stacked.dtype, device=stacked.device)
    for r, l in enumerate(lists):
        if l:
            out[torch.as_tensor(l, device=stacked.device)] = stacked[r, :len(l)]
    return out


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


def pad_to_common(local, n_items, world, per=None):
    """Ranks hold ceil or floor of n_items / world items; pad with zeros to the ceil (or to `per`, the
    longest list of a partition that is not round-robin) so that the gather has one shape.
    Returns (padded, n_valid)."""
    per = -(-n_items // world) if per is None else per
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
