"""Multi-GPU partitioning helpers: one process per GPU, torch.distributed (RCCL on GPUs).

Only the map-matching path has an exchange step: the train descriptors are split into
contiguous index ranges, every rank scans its shard for the per-query top-2 and the 32 000-byte
lists (2000 queries x 2 x (dist, idx)) are all-gathered and merged by (distance, global index),
which is the order one scan over the whole map would produce.  Independent streams / frames
are simply sharded, with no collective.
"""


def shard_range(n, rank, world):
    """Contiguous [begin, end) of `n` items owned by `rank`: sizes differ by at most one, lower
    ranks get the larger shards."""
    base, rem = divmod(n, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def streams_for_rank(n_streams, rank, world):
    b, e = shard_range(n_streams, rank, world)
    return list(range(b, e))


def sharded_match_knn2(query, train_shard, base_index, group=None, match_fn=None, merge_fn=None, force_collective=False):
    """Global kNN-2 of `query` against a map whose rows [base_index, base_index + len(train_shard))
    live on this rank.  query / train_shard: uint8 tensors [n, 32] on this rank's device.
    Returns (idx [nq, 2], dist [nq, 2]) identical on every rank.
    match_fn / merge_fn default to the HIP entry points (vfe_dev_match_knn2 / vfe_dev_merge_knn2).
    force_collective: run the all-gather and the merge in a one-rank group too (needs an initialised process group);
    a one-GPU machine exercises the exchange that way, the result is that of the plain scan."""
    import torch
    import torch.distributed as dist
    if match_fn is None or merge_fn is None:
        from . import capi
        match_fn = match_fn or capi.match_knn2
        merge_fn = merge_fn or capi.merge_knn2
    idx, d = match_fn(query, train_shard, base_index)
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    if force_collective and not dist.is_initialized():
        raise RuntimeError("force_collective needs an initialised torch.distributed process group")
    if world == 1 and not force_collective:
        return idx, d
    packed = torch.stack([idx, d]).contiguous()           # [2, nq, 2] int32: one message per rank
    device = packed.device
    if dist.get_backend(group) == "gloo":
        # CPU tests and one-GPU rehearsals: gloo gathers host tensors, one list entry per rank
        host = packed.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        gathered = torch.stack(parts).to(device)
    else:
        # RCCL: one collective into one [world, 2, nq, 2] buffer on the GPU
        gathered = torch.empty((world,) + tuple(packed.shape), dtype=packed.dtype, device=device)
        dist.all_gather_into_tensor(gathered, packed, group=group)
    return merge_fn(gathered[:, 0].contiguous(), gathered[:, 1].contiguous())   # [world, nq, 2] each
