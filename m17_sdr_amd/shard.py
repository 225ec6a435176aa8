"""Channel sharding across the GPUs of one node.

The path shards by channel with no data-path collective: rank r owns the
contiguous channel range channel_range(r, world, C) and keeps that range's state
on its own GPU for the life of the stream.  The only exchange is the optional
gather of the 64-byte decoded-frame records to one rank (RCCL when the tensors
are on GPUs, gloo in the CPU tests) -- about 1 MB per 16,384 channels per step
against 125.8 MB of IQ, so it is kept off the timed path by default."""
import torch
import torch.distributed as dist


def channel_range(rank, world, n_channels):
    """Contiguous, balanced split: the first (C mod world) ranks get one extra channel."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def gather_records(recs, counts, dst=0, group=None):
    """Gather per-rank records [Cr, cap, 64] (uint8) and counts [Cr] (int32) to `dst`.

    Ranks may own different numbers of channels; shards are padded to the largest.
    Returns (recs [C, cap, 64], counts [C]) on dst, (None, None) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([recs.shape[0]], dtype=torch.int64, device=recs.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cmax = max(sizes)
    pad_r = torch.zeros((cmax,) + tuple(recs.shape[1:]), dtype=recs.dtype, device=recs.device)
    pad_c = torch.zeros((cmax,), dtype=counts.dtype, device=counts.device)
    pad_r[:recs.shape[0]] = recs
    pad_c[:counts.shape[0]] = counts
    out_r = [torch.empty_like(pad_r) for _ in range(world)] if rank == dst else None
    out_c = [torch.empty_like(pad_c) for _ in range(world)] if rank == dst else None
    dist.gather(pad_r, out_r, dst=dst, group=group)
    dist.gather(pad_c, out_c, dst=dst, group=group)
    if rank != dst:
        return None, None
    return (torch.cat([out_r[r][:sizes[r]] for r in range(world)]),
            torch.cat([out_c[r][:sizes[r]] for r in range(world)]))


def scatter_iq(iq_full, n_channels, src=0, group=None, device=None):
    """Fan the [C, nblk, 1920, 2] int16 IQ of `src` out to the owning ranks
    (point-to-point sends, one per peer: xGMI is a full mesh, no ring needed)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = channel_range(rank, world, n_channels)
    if rank == src:
        reqs = []
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r != src and b > a:
                reqs.append(dist.isend(iq_full[a:b].contiguous(), dst=r, group=group))
        mine = iq_full[lo:hi].contiguous()
        for q in reqs:
            q.wait()
        return mine
    shape = (hi - lo,) + tuple(iq_full.shape[1:]) if iq_full is not None else None
    raise_if = shape is None
    if raise_if:
        raise ValueError("non-source ranks pass a template tensor of the full shape (any device)")
    mine = torch.empty(shape, dtype=torch.int16, device=device or iq_full.device)
    if hi > lo:
        dist.recv(mine, src=src, group=group)
    return mine
