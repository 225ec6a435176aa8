"""Channel sharding across the GPUs of one node.

The path shards by channel with no data-path collective: rank r owns the contiguous channel range
channel_range(r, world, C) and keeps that range's state on its own GPU for the life of the stream.
The only exchanges are at the edges of a step, both point-to-point from / to one ingest rank:

    scatter_iq      the IQ of all channels fans out from the ingest rank, one send per peer (xGMI is a
                    full mesh: 7 links x ~153 GB/s, so 7 direct sends run on 7 links; a ring would be
                    per-link bound for no benefit)
    gather_packed   the 64-byte decoded-frame records come back, valid rows only (Receiver.pack_records /
                    m17gpu_pack_records first): about 1 MB per 16,384 channels x 12 blocks against
                    125.8 MB of IQ per block
    gather_records  the same for the unpacked [C, cap, 64] array (27 MB at that size): kept for callers that
                    hold no packed form

With the "nccl" backend (= RCCL on ROCm) device tensors move GPU to GPU.  Backends that cannot move
device tensors (gloo: CPU tests, and the 1-GPU rehearsal of bench.py) stage through host memory; the
call sites are the same."""
import torch
import torch.distributed as dist


def channel_range(rank, world, n_channels):
    """Contiguous, balanced split: the first (C mod world) ranks get one extra channel."""
    base, extra = divmod(n_channels, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def _moves_device_tensors(group=None):
    return dist.get_backend(group) == "nccl"


def gather_records(recs, counts, dst=0, group=None):
    """Gather per-rank records [Cr, cap, 64] (uint8) and counts [Cr] (int32) to `dst`.

    Ranks may own different numbers of channels; shards are padded to the largest.
    Returns (recs [C, cap, 64], counts [C]) on dst (on the device the inputs live on), (None, None) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = recs.device
    wire = home if (_moves_device_tensors(group) or home.type == "cpu") else torch.device("cpu")
    n = torch.tensor([recs.shape[0]], dtype=torch.int64, device=wire)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cmax = max(sizes)
    pad_r = torch.zeros((cmax,) + tuple(recs.shape[1:]), dtype=recs.dtype, device=wire)
    pad_c = torch.zeros((cmax,), dtype=counts.dtype, device=wire)
    pad_r[:recs.shape[0]] = recs
    pad_c[:counts.shape[0]] = counts
    out_r = [torch.empty_like(pad_r) for _ in range(world)] if rank == dst else None
    out_c = [torch.empty_like(pad_c) for _ in range(world)] if rank == dst else None
    dist.gather(pad_r, out_r, dst=dst, group=group)
    dist.gather(pad_c, out_c, dst=dst, group=group)
    if rank != dst:
        return None, None
    return (torch.cat([out_r[r][:sizes[r]] for r in range(world)]).to(home),
            torch.cat([out_c[r][:sizes[r]] for r in range(world)]).to(home))


def gather_packed(packed, offsets, dst=0, group=None):
    """Gather the PACKED records of a step to `dst`: packed [rows, 64] uint8 whose first offsets[-1] rows are valid,
    offsets [Cr + 1] int32 (exclusive scan of the rank's per-channel counts).  Two exchanges: the offset tables
    (4 bytes per channel) and sum(counts) x 64 bytes per rank -- nothing of the unused record capacity moves.  The row
    counts size the second exchange, so every rank reads its own total back (one host sync per step, behind the step).
    A rank whose `packed` is smaller than its offsets[-1] rows makes the call raise ValueError on EVERY rank before
    anything moves.
    Returns (packed_all [sum, 64], offsets_all [C + 1] GLOBAL offsets, totals list) on dst, (None, None, None) elsewhere."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    home = packed.device
    direct = _moves_device_tensors(group) or home.type == "cpu"
    wire = home if direct else torch.device("cpu")
    n_mine = int(offsets[-1].item())
    # Receiver.pack_records / m17gpu_pack_records write no row beyond the capacity of `packed` while `offsets` still
    # counts them all: a rank whose step outgrew its buffer says so in the exchange every rank takes part in, and then
    # EVERY rank raises -- none is left in a send or a receive that has no partner
    meta = torch.tensor([offsets.shape[0] - 1, n_mine, 1 if 0 <= n_mine <= packed.shape[0] else 0], dtype=torch.int64, device=wire)
    metas = [torch.zeros_like(meta) for _ in range(world)]
    dist.all_gather(metas, meta, group=group)
    chans = [int(t[0].item()) for t in metas]
    totals = [int(t[1].item()) for t in metas]
    short = [r for r, t in enumerate(metas) if int(t[2].item()) == 0]
    if short:
        raise ValueError(f"gather_packed: rank(s) {short} packed more records than their buffer holds ({totals}); "
                         "refused on every rank, nothing moved")
    ops, keep = [], []
    if rank == dst:
        offs_all = torch.zeros((sum(chans) + 1,), dtype=torch.int32, device=wire)
        rows_all = torch.empty((sum(totals), 64), dtype=torch.uint8, device=wire)
        c0 = r0 = 0
        for r in range(world):
            o_r = offs_all[c0 + 1:c0 + 1 + chans[r]]
            p_r = rows_all[r0:r0 + totals[r]]
            if r == rank:
                o_r.copy_(offsets[1:].to(wire))
                p_r.copy_(packed[:n_mine].to(wire))
            else:
                if chans[r]:
                    ops.append(dist.P2POp(dist.irecv, o_r, r, group))
                if totals[r]:
                    ops.append(dist.P2POp(dist.irecv, p_r, r, group))
            c0 += chans[r]
            r0 += totals[r]
    else:
        if chans[rank]:
            keep.append(offsets[1:].to(wire).contiguous())
            ops.append(dist.P2POp(dist.isend, keep[-1], dst, group))
        if n_mine:
            keep.append(packed[:n_mine].to(wire).contiguous())
            ops.append(dist.P2POp(dist.isend, keep[-1], dst, group))
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    if rank != dst:
        return None, None, None
    # local offsets -> global: every rank's table moves behind the rows of the ranks before it
    c0 = r0 = 0
    for r in range(world):
        offs_all[c0 + 1:c0 + 1 + chans[r]] += r0
        c0 += chans[r]
        r0 += totals[r]
    return rows_all.to(home), offs_all.to(home), totals


def scatter_iq(iq_full, n_channels, nblk, src=0, group=None, device=None):
    """Fan the [C, nblk, 1920, 2] int16 IQ held by `src` out to the owning ranks point to point, as ONE grouped
    operation (batch_isend_irecv = ncclGroupStart / ncclSend x (world-1) / ncclGroupEnd on RCCL): the sends to
    the seven peers are in flight together, each on its own xGMI link.  Non-source ranks pass iq_full=None.
    Returns this rank's shard [hi-lo, nblk, 1920, 2] on `device` (default: the source tensor's device / the
    current device)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    lo, hi = channel_range(rank, world, n_channels)
    direct = _moves_device_tensors(group)
    ops, keep = [], []
    if rank == src:
        if iq_full is None or tuple(iq_full.shape) != (n_channels, nblk, 1920, 2) or iq_full.dtype != torch.int16:
            raise ValueError(f"the source rank passes the full IQ tensor [{n_channels}, {nblk}, 1920, 2] int16")
        home = device if device is not None else iq_full.device
        for r in range(world):
            a, b = channel_range(r, world, n_channels)
            if r != src and b > a:
                part = iq_full[a:b]                                  # contiguous: a range of whole channels
                if not direct and part.device.type != "cpu":
                    part = part.cpu()
                keep.append(part)
                ops.append(dist.P2POp(dist.isend, part, r, group))
        mine = iq_full[lo:hi].to(home)
    else:
        if iq_full is not None:
            raise ValueError("only the source rank passes the full IQ tensor")
        home = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        wire = home if direct else torch.device("cpu")
        mine = torch.empty((hi - lo, nblk, 1920, 2), dtype=torch.int16, device=wire)
        if hi > lo:
            ops.append(dist.P2POp(dist.irecv, mine, src, group))
    for q in (dist.batch_isend_irecv(ops) if ops else []):
        q.wait()
    return mine.to(home)
