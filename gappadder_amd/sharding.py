"""Multi-GPU layout of the hot path: one process per GPU, READS sharded (rank r owns a contiguous range
of read pairs), gaps/flanks replicated; by default no collective on the data path and the only exchange is the final gather
of the assembled sequences on rank 0 (north_star: "RCCL over xGMI only for the final gather of closed sequences").
`exchange_pools` is the optional all-to-all-v that gives every gap ONE owner holding the recruits of all ranks.
Backend-agnostic: `nccl` (= RCCL) on GPUs, `gloo` in the CPU tests."""
import numpy as np
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous, balanced split: ranks [0, n % world) get one extra item."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_bytes(payload, dst=0, device=None):
    """Gather one variable-length byte string per rank on `dst` (sizes all-gathered first, payloads padded to the max).
    Returns the list of payloads on dst, None elsewhere."""
    world, rank = dist.get_world_size(), dist.get_rank()
    device = device or torch.device("cpu")
    n = torch.tensor([len(payload)], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    mx = max(max(sizes), 1)
    buf = torch.zeros(mx, dtype=torch.uint8, device=device)
    if payload:
        buf[:len(payload)] = torch.from_numpy(np.frombuffer(payload, dtype=np.uint8).copy()).to(device)
    parts = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(world)] if rank == dst else None
    dist.gather(buf, parts, dst=dst)
    if rank != dst:
        return None
    return [bytes(p[:s].cpu().numpy().tobytes()) for p, s in zip(parts, sizes)]


def owner_batch(n_gaps, world, batch=256):
    """Batch size of the round-robin deal: 256 consecutive gaps, fewer when that would leave ranks without gaps."""
    return max(1, min(batch, -(-n_gaps // max(1, world))))


def gap_owner(n_gaps, world, batch=256):
    """Owner rank of every gap: batches of consecutive gaps dealt round-robin."""
    return (torch.arange(n_gaps, dtype=torch.int64) // owner_batch(n_gaps, world, batch)) % world


def exchange_pools(pool, pool_off, coll_device=None, batch=256):
    """The one exchange step of a multi-GPU run that wants whole-data pools per gap: every rank has built
    per-gap pools from ITS shard of the reads; each gap's pool is sent to the rank that owns the gap (all-to-all-v) and the
    owner concatenates the contributions in source-rank order — with contiguous read shards that is global read order, so
    the merged pool of a gap is byte-identical to the pool a single process builds from all reads.

    pool:     uint8 tensor [n_rows, row_bytes] — packed reads of all gaps, gap after gap
    pool_off: int64/uint64 tensor [n_gaps + 1] — row offsets per gap
    coll_device: device the collectives run on (the pool's device with nccl/RCCL, cpu with gloo)
    Returns (merged_pool [m_rows, row_bytes] on pool.device, merged_off int64 [n_gaps + 1] on pool.device);
    gaps this rank does not own come back empty."""
    world, rank = dist.get_world_size(), dist.get_rank()
    dev = pool.device
    cdev = coll_device or dev
    n_gaps = pool_off.numel() - 1
    rb = pool.shape[1] if pool.dim() == 2 else 1
    off = pool_off.to(torch.int64)
    counts = (off[1:] - off[:-1]).to(cdev)
    all_counts = [torch.zeros_like(counts) for _ in range(world)]
    dist.all_gather(all_counts, counts)
    all_counts = torch.stack(all_counts)                      # [world, n_gaps]
    owner = gap_owner(n_gaps, world, batch).to(cdev)
    # send side: rows regrouped by destination (stable: gap order, then read order, survives inside each group)
    gid = torch.repeat_interleave(torch.arange(n_gaps, device=cdev), counts)
    perm = torch.argsort(owner[gid], stable=True)
    send = pool.reshape(-1, rb)[perm.to(dev)].to(cdev).reshape(-1)
    send_rows = torch.zeros(world, dtype=torch.int64, device=cdev).index_add_(0, owner, counts)
    mine = owner == rank
    recv_rows = (all_counts * mine).sum(dim=1)                # rows coming from every source
    recv = torch.empty(int(recv_rows.sum()) * rb, dtype=torch.uint8, device=cdev)
    dist.all_to_all_single(recv, send, [int(x) * rb for x in recv_rows], [int(x) * rb for x in send_rows])
    # receive side: rows arrive as [source 0: my gaps in order | source 1: ...]; regroup by gap, sources in rank order
    my_gaps = torch.nonzero(mine).reshape(-1)
    gid_r = torch.cat([torch.repeat_interleave(my_gaps, all_counts[s][my_gaps]) for s in range(world)]) if world else gid
    perm2 = torch.argsort(gid_r, stable=True)                 # stable: keeps source order inside a gap
    merged = recv.reshape(-1, rb)[perm2].to(dev)
    mcounts = (all_counts.sum(dim=0) * mine).to(torch.int64)
    moff = torch.zeros(n_gaps + 1, dtype=torch.int64, device=cdev)
    moff[1:] = torch.cumsum(mcounts, 0)
    return merged, moff.to(dev)


def encode_contigs(records):
    """[(gap, k, kv, n_nodes, cov_sum, seq)] -> bytes (line-oriented, order preserved)."""
    return "".join("%d\t%d\t%d\t%d\t%d\t%s\n" % r for r in records).encode()


def decode_contigs(blob):
    out = []
    for line in blob.decode().splitlines():
        g, k, kv, n, c, s = line.split("\t")
        out.append((int(g), int(k), int(kv), int(n), int(c), s))
    return out
