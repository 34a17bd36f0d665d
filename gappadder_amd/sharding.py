"""Multi-GPU layout of the hot path: one process per GPU, READS sharded (rank r owns a contiguous range of
read pairs), gaps / flanks replicated, every gap ASSEMBLED ONCE by its owner rank from the recruits of all ranks (the reference
maps each gap to one Pool task, assemble_gaps.py:296-299).  `ExactOwnerExchange` is the one exchange step on the data path — device
pack by owner, ONE all-to-all with exact split sizes that also carries the per-gap counts, device merge (`OwnerExchange`: the older
form with equal-sized padded slots and an all-gather of the counts, kept as an option) — and `gather_bytes` the final gather of
the closed sequences on rank 0 (north_star).  Backend-agnostic: `nccl` (= RCCL) on GPUs, `gloo` in the CPU tests and in the
one-GPU multi-rank mode of bench.py."""
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


def all_gather_slots(dst, src, backend):
    """Fixed-size slots: no host sizes, no host sync with RCCL; gloo goes through host memory (CPU tensors pass straight through)."""
    if backend == "nccl":
        dist.all_gather_into_tensor(dst, src)
        return
    world = dist.get_world_size()
    parts = list(torch.empty_like(dst, device="cpu").chunk(world))
    dist.all_gather(parts, src.cpu())
    dst.copy_(torch.cat(parts))


def all_to_all_slots(dst, src, backend):
    if backend == "nccl":
        dist.all_to_all_single(dst, src)
        return
    r_ = torch.empty_like(src, device="cpu")
    dist.all_to_all_single(r_, src.cpu())
    dst.copy_(r_)


class OwnerExchange:
    """The one exchange step of a multi-GPU run: every rank has built per-gap pools (one array per library) from ITS
    shard of the reads; each gap's rows go to the rank that owns the gap, and the owner concatenates them in (library, source
    rank) order — with contiguous read shards that is the order of a single-process run over all reads, so pools, contigs and
    closed flags are bit-identical to the 1-GPU run.  Buffers are equal-sized slots `[owner][library][slot_cap]` (no host sizes, no
    host sync); the two kernels are injected: bench.py passes gf_pools_pack_for_owners_dev / gf_pools_merge_dev, the CPU test
    (tests/test_distributed_cpu.py) their definitions in numpy — the collectives and the buffer layout are the same code."""

    def __init__(self, world, n_lib, n_gaps, slot_cap, row_bytes, device, backend):
        self.world, self.n_lib, self.n_gaps, self.slot_cap, self.rb, self.backend = world, n_lib, n_gaps, slot_cap, row_bytes, backend
        self.send = torch.empty(world * n_lib * slot_cap * row_bytes, dtype=torch.uint8, device=device)
        self.recv = torch.empty(world * n_lib * slot_cap * row_bytes, dtype=torch.uint8, device=device)
        self.lib_cnt = torch.zeros(n_lib * n_gaps, dtype=torch.int32, device=device)            # [n_lib][n_gaps], this rank's rows
        self.all_cnt = torch.zeros(world * n_lib * n_gaps, dtype=torch.int32, device=device)    # [world][n_lib][n_gaps]

    def run(self, pack, merge):
        """pack(lib, send, slot_cap, lib_cnt_of_that_library) per library; merge(recv, slot_cap, all_cnt)."""
        for l in range(self.n_lib):
            pack(l, self.send, self.slot_cap, self.lib_cnt[l * self.n_gaps:(l + 1) * self.n_gaps])
        all_gather_slots(self.all_cnt, self.lib_cnt, self.backend)
        all_to_all_slots(self.recv, self.send, self.backend)
        merge(self.recv, self.slot_cap, self.all_cnt)


class ExactOwnerExchange:
    """The same exchange with EXACT split sizes ("all-to-all-v of the recruited rows") and ONE collective: the chunk a
    rank sends to peer d is  [per library: u32 counts of every gap (zero where d is not the owner)] [per library: the rows of d's gaps]
    and holds exactly the rows the sizing pass found for (this rank, d, library) — `rows[src][dst][lib]`, the same table on every
    rank —, so `all_to_all_single` runs with split-size lists that are host constants (no host sizes inside the step, no host sync),
    nothing is padded, and the per-gap counts ride in the same collective instead of an all-gather of their own.  A step that produces
    more rows for a slot than the table gives it sets the pack kernel's error bit (Pipeline.fetch raises); fewer rows leave the tail of
    the slot unused (the counts say how many are valid).  Layout tables (device, one entry per slot s = peer * n_lib + lib):
    `slot_base` (byte offset of the slot's rows), `slot_cap` (rows), `cnt_base` (byte offset of its counts) — for the send and for
    the receive buffer; gf_pools_pack_for_owners_v_dev / gf_pools_merge_v_dev take them (the CPU test their numpy definitions)."""

    def __init__(self, world, rank, n_lib, n_gaps, rows, row_bytes, device, backend):
        rows = np.asarray(rows, dtype=np.int64).reshape(world, world, n_lib)
        self.world, self.rank, self.n_lib, self.n_gaps, self.rb, self.backend = world, rank, n_lib, n_gaps, row_bytes, backend
        self.rows = rows
        hdr = n_lib * n_gaps * 4

        def chunk(src, dst):      # bytes of the chunk src -> dst (a multiple of 16)
            return (hdr + int(rows[src, dst].sum()) * row_bytes + 15) // 16 * 16

        def tables(peers_rows, sizes):      # peers_rows[p][l] = rows of slot (p, l); sizes[p] = bytes of peer p's chunk
            base, cap, cnt, at = [], [], [], 0
            for p in range(world):
                o = at + hdr
                for l in range(n_lib):
                    base.append(o)
                    cap.append(int(peers_rows[p][l]))
                    cnt.append(at + l * n_gaps * 4)
                    o += int(peers_rows[p][l]) * row_bytes
                at += sizes[p]
            return (torch.tensor(base, dtype=torch.int64, device=device), torch.tensor(cap, dtype=torch.int32, device=device),
                    torch.tensor(cnt, dtype=torch.int64, device=device))
        self.in_splits = [chunk(rank, d) for d in range(world)]
        self.out_splits = [chunk(s_, rank) for s_ in range(world)]
        self.send = torch.zeros(max(16, sum(self.in_splits)), dtype=torch.uint8, device=device)
        self.recv = torch.zeros(max(16, sum(self.out_splits)), dtype=torch.uint8, device=device)
        self.send_base, self.send_cap, self.send_cnt = tables(rows[rank], self.in_splits)
        self.recv_base, self.recv_cap, self.recv_cnt = tables(rows[:, rank], self.out_splits)
        self.lib_cnt = torch.zeros(n_lib * n_gaps, dtype=torch.int32, device=device)            # [n_lib][n_gaps], this rank's rows (by-product of the pack)
        self.bytes_sent = sum(b for d, b in enumerate(self.in_splits) if d != rank)             # per step, to the other ranks
        self.header_bytes = hdr

    def run(self, pack, merge):
        """pack(lib, send, slot_base, slot_cap, cnt_base, lib_cnt_of_that_library) per library; merge(recv, slot_base, cnt_base)."""
        for l in range(self.n_lib):
            pack(l, self.send, self.send_base, self.send_cap, self.send_cnt, self.lib_cnt[l * self.n_gaps:(l + 1) * self.n_gaps])
        if self.backend == "nccl":
            dist.all_to_all_single(self.recv, self.send, output_split_sizes=self.out_splits, input_split_sizes=self.in_splits)
        else:
            r_ = torch.empty(self.recv.shape, dtype=torch.uint8)
            dist.all_to_all_single(r_, self.send.cpu(), output_split_sizes=self.out_splits, input_split_sizes=self.in_splits)
            self.recv.copy_(r_)
        merge(self.recv, self.recv_base, self.recv_cnt)


def exchange_rows_table(per_dst_local, coll_device, backend):
    """per_dst_local: int64 [n_lib, world] = rows this rank's sizing pass found per (library, owner rank).  All-gathered:
    int64 numpy [src][dst][lib], the same on every rank (a one-off of the sizing pass, outside the step)."""
    world = dist.get_world_size()
    mine = per_dst_local.to(torch.int64).t().contiguous().to(coll_device)          # [dst][lib]
    parts = [torch.zeros_like(mine) for _ in range(world)]
    dist.all_gather(parts, mine)
    return torch.stack(parts).cpu().numpy()


def encode_contigs(records):
    """[(gap, k, kv, n_nodes, cov_sum, seq)] -> bytes (line-oriented, order preserved)."""
    return "".join("%d\t%d\t%d\t%d\t%d\t%s\n" % r for r in records).encode()


def decode_contigs(blob):
    out = []
    for line in blob.decode().splitlines():
        g, k, kv, n, c, s = line.split("\t")
        out.append((int(g), int(k), int(kv), int(n), int(c), s))
    return out
