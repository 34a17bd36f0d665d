"""Multi-GPU layout of the hot path: one process per GPU, READS sharded (rank r owns a contiguous range of
read pairs), gaps / flanks replicated, every gap ASSEMBLED ONCE by its owner rank from the recruits of all ranks (the reference
maps each gap to one Pool task, assemble_gaps.py:296-299).  `OwnerExchange` is the one exchange step on the data path — device
pack by owner, all-gather of the per-gap counts, equal-slot all-to-all, device merge — and `gather_bytes` the final gather of
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


def encode_contigs(records):
    """[(gap, k, kv, n_nodes, cov_sum, seq)] -> bytes (line-oriented, order preserved)."""
    return "".join("%d\t%d\t%d\t%d\t%d\t%s\n" % r for r in records).encode()


def decode_contigs(blob):
    out = []
    for line in blob.decode().splitlines():
        g, k, kv, n, c, s = line.split("\t")
        out.append((int(g), int(k), int(kv), int(n), int(c), s))
    return out
