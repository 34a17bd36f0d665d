"""Multi-GPU layout of the hot path: one process per GPU, READS sharded (rank r owns a contiguous range
of read pairs), gaps/flanks replicated, no collective on the data path; the only exchange is the final gather of the
assembled sequences on rank 0 (north_star: "RCCL over xGMI only for the final gather of closed sequences").
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


def encode_contigs(records):
    """[(gap, k, kv, n_nodes, cov_sum, seq)] -> bytes (line-oriented, order preserved)."""
    return "".join("%d\t%d\t%d\t%d\t%d\t%s\n" % r for r in records).encode()


def decode_contigs(blob):
    out = []
    for line in blob.decode().splitlines():
        g, k, kv, n, c, s = line.split("\t")
        out.append((int(g), int(k), int(kv), int(n), int(c), s))
    return out
