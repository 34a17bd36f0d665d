"""N > 1 path on CPU: two `gloo` ranks shard the read pairs, screen their shard (oracle stands in for the GPU kernels
here — this test is about the sharding and the final gather), and rank 0 gathers; the union must equal the 1-rank result."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3)
    gaps, flanks = CO.synth_layout(cfg)
    a, b = SH.shard_range(n_pairs, rank, world)
    packed, recs = CO.synth_pairs(cfg, a, b - a)
    hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)
    recs_out = [(int(h["gap"]), 31, 29, int(h["read"]) + 2 * a, 0, "ACGT") for h in hits]   # global read ids
    parts = SH.gather_bytes(SH.encode_contigs(recs_out), dst=0)
    if rank == 0:
        q.put([SH.decode_contigs(p) for p in parts])
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_gather_equal_single_rank():
    sys.path.insert(0, ROOT)
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    n_pairs = 30001
    assert [SH.shard_range(10, r, 3) for r in range(3)] == [(0, 4), (4, 7), (7, 10)]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs:
        p.start()
    parts = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3)
    gaps, flanks = CO.synth_layout(cfg)
    packed, _ = CO.synth_pairs(cfg, 0, n_pairs)
    hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)
    single = sorted((int(h["gap"]), int(h["read"])) for h in hits)
    multi = sorted((r[0], r[3]) for part in parts for r in part)
    assert len(parts) == 2 and multi == single and len(single) > 100
    # rank order is preserved inside the gathered list: shard 0's reads precede shard 1's
    assert max(r[3] for r in parts[0]) < min(r[3] for r in parts[1])


def _exchange_worker(rank, world, port, n_pairs, batch, q):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3)
    gaps, flanks = CO.synth_layout(cfg)
    a, b = SH.shard_range(n_pairs, rank, world)
    packed, _ = CO.synth_pairs(cfg, a, b - a)
    hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)
    pools = [sorted(set(int(h["read"]) for h in hits if int(h["gap"]) == g)) for g in range(len(gaps))]
    rows = np.concatenate([packed[np.array(p, dtype=np.int64)] for p in pools if p] or [np.zeros((0, 38), np.uint8)])
    off = np.cumsum([0] + [len(p) for p in pools]).astype(np.int64)
    merged, moff = SH.exchange_pools(torch.from_numpy(rows), torch.from_numpy(off), batch=batch)
    q.put((rank, merged.numpy().tobytes(), moff.numpy().tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_pool_exchange_gives_every_gap_one_owner_with_the_single_rank_pool():
    """All-to-all-v of the per-gap pools: the owner's merged pool == the pool built from all reads."""
    sys.path.insert(0, ROOT)
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    n_pairs, world, batch = 20001, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_pairs, batch, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, blob, moff = q.get(timeout=300)
        got[r] = (blob, moff)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3)
    gaps, flanks = CO.synth_layout(cfg)
    packed, _ = CO.synth_pairs(cfg, 0, n_pairs)
    hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)
    owner = SH.gap_owner(len(gaps), world, batch).tolist()
    assert sorted(set(owner)) == [0, 1, 2]
    total = 0
    for g in range(len(gaps)):
        ids = sorted(set(int(h["read"]) for h in hits if int(h["gap"]) == g))
        expect = packed[np.array(ids, dtype=np.int64)].tobytes() if ids else b""
        for r in range(world):
            blob, moff = got[r]
            seg = blob[moff[g] * 38:moff[g + 1] * 38]
            assert seg == (expect if r == owner[g] else b""), (g, r)
        total += len(ids)
    assert total > 100
