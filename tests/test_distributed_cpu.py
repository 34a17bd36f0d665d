"""N > 1 path on CPU (gloo): read sharding + final gather on two ranks, and the owner exchange that bench.py's step() runs
(sharding.OwnerExchange) on three ranks; the oracle / numpy definitions stand in for the GPU kernels."""
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


def _np_pack_for_owners(rows, off, world, batch, lib, n_lib, send, slot_cap, cnt, rb):
    """Definition of gf_pools_pack_for_owners_dev (include/gapfill_hip.h): slot (owner * n_lib + lib) of the send buffer holds that
    owner's gaps' rows in gap order; cnt[g] = this library's rows of gap g."""
    n_gaps = len(off) - 1
    fill = [0] * world
    sv = send.numpy().reshape(world, n_lib, slot_cap, rb)
    for g in range(n_gaps):
        o = (g // batch) % world
        n = int(off[g + 1] - off[g])
        sv[o, lib, fill[o]:fill[o] + n] = rows[int(off[g]):int(off[g + 1])]
        fill[o] += n
        cnt[g] = n


def _np_merge(recv, slot_cap, all_cnt, n_lib, world, n_gaps, rank, batch, rb):
    """Definition of gf_pools_merge_dev: for each of MY gaps, libraries in order, inside a library the source ranks in order."""
    rv = recv.numpy().reshape(world, n_lib, slot_cap, rb)
    cnt = all_cnt.numpy().reshape(world, n_lib, n_gaps)
    pos = np.zeros((world, n_lib), dtype=np.int64)
    out, moff = [], [0]
    for g in range(n_gaps):
        n = 0
        if (g // batch) % world == rank:
            for l in range(n_lib):
                for r in range(world):
                    c = int(cnt[r, l, g])
                    out.append(rv[r, l, pos[r, l]:pos[r, l] + c].copy())
                    pos[r, l] += c
                    n += c
        moff.append(moff[-1] + n)
    return (np.concatenate(out) if out else np.zeros((0, rb), np.uint8)), moff


def _np_pack_for_owners_v(rows, off, world, batch, lib, n_lib, send, slot_base, slot_cap, cnt_base, cnt, rb):
    """Definition of gf_pools_pack_for_owners_v_dev: slot s = owner * n_lib + lib starts at BYTE slot_base[s] of the send buffer and
    holds slot_cap[s] rows; its per-gap counts (zero where the peer is not the owner) go to u32[n_gaps] at byte cnt_base[s]."""
    n_gaps = len(off) - 1
    sv = send.numpy()
    fill = [0] * world
    for d in range(world):
        hc = sv[int(cnt_base[d * n_lib + lib]):int(cnt_base[d * n_lib + lib]) + 4 * n_gaps].view(np.uint32)
        hc[:] = 0
    for g in range(n_gaps):
        o = (g // batch) % world
        n = int(off[g + 1] - off[g])
        s_ = o * n_lib + lib
        assert fill[o] + n <= int(slot_cap[s_])
        at = int(slot_base[s_]) + fill[o] * rb
        sv[at:at + n * rb] = rows[int(off[g]):int(off[g + 1])].reshape(-1)
        fill[o] += n
        cnt[g] = n
        sv[int(cnt_base[s_]):int(cnt_base[s_]) + 4 * n_gaps].view(np.uint32)[g] = n


def _np_merge_v(recv, slot_base, cnt_base, n_lib, world, n_gaps, rank, batch, rb):
    """Definition of gf_pools_merge_v_dev: the counts are read from the received buffer itself."""
    rv = recv.numpy()
    cnt = np.stack([rv[int(cnt_base[s_]):int(cnt_base[s_]) + 4 * n_gaps].view(np.uint32) for s_ in range(world * n_lib)]).reshape(world, n_lib, n_gaps)
    pos = np.zeros((world, n_lib), dtype=np.int64)
    out, moff = [], [0]
    for g in range(n_gaps):
        n = 0
        if (g // batch) % world == rank:
            for l in range(n_lib):
                for r in range(world):
                    c = int(cnt[r, l, g])
                    at = int(slot_base[r * n_lib + l]) + int(pos[r, l]) * rb
                    out.append(rv[at:at + c * rb].reshape(c, rb).copy())
                    pos[r, l] += c
                    n += c
        moff.append(moff[-1] + n)
    return (np.concatenate(out) if out else np.zeros((0, rb), np.uint8)), moff


def _exchange_worker(rank, world, port, n_pairs, batch, q, exact=False):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n_lib, rb = 2, 38
    gaps = flanks = None
    libs = []
    for l in range(n_lib):                                   # two libraries drawn from the same draft (bench.py's C5 shape)
        cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3, library=l, insert_mean=300 if l == 0 else 900, insert_sd=30)
        gaps, flanks = CO.synth_layout(cfg)
        a, b = SH.shard_range(n_pairs, rank, world)
        packed, _ = CO.synth_pairs(cfg, a, b - a)
        hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)    # the oracle stands in for the recruit kernels
        pools = [sorted(set(int(h["read"]) for h in hits if int(h["gap"]) == g)) for g in range(len(gaps))]
        rows = np.concatenate([packed[np.array(p, dtype=np.int64)] for p in pools if p] or [np.zeros((0, rb), np.uint8)])
        libs.append((rows, np.cumsum([0] + [len(p) for p in pools]).astype(np.int64)))
    n_gaps = len(gaps)
    assert batch == SH.owner_batch(n_gaps, world, batch)
    res = {}
    if exact:
        # the sizing pass of Pipeline.prepare(): rows per (library, owner) on this rank, all-gathered into the [src][dst][lib] table
        owner = SH.gap_owner(n_gaps, world, batch)
        per_dst = torch.zeros(n_lib, world, dtype=torch.int64)
        for l in range(n_lib):
            per_dst[l].index_add_(0, owner, torch.from_numpy(np.diff(libs[l][1])))
        table = SH.exchange_rows_table(per_dst, torch.device("cpu"), "gloo")
        x = SH.ExactOwnerExchange(world, rank, n_lib, n_gaps, table, rb, torch.device("cpu"), "gloo")     # the class Pipeline.step() drives
        assert sum(x.in_splits) == len(x.send) and all(b % 16 == 0 for b in x.in_splits + x.out_splits)
        assert int(table[rank].sum()) * rb + world * x.header_bytes <= sum(x.in_splits) < int(table[rank].sum()) * rb + world * (x.header_bytes + 16)   # nothing padded

        def pack(l, send, slot_base, slot_cap, cnt_base, cnt):
            _np_pack_for_owners_v(libs[l][0], libs[l][1], world, batch, l, n_lib, send, slot_base.numpy(), slot_cap.numpy(), cnt_base.numpy(), cnt.numpy(), rb)

        def merge(recv, slot_base, cnt_base):
            res["m"] = _np_merge_v(recv, slot_base.numpy(), cnt_base.numpy(), n_lib, world, n_gaps, rank, batch, rb)
    else:
        x = SH.OwnerExchange(world, n_lib, n_gaps, 512, rb, torch.device("cpu"), "gloo")     # the equal-slot form (GF_XCHG=slots)

        def pack(l, send, cap, cnt):
            _np_pack_for_owners(libs[l][0], libs[l][1], world, batch, l, n_lib, send, cap, cnt.numpy(), rb)

        def merge(recv, cap, all_cnt):
            res["m"] = _np_merge(recv, cap, all_cnt, n_lib, world, n_gaps, rank, batch, rb)
    x.run(pack, merge)
    x.run(pack, merge)                                       # a second step reuses the buffers
    merged, moff = res["m"]
    q.put((rank, merged.tobytes(), moff))
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("exact", [True, False])
def test_owner_exchange_gives_every_gap_one_owner_with_the_single_rank_pool(exact):
    """sharding.ExactOwnerExchange — the exchange Pipeline.step() runs at N > 1: pack by owner, ONE all-to-all with exact split sizes
    (all-to-all-v) that also carries the per-gap counts, merge — and sharding.OwnerExchange, the older equal-slot form
    with an all-gather of the counts (GF_XCHG=slots), on three gloo ranks with the kernels' definitions in numpy: every gap's merged
    pool sits at exactly one rank (assemble_gaps.py:296-299: one owner per gap) and equals the pool a single process builds from all
    reads of both libraries, libraries in order."""
    sys.path.insert(0, ROOT)
    from gappadder_amd import sharding as SH
    from oracle import c_oracle as CO
    n_pairs, world, batch = 20001, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_exchange_worker, args=(r, world, port, n_pairs, batch, q, exact)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict()
    for _ in range(world):
        r, blob, moff = q.get(timeout=300)
        got[r] = (blob, moff)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    expect = None
    for l in range(2):
        cfg = CO.synth_cfg(scaffold_len=200000, n_scaffolds=4, gaps_per_scaffold=3, library=l, insert_mean=300 if l == 0 else 900, insert_sd=30)
        gaps, flanks = CO.synth_layout(cfg)
        packed, _ = CO.synth_pairs(cfg, 0, n_pairs)
        hits = CO.screen_reads(CO.unpack_reads(packed, 150), 150, flanks, 31, threads=2)
        if expect is None:
            expect = [b""] * len(gaps)
        for g in range(len(gaps)):
            ids = sorted(set(int(h["read"]) for h in hits if int(h["gap"]) == g))
            expect[g] += packed[np.array(ids, dtype=np.int64)].tobytes() if ids else b""
    owner = SH.gap_owner(len(gaps), world, batch).tolist()
    assert sorted(set(owner)) == [0, 1, 2]
    for g in range(len(gaps)):
        for r in range(world):
            blob, moff = got[r]
            seg = blob[moff[g] * 38:moff[g + 1] * 38]
            assert seg == (expect[g] if r == owner[g] else b""), (g, r)
    assert sum(len(e) for e in expect) > 200 * 38
