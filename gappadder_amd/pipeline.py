"""The device-resident hot path as ONE object: recruit -> per-gap pools -> per-gap assembly -> flank anchoring.

What the reference does with files between processes —
    run_multi_threads_collect_reads.py:16-38 + collect_reads_for_gaps.py:68-263        (alignment tagger, per scaffold)
    run_multi_threads_discordant.py:19-138 + collect_discordant_low_mapq_reads.py:4-84 (second hop)
    run_multi_threads_discordant.py:141-317                                            (FASTQ join -> gap_reads/{id}.fastq)
    merge_reads.py:12-56                                                               (`cat` over the libraries)
    assemble_gaps.py:82-136, 277-299                                                   (KMC + Velvet per gap and (k, kv))
    pick_contigs.py:64-358 at scores 30 then 15 (assemble_gaps.py:336, 365)            (which gaps are closed)
— runs here as a chain of launches on libraries that stay in HBM (packed reads + 32-byte alignment records, `DeviceLibrary`):

    gf_screen_reads_dev + gf_tag_alignments_low_dev            recruit (k-mer screen: north_star; tagger: a-2)
    gf_second_hop_table_dev + gf_tag_low_mapq_table_dev        second hop (a-3)            [N > 1: union of the ranks' rows]
    gf_pool_keys_all_dev + gf_build_pools_dev                  pools per library (a-4)
    gf_pools_merge_dev                                         libraries in order (a-5)    [N > 1: sharding.OwnerExchange]
    gf_assemble_multi_dev                                      every (k, kv) pair (a-6)
    gf_pick_anchored2_dev                                      closed gaps (f-1)
    gf_merge_open_gaps_dev + gf_pick_anchored2_from_dev        contig merger for the gaps still open, second pick (f-3; merge_in_step)

`Pipeline` owns the sizing pass (capacities follow what the libraries actually recruit), every intermediate buffer, the stream
wiring and — in a multi-rank run — the one exchange step.  Two callers: bench.py (libraries synthesised on the
device; `step()` is what it times) and the CLI (`gappadder_amd/device_collect.py`: libraries ingested from BAM + FASTQ files,
per-gap FASTQ / FASTA files WRITTEN FROM the results).  torch is the device-memory and collective plumbing; every kernel is
behind the C ABI.  There is no CPU path: `_lib.lib()` raises when libgapfill_hip.so is missing."""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as B
from . import sharding as SH

# counters of a library (device u32 words of `d_cnt`): 0 screen hits, 4 tagger hits, 8 second-hop hits, 12 pool keys,
# 24 pool error, 28 MAPQ==0 records, 29 second-hop table rows
CNT_SCREEN, CNT_TAG, CNT_HOP, CNT_KEYS, CNT_POOLERR, CNT_LOW, CNT_ROWS = 0, 4, 8, 12, 24, 28, 29


class DeviceLibrary:
    """One read library (an `alignments[]` + `raw_reads[]` entry of the reference's JSON, main.py:63-95) resident on this rank:
    `n_reads` packed reads (read 2*pair + mate; gf_pack_reads layout) and `n_recs` alignment records whose `.read` is the index
    of the record's read in THIS array.  `pull_mates`: a k-mer-screen hit also recruits the read's mate (north_star: "candidate
    read pairs").  `d_nmask`: the reads' N masks (files only; the synthetic reads have no N).  `n_total` = the library's reads
    over all ranks, `first_pair` = this rank's first pair (contiguous shards)."""

    def __init__(self, name, is_mean, is_sd, n_reads, d_reads, d_recs, n_recs=None, pull_mates=1, d_nmask=None, n_total=None,
                 first_pair=0, tag_ctx=None, screen=True, d_rec_keys=None):
        self.name, self.is_mean, self.is_sd, self.pull_mates = name, int(is_mean), int(is_sd), int(pull_mates)
        self.n_reads, self.d_reads, self.d_recs, self.d_nmask = int(n_reads), d_reads, d_recs, d_nmask
        self.n_recs = int(n_reads if n_recs is None else n_recs)
        self.n_total = int(self.n_reads if n_total is None else n_total)
        self.first_pair, self.n_pairs = int(first_pair), self.n_reads // 2
        self.tag_ctx = tag_ctx          # a second GapFill (second stream) for the tagger + second hop, or None: the pipeline's
        self.screen = bool(screen)      # False: alignment-based recruitment only (the reference's own mode)
        self.d_rec_keys = d_rec_keys    # the records' key column (gf_alnrec_keys_dev); None: Pipeline.add_library builds it
        self.counts = {}


class Results:
    """What one step left on the device, fetched once (contigs, their bases, the pick words, the pools when asked for)."""
    pass


class Pipeline:
    def __init__(self, gf, n_gaps, read_len, k_pairs, device=None, world=1, rank=0, backend="nccl", force_exchange=False,
                 min_count=2, min_contig=40, anchors=(30, 15), clip_dist=250, anchor_mapq=30, k_screen=None, keep_read_ids=False,
                 key_column=True, merge_in_step=False, merge_max_set=128):
        """gf: a GapFill whose gaps (and flanks, when a library is screened) are set.  k_pairs: [(k, k_velvet)] of
        assemble_gaps.py:87-122.  The screen runs at the SMALLEST k of the list: a read that shares a 51-mer with a flank shares
        its 31-mers too, so this is the superset every assembly k needs (the reference recruits once, then assembles at every k).
        anchors: flank-anchor lengths of the two pick rounds (the reference's bwa scores 30 then 15, assemble_gaps.py:336, 365);
        clip_dist / anchor_mapq: main.py:215-216.  key_column: the libraries keep an 8-byte key per alignment record beside the records
        ({pos, scaffold | MAPQ-0 bit}, built once when a library is added: 7.2 GB for C4's 900 M records) and the tagger streams THAT —
        a record far from every gap, 99 % of a BAM, is decided by (scaffold, position) alone (the reference's `focal_region.has_key(POS)`,
        collect_reads_for_gaps.py:104) — fetching the 32-byte record only of what passes its bin maps."""
        self.gf, self.lib, self.h = gf, B.lib(), gf.handle
        self.n_gaps, self.L, self.kk = int(n_gaps), int(read_len), [(int(a), int(b)) for a, b in k_pairs]
        self.rb = self.lib.gf_packed_read_bytes(self.L)
        self.dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
        self.world, self.rank, self.backend = int(world), int(rank), backend
        self.multi = self.world > 1 or bool(force_exchange)
        self.coll_dev = self.dev if backend == "nccl" else torch.device("cpu")
        self.min_count, self.min_contig, self.anchors = int(min_count), int(min_contig), tuple(int(a) for a in anchors)
        self.clip_dist, self.anchor_mapq = int(clip_dist), int(anchor_mapq)
        self.k_screen = int(k_screen) if k_screen else (min(a for a, _ in self.kk) if self.kk else 31)
        self.keep_read_ids = bool(keep_read_ids)
        self.key_column = bool(key_column)
        # merge_in_step: the contig-merge round (assemble_gaps.py:301-306 run_contigs_merge: dedup + ContigsMerger per gap) runs INSIDE the
        # step, on the device, for the gaps the first pick leaves open, followed by a second pick over the merged contigs — the reference
        # merges before it picks (:335-339); a gap the pick closes from its own contigs gains nothing from merging.  Merged contigs are
        # appended to the step's contig list with k = kv = 0.  Sets of more than merge_max_set contigs after the dedup are left alone.
        self.merge_in_step, self.merge_max_set = bool(merge_in_step), int(merge_max_set)
        self.libs = []
        self.batch = SH.owner_batch(self.n_gaps, self.world)
        self.stream = None
        if self.multi:
            # the collectives are ordered against the kernels by running everything on ONE torch side stream (not the legacy
            # default stream): the library adopts it
            self.stream = torch.cuda.Stream(device=self.dev)
            self._chk(self.lib.gf_set_stream(self.h, C.c_void_p(self.stream.cuda_stream)), "gf_set_stream")
        self.tag_after_filter = False
        # True: the alignment tagger of the NEXT step's records is launched on the libraries' second stream right after this step's pools
        # are built, i.e. beside the assembly (latency-bound, little HBM traffic) instead of in a row with the memory-bound filter: a
        # software pipeline over successive steps (every step still runs one tagger pass; prepare() launches the first)
        self.tag_ahead = False
        self.assemble_in_step = True      # False: step() stops at the pools; the caller runs assemble() itself (N masks of file-born pools)
        self.fixed_spans, self.fixed_on = [], False
        self.screen_dropped = 0
        self.prepared = False

    # ---- plumbing ---------------------------------------------------------------------------------------------------------
    def _chk(self, rc, what):
        if rc:
            raise B.GapFillError(rc, what, self.lib.gf_last_error(self.h).decode())

    def _u8(self, n):
        return torch.empty(max(1, int(n)), dtype=torch.uint8, device=self.dev)

    def contexts(self):
        """The distinct GapFill contexts (= streams) the step runs on."""
        return list({id(x): x for x in [self.gf] + [lb.tag_ctx for lb in self.libs if lb.tag_ctx is not None]}.values())

    def sync(self):
        for g in self.contexts():
            g.sync()

    def add_library(self, lb, hit_cap=None):
        """Registers a library and allocates its recruit buffers.  hit_cap (hits of one kind per step) defaults to an eighth of the
        reads — the screen finds 0.5 %, the tagger 1-2 % on the BASELINE workloads —; prepare() grows the buffers of a library that finds
        more (the device calls report their counts beyond the capacity) and sizes again; a STEP that outgrows them later fails loudly
        in fetch()."""
        assert not self.prepared
        dev = self.dev
        lb.h2 = lb.tag_ctx.handle if lb.tag_ctx is not None else self.h
        lb.second_stream = lb.h2.value != self.h.value
        self._alloc_hit_buffers(lb, int(hit_cap) if hit_cap else max(1 << 20, max(lb.n_reads, lb.n_recs) // 8))
        lb.d_pool_off = torch.zeros(self.n_gaps + 1, dtype=torch.int64, device=dev)
        lb.d_cnt = torch.zeros(32, dtype=torch.int32, device=dev)
        lb.cp = lb.d_cnt.data_ptr()
        lb.d_ids = None
        if self.key_column and lb.d_rec_keys is None:
            lb.d_rec_keys = torch.empty(lb.n_recs + 1, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            self._chk(self.lib.gf_alnrec_keys_dev(self.h, lb.d_recs.data_ptr(), lb.n_recs, lb.d_rec_keys.data_ptr()), "gf_alnrec_keys_dev")
            self.gf.sync()
        self.libs.append(lb)
        return lb

    def _alloc_hit_buffers(self, lb, hit_cap):
        lb.hit_cap = int(hit_cap)
        lb.d_hits = self._u8(lb.hit_cap * 8)
        lb.d_thits = self._u8(lb.hit_cap * 12)
        lb.d_lhits = self._u8(lb.hit_cap * 12)
        lb.d_low = self._u8(lb.hit_cap * 12)           # MAPQ==0 records compacted by the tagger (2 % of a typical BAM)
        lb.key_cap = 4 * lb.hit_cap
        lb.d_keys = torch.empty(lb.key_cap, dtype=torch.int64, device=self.dev)

    # ---- the phases of a step ---------------------------------------------------------------------------------------------
    def recruit(self, lb, tagger=True):
        lib, h = self.lib, self.h
        if lb.second_stream and tagger:
            self._chk(lib.gf_stream_wait(lb.h2, h), "gf_stream_wait")          # the previous consumers of the tagger buffers are done
            if self.tag_after_filter:
                # the tagger (a pure 32-B-record stream) starts when the k-mer FILTER has finished and runs beside the latency-bound
                # verification pass: beside the filter the two only took turns on the memory system (C4: 36.5 ms together, 26.6 + 10.5 alone)
                self._chk(lib.gf_stream_wait_after_filter(lb.h2, h), "gf_stream_wait_after_filter")
        if lb.screen:
            self._chk(lib.gf_screen_reads_dev(h, lb.d_reads.data_ptr(), lb.d_nmask.data_ptr() if lb.d_nmask is not None else None, lb.n_reads,
                                              self.L, self.k_screen, 1, lb.d_hits.data_ptr(), lb.hit_cap, lb.cp), "gf_screen_reads_dev")
        if tagger:
            self.tagger(lb)

    def tagger(self, lb):
        if self.key_column and lb.d_rec_keys is not None:
            self._chk(self.lib.gf_tag_alignments_keys_dev(lb.h2, lb.d_recs.data_ptr(), lb.d_rec_keys.data_ptr(), lb.n_recs, lb.is_mean, lb.is_sd, self.clip_dist,
                                                          self.anchor_mapq, lb.d_thits.data_ptr(), lb.hit_cap, lb.cp + 4 * CNT_TAG, lb.d_low.data_ptr(),
                                                          lb.hit_cap, lb.cp + 4 * CNT_LOW), "gf_tag_alignments_keys_dev")
            return
        self._chk(self.lib.gf_tag_alignments_low_dev(lb.h2, lb.d_recs.data_ptr(), lb.n_recs, lb.is_mean, lb.is_sd, self.clip_dist, self.anchor_mapq,
                                                     lb.d_thits.data_ptr(), lb.hit_cap, lb.cp + 4 * CNT_TAG, lb.d_low.data_ptr(), lb.hit_cap,
                                                     lb.cp + 4 * CNT_LOW), "gf_tag_alignments_low_dev")

    def _fixed_mark(self):
        if not self.fixed_on:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _fixed_span(self, name, e0):
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.fixed_spans.append((name, e0, e1))

    def hop_and_keys(self, lb):
        lib, h, world = self.lib, self.h, self.world
        self._chk(lib.gf_second_hop_table_dev(lb.h2, lb.d_recs.data_ptr(), lb.d_thits.data_ptr(), lb.cp + 4 * CNT_TAG, lb.hit_cap,
                                              lb.d_rows.data_ptr(), lb.d_row_gap.data_ptr(), lb.row_cap, lb.cp + 4 * CNT_ROWS),
                  "gf_second_hop_table_dev")
        if not self.multi:
            rows_p, rowgap_p, nrows_p, rcap, hh = lb.d_rows.data_ptr(), lb.d_row_gap.data_ptr(), lb.cp + 4 * CNT_ROWS, lb.row_cap, lb.h2
        else:
            # the reads are sharded, the second hop is not: a MAPQ-0 record is linked to discordant mates of ANY rank's reads
            # (collect_discordant_low_mapq_reads.py reads the whole discordant_reads_pos file), so the ranks all-gather their
            # rows (fixed-size slots) and every rank merges the union
            if lb.second_stream:
                self._chk(lib.gf_stream_wait(h, lb.h2), "gf_stream_wait")
            ev0 = self._fixed_mark()
            # ONE all-gather: rows (16 B each), their gaps (4 B each) and the row count travel as one packed slot per rank
            # (lb.d_hop_pack: the table kernel writes rows and gaps straight into it), unpacked by three strided device copies
            rc16, pb = lb.row_cap * 16, lb.hop_pack_bytes
            lb.d_hop_pack[pb - 16:pb - 12].copy_(lb.d_cnt[CNT_ROWS:CNT_ROWS + 1].view(torch.uint8))
            SH.all_gather_slots(lb.d_hop_all, lb.d_hop_pack, self.backend)
            pk = lb.d_hop_all.view(world, pb)
            lb.d_rows_all.view(world, rc16).copy_(pk[:, :rc16])
            lb.d_rowgap_all.view(torch.uint8).view(world, lb.row_cap * 4).copy_(pk[:, rc16:rc16 + lb.row_cap * 4])
            lb.d_nrows_all.view(torch.uint8).view(world, 4).copy_(pk[:, pb - 16:pb - 12])
            self._chk(lib.gf_second_hop_table_merge_dev(h, lb.d_rows_all.data_ptr(), lb.d_rowgap_all.data_ptr(), lb.d_nrows_all.data_ptr(), world,
                                                        lb.row_cap, lb.d_rows_u.data_ptr(), lb.d_rowgap_u.data_ptr(), world * lb.row_cap,
                                                        lb.d_nrows_u.data_ptr()), "gf_second_hop_table_merge_dev")
            self._fixed_span("second_hop_union", ev0)
            rows_p, rowgap_p, nrows_p, rcap, hh = lb.d_rows_u.data_ptr(), lb.d_rowgap_u.data_ptr(), lb.d_nrows_u.data_ptr(), world * lb.row_cap, h
        self._chk(lib.gf_tag_low_mapq_table_dev(hh, lb.d_low.data_ptr(), lb.cp + 4 * CNT_LOW, lb.hit_cap, rows_p, nrows_p, rcap,
                                                lb.d_lhits.data_ptr(), lb.hit_cap, lb.cp + 4 * CNT_HOP), "gf_tag_low_mapq_table_dev")
        lb.rowgap_p = rowgap_p
        if hh.value != h.value:
            self._chk(lib.gf_stream_wait(h, lb.h2), "gf_stream_wait")          # pools need the tagger's and the second hop's hits
        self._chk(lib.gf_pool_keys_all_dev(h, lb.d_hits.data_ptr(), lb.cp + 4 * CNT_SCREEN, lb.hit_cap, lb.pull_mates,
                                           lb.d_recs.data_ptr(), lb.d_thits.data_ptr(), lb.cp + 4 * CNT_TAG, lb.hit_cap, lb.d_lhits.data_ptr(),
                                           lb.cp + 4 * CNT_HOP, lb.hit_cap, rowgap_p, lb.d_keys.data_ptr(), lb.key_cap, lb.cp + 4 * CNT_KEYS),
                  "gf_pool_keys_all_dev")

    def build_pools(self, lb, pool_ptr, pool_cap):
        self._chk(self.lib.gf_build_pools_dev(self.h, lb.d_reads.data_ptr(), lb.n_reads, self.L, lb.d_keys.data_ptr(), lb.cp + 4 * CNT_KEYS,
                                              lb.key_cap, pool_ptr, pool_cap, lb.d_pool_off.data_ptr(),
                                              lb.d_ids.data_ptr() if (lb.d_ids is not None and pool_ptr) else None, lb.cp + 4 * CNT_POOLERR),
                  "gf_build_pools_dev")

    def _on_stream(self, fn):
        if self.stream is not None:
            with torch.cuda.stream(self.stream):
                return fn()
        return fn()

    # ---- sizing pass (untimed): second-hop table rows, pooled reads, exchange slots, assembly workspace --------------------
    def prepare(self, strict_screen=False, _depth=0):
        import torch.distributed as dist
        lib, h, dev, world, n_gaps, rb = self.lib, self.h, self.dev, self.world, self.n_gaps, self.rb
        n_lib = len(self.libs)
        assert n_lib >= 1
        torch.cuda.synchronize()
        for attempt in range(4):
            self.screen_dropped = 0
            for lb in self.libs:
                self._on_stream(lambda: self.recruit(lb))
                if lb.screen:
                    nd = C.c_size_t(0)
                    self._chk(lib.gf_screen_last_overflow(h, C.byref(nd)), "gf_screen_last_overflow")
                    # reads with more (position, gap) matches than the verification lists (low-complexity reads against hundreds of flanks)
                    self.screen_dropped += nd.value
            self.sync()
            # a library that recruits more than an eighth of its reads (dense gaps, short scaffolds): the device calls report how many hits
            # they found beyond the capacity — the hit buffers grow to that and the pass runs again (the reference has no such bound)
            grown = False
            for lb in self.libs:
                c = lb.d_cnt.cpu().numpy()
                need = max(int(c[CNT_SCREEN]) if lb.screen else 0, int(c[CNT_TAG]), int(c[CNT_LOW]))
                if need > lb.hit_cap:
                    self._alloc_hit_buffers(lb, int(1.25 * need) + 1024)
                    grown = True
            if self.multi:      # (every rank repeats the pass together: the sizes below are all-reduced)
                g_t = torch.tensor([int(grown)], dtype=torch.int64, device=self.coll_dev)
                dist.all_reduce(g_t, op=dist.ReduceOp.MAX)
                grown = bool(int(g_t))
            if not grown:
                break
            torch.cuda.synchronize()
        if strict_screen and self.screen_dropped:
            raise RuntimeError("%d reads were not verified in full by the k-mer screen" % self.screen_dropped)
        for lb in self.libs:
            n_th = int(lb.d_cnt[CNT_TAG])
            self._check_cap(n_th, lb.hit_cap, "tagger hits", lb)
            th = np.frombuffer(lb.d_thits[:n_th * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
            lb.row_cap = 2 * int((th["kind"] == B.KIND_DISCORDANT).sum()) + 4096
            if self.multi:
                rc_t = torch.tensor([lb.row_cap], dtype=torch.int64, device=self.coll_dev)
                dist.all_reduce(rc_t, op=dist.ReduceOp.MAX)
                lb.row_cap = int(rc_t)
                lb.d_rows_all = self._u8(world * lb.row_cap * 16)
                lb.d_rowgap_all = torch.empty(world * lb.row_cap, dtype=torch.int32, device=dev)
                lb.d_nrows_all = torch.zeros(world, dtype=torch.int32, device=dev)
                lb.d_rows_u = self._u8(world * lb.row_cap * 16)
                lb.d_rowgap_u = torch.empty(world * lb.row_cap, dtype=torch.int32, device=dev)
                lb.d_nrows_u = torch.zeros(4, dtype=torch.int32, device=dev)
            # (rows, their gaps and — multi-rank runs — the row count in ONE buffer: the slot of the second hop's single all-gather)
            lb.hop_pack_bytes = lb.row_cap * 20 + 16
            lb.d_hop_pack = torch.zeros(lb.hop_pack_bytes, dtype=torch.uint8, device=dev)
            lb.d_rows = lb.d_hop_pack[:lb.row_cap * 16]
            lb.d_row_gap = lb.d_hop_pack[lb.row_cap * 16:lb.row_cap * 20].view(torch.int32)
            if self.multi:
                lb.d_hop_all = torch.zeros(world * lb.hop_pack_bytes, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize()

        def sizing_pools():
            for lb in self.libs:
                self.hop_and_keys(lb)
                self.build_pools(lb, None, 0)        # offsets only
        self._on_stream(sizing_pools)
        self.sync()
        torch.cuda.synchronize()
        again = False
        for lb in self.libs:      # second-hop hits and pool keys have their own counts: grow for them too, then size everything once more
            n_hop, n_keys = int(lb.d_cnt[CNT_HOP]), int(lb.d_cnt[CNT_KEYS])
            if n_hop > lb.hit_cap or n_keys > lb.key_cap:
                self._alloc_hit_buffers(lb, int(1.25 * max(n_hop, (n_keys + 3) // 4)) + 1024)
                again = True
        if self.multi:
            g_t = torch.tensor([int(again)], dtype=torch.int64, device=self.coll_dev)
            dist.all_reduce(g_t, op=dist.ReduceOp.MAX)
            again = bool(int(g_t))
        if again:
            if _depth >= 3:
                raise RuntimeError("the hit buffers keep overflowing (%s)" % ", ".join("%s: %d" % (lb.name, lb.hit_cap) for lb in self.libs))
            return self.prepare(strict_screen, _depth + 1)
        self.rows_lib = [int(lb.d_pool_off[-1]) for lb in self.libs]
        # largest merged pool (all libraries, all ranks): bounds the assembly's per-workgroup workspace slices
        per_gap = sum((lb.d_pool_off[1:] - lb.d_pool_off[:-1]) for lb in self.libs).to(self.coll_dev)
        if self.multi:
            dist.all_reduce(per_gap, op=dist.ReduceOp.SUM)
        self.per_gap = per_gap.cpu().numpy()
        self.max_pool_rows = int(self.per_gap.max()) if n_gaps else 0
        # bound on the rows of one pool = the workspace slice of the assembly's main launch.  Deeper pools are not an error (they take the
        # assembly's last launch, option asm_big_pool_reads), so the bound needs no safety margin, and where a few repeat gaps hold many
        # times the reads of the others it follows the bulk of the pools, not the deepest one
        self.asm_bound = self.max_pool_rows if self.max_pool_rows <= 4096 else max(4096, int(np.percentile(self.per_gap, 99)))
        self.gf.set_option("asm_max_pool_reads", max(1, self.asm_bound))
        # ... and the slices of that last launch hold the deepest pool actually present (the library's default, 131 072 rows x 8 slices, is
        # 7-9 GB of workspace per context whether or not a pool needs it: ADVICE r4)
        self.gf.set_option("asm_big_pool_reads", max(1, min(0x1FFFFF, self.max_pool_rows)))
        self.lib_cap = max(4096, int(1.25 * max(self.rows_lib)) + 1024)            # rows of one library's pool array
        # local pool arrays: [n_lib][lib_cap] rows (slot l = library l: the source layout of gf_pools_merge_dev)
        self.d_pools = self._u8(n_lib * self.lib_cap * rb + 64)
        self.pool_ptr = [self.d_pools.data_ptr() + l * self.lib_cap * rb for l in range(n_lib)]
        if self.keep_read_ids:
            for lb in self.libs:
                lb.d_ids = torch.empty(self.lib_cap, dtype=torch.int32, device=dev)
        self.d_xerr = torch.zeros(4, dtype=torch.int32, device=dev)
        self.d_libcnt = torch.zeros(n_lib * n_gaps, dtype=torch.int32, device=dev)         # [n_lib][n_gaps]
        if self.multi:
            owner = SH.gap_owner(n_gaps, world).to(dev)
            per_dst = torch.zeros(n_lib, world, dtype=torch.int64, device=dev)
            for l, lb in enumerate(self.libs):
                cnt = (lb.d_pool_off[1:] - lb.d_pool_off[:-1])
                per_dst[l].index_add_(0, owner, cnt)
            mx = per_dst.max().to(self.coll_dev)
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            self.slot_cap = max(1024, int(1.25 * int(mx)) + 256)
            tot = torch.stack([per_dst[:, r].sum() for r in range(world)]).to(self.coll_dev)
            dist.all_reduce(tot, op=dist.ReduceOp.SUM)
            self.merged_cap = max(4096, int(1.25 * int(tot.max())) + 1024)
            # exact split sizes (default): rows per (source, owner, library) from this sizing pass, the same table on every rank — one
            # all-to-all per step, nothing padded, the counts inside it.  GF_XCHG=slots: equal-sized slots padded to 1.25 x the largest
            # block + an all-gather of the counts (the form of rounds 2-5)
            self.exact_exchange = os.environ.get("GF_XCHG", "exact") != "slots"
            if self.exact_exchange:
                self.xchg_rows = SH.exchange_rows_table(per_dst, self.coll_dev, self.backend)
                self.xchg = SH.ExactOwnerExchange(world, self.rank, n_lib, n_gaps, self.xchg_rows, rb, dev, self.backend)
            else:
                self.xchg = SH.OwnerExchange(world, n_lib, n_gaps, self.slot_cap, rb, dev, self.backend)
        else:
            self.slot_cap = self.lib_cap
            self.merged_cap = max(4096, int(1.25 * sum(self.rows_lib)) + 1024)
        self.need_merge = self.multi or n_lib > 1
        self.d_merged = self._u8(self.merged_cap * rb + 64) if self.need_merge else None
        self.d_moff = torch.zeros(n_gaps + 1, dtype=torch.int64, device=dev)
        # (contigs: a few dozen per gap and k on an i.i.d. draft; the deep pools of a repeat-bearing draft fragment into many more:
        # room grows with the pooled reads)
        nk = max(1, len(self.kk))
        self.contig_cap = (64 * n_gaps + 4096 + sum(self.rows_lib) // 4) * nk
        self.seq_cap = (24576 * n_gaps + (1 << 20) + 32 * sum(self.rows_lib)) * nk
        self.d_ctg = self._u8(self.contig_cap * 32)
        self.d_seq = self._u8(self.seq_cap)
        self.d_gap_err = torch.zeros(max(1, n_gaps), dtype=torch.int32, device=dev)
        self.d_best = torch.zeros(max(1, n_gaps), dtype=torch.int64, device=dev)
        # assembly counters: 0 contigs (u32), 2-3 contig bases (u64), 4 gaps closed (u32)
        self.d_acnt = torch.zeros(8, dtype=torch.int32, device=dev)
        self.ap = self.d_acnt.data_ptr()
        self.d_mstats = torch.zeros(B.MG_WORDS, dtype=torch.int32, device=dev)      # statistics of the merge round (gf_merge_open_gaps_dev)
        pr = np.zeros(1, dtype=B.OVL_PARAMS)
        pr[0] = tuple(self.gf.MERGER_PARAMS)[:7] + (0.0,)       # ContigsMerger's options as GAPPadder sets them (MergeContigs.py:75)
        self.merge_params = pr
        self.k_arr = (C.c_int * nk)(*[a for a, _ in self.kk])
        self.kv_arr = (C.c_int * nk)(*[b for _, b in self.kk])
        torch.cuda.synchronize()
        if self.tag_ahead:          # the first step's tagger pass (untimed, like a warm-up step's)
            assert all(lb.second_stream for lb in self.libs), "tag_ahead needs DeviceLibrary(tag_ctx=...)"
            self._on_stream(lambda: [self.tagger(lb) for lb in self.libs])
            self.sync()
        self.prepared = True

    def _check_cap(self, n, cap, what, lb):
        if n > cap:
            raise RuntimeError("library %s: %d %s exceed the capacity %d (add_library(hit_cap=...))" % (lb.name, n, what, cap))

    # ---- one pass of the hot path (enqueued; no host synchronisation) ------------------------------------------------------
    def _step(self, recruited=False):
        """recruited: the hit lists, second-hop rows and pool keys of the libraries are in place already (the sizing pass just left them:
        finish()) — only the pools are built."""
        lib, h, n_gaps, n_lib, L = self.lib, self.h, self.n_gaps, len(self.libs), self.L
        if not recruited:
            for lb in self.libs:
                self.recruit(lb, tagger=not self.tag_ahead)
        for l, lb in enumerate(self.libs):
            if not recruited:
                self.hop_and_keys(lb)
            self.build_pools(lb, self.pool_ptr[l], self.lib_cap)
        if self.tag_ahead:
            for lb in self.libs:        # the next step's tagger pass: behind every consumer of this step's hits, beside the assembly
                self._chk(lib.gf_stream_wait(lb.h2, h), "gf_stream_wait")
                self.tagger(lb)
        # (zeroed through the library = on its stream; a torch op here would run on torch's stream)
        self._chk(lib.gf_memset_dev(h, self.d_xerr.data_ptr(), 0, 16) or lib.gf_memset_dev(h, self.d_best.data_ptr(), 0, 8 * max(1, n_gaps))
                  or lib.gf_memset_dev(h, self.ap + 16, 0, 16), "gf_memset_dev")
        if not self.need_merge:
            self.asm_ptr, self.asm_off, self.asm_rows = self.pool_ptr[0], self.libs[0].d_pool_off.data_ptr(), self.lib_cap
        elif not self.multi:
            for l, lb in enumerate(self.libs):
                self._chk(lib.gf_pool_counts_dev(h, lb.d_pool_off.data_ptr(), n_gaps, self.d_libcnt.data_ptr() + 4 * l * n_gaps), "gf_pool_counts_dev")
            self._chk(lib.gf_pools_merge_dev(h, self.d_pools.data_ptr(), self.lib_cap, self.d_libcnt.data_ptr(), n_lib, 1, n_gaps, L, 0, 1, self.batch,
                                             self.d_merged.data_ptr(), self.merged_cap, self.d_moff.data_ptr(), self.d_xerr.data_ptr()),
                      "gf_pools_merge_dev")
            self.asm_ptr, self.asm_off, self.asm_rows = self.d_merged.data_ptr(), self.d_moff.data_ptr(), self.merged_cap
        else:
            # the one exchange step: rows regrouped by owner rank, counts all-gathered, slots all-to-all'ed
            # (equal-sized slots: no host sizes, no host sync), owners merge in (library, source rank) order
            def pack(l, send, cap, cnt):
                self._chk(lib.gf_pools_pack_for_owners_dev(h, self.pool_ptr[l], self.libs[l].d_pool_off.data_ptr(), n_gaps, L, self.world, self.batch,
                                                           l, n_lib, send.data_ptr(), cap, cnt.data_ptr(), self.d_xerr.data_ptr()),
                          "gf_pools_pack_for_owners_dev")

            def merge(recv, cap, all_cnt):
                self._chk(lib.gf_pools_merge_dev(h, recv.data_ptr(), cap, all_cnt.data_ptr(), n_lib, self.world, n_gaps, L, self.rank, self.world,
                                                 self.batch, self.d_merged.data_ptr(), self.merged_cap, self.d_moff.data_ptr(), self.d_xerr.data_ptr()),
                          "gf_pools_merge_dev")
            def pack_v(l, send, slot_base, slot_cap, cnt_base, cnt):
                self._chk(lib.gf_pools_pack_for_owners_v_dev(h, self.pool_ptr[l], self.libs[l].d_pool_off.data_ptr(), n_gaps, L, self.world, self.batch,
                                                             l, n_lib, send.data_ptr(), slot_base.data_ptr(), slot_cap.data_ptr(), cnt_base.data_ptr(),
                                                             cnt.data_ptr(), self.d_xerr.data_ptr()), "gf_pools_pack_for_owners_v_dev")

            def merge_v(recv, slot_base, cnt_base):
                self._chk(lib.gf_pools_merge_v_dev(h, recv.data_ptr(), slot_base.data_ptr(), cnt_base.data_ptr(), n_lib, self.world, n_gaps, L, self.rank,
                                                   self.world, self.batch, self.d_merged.data_ptr(), self.merged_cap, self.d_moff.data_ptr(),
                                                   self.d_xerr.data_ptr()), "gf_pools_merge_v_dev")
            ev0 = self._fixed_mark()
            if self.exact_exchange:
                self.xchg.run(pack_v, merge_v)
            else:
                self.xchg.run(pack, merge)
            self._fixed_span("owner_exchange", ev0)
            self.asm_ptr, self.asm_off, self.asm_rows = self.d_merged.data_ptr(), self.d_moff.data_ptr(), self.merged_cap
        if self.kk and self.assemble_in_step:
            self.assemble()

    def assemble(self, d_nmask=None):
        """Assembly of the pools the step left (every (k, kv) pair) + flank anchoring; also callable on its own after the pools
        changed (a second recruitment round, assemble_gaps.py:349-351)."""
        lib, h = self.lib, self.h
        self._chk(lib.gf_assemble_multi_dev(h, self.asm_ptr, d_nmask, self.asm_off, self.n_gaps, self.asm_rows, self.L, self.k_arr, self.kv_arr,
                                            len(self.kk), self.min_count, self.min_contig, self.d_ctg.data_ptr(), self.contig_cap, self.ap,
                                            self.d_seq.data_ptr(), self.seq_cap, self.ap + 8, self.d_gap_err.data_ptr()), "gf_assemble_multi_dev")
        # which gaps are closed: both flanks anchored on one contig (pick_contigs.py:64-358; scores 30 then 15, assemble_gaps.py:336, 365)
        a0, a1 = self.anchors[0], (self.anchors[1] if len(self.anchors) > 1 else 0)
        self._chk(lib.gf_pick_anchored2_dev(h, self.d_ctg.data_ptr(), self.ap, self.contig_cap, self.d_seq.data_ptr(), a0, a1,
                                            self.d_best.data_ptr(), self.ap + 16), "gf_pick_anchored2_dev")
        if self.merge_in_step:
            # the open gaps' contigs through the contig merger, merged contigs appended (k = kv = 0), second pick over THEM only
            self._chk(lib.gf_merge_open_gaps_dev(h, self.d_ctg.data_ptr(), self.ap, self.contig_cap, self.d_seq.data_ptr(), self.ap + 8, self.seq_cap,
                                                 self.d_best.data_ptr(), self.n_gaps, B._p(self.merge_params), 10, self.merge_max_set,
                                                 self.k_arr, self.kv_arr, min(16, len(self.kk)), self.d_mstats.data_ptr()), "gf_merge_open_gaps_dev")
            self._chk(lib.gf_pick_anchored2_from_dev(h, self.d_ctg.data_ptr(), self.ap, self.contig_cap, self.d_seq.data_ptr(), a0, a1,
                                                     self.d_mstats.data_ptr() + 4 * B.MG_N0, self.d_best.data_ptr(), self.ap + 16),
                      "gf_pick_anchored2_from_dev")

    def step(self, n=1):
        assert self.prepared, "Pipeline.prepare() first"
        def run():
            for _ in range(n):
                self._step()
        self._on_stream(run)

    def finish(self):
        """A one-shot run (the CLI): prepare() has recruited every library once to size the buffers, and its hits and pool keys are still
        there — build the pools from them (and assemble, unless assemble_in_step is off) instead of recruiting a second time."""
        assert self.prepared, "Pipeline.prepare() first"
        self._on_stream(lambda: self._step(recruited=True))

    def barrier(self):
        self.sync()
        torch.cuda.synchronize()
        if self.multi:
            import torch.distributed as dist
            dist.barrier()
        torch.cuda.synchronize()

    # ---- results of the last step -----------------------------------------------------------------------------------------
    def fetch(self, pools=False):
        """Synchronises and copies the last step's results to the host; raises on any overflow flag."""
        self.sync()
        torch.cuda.synchronize()
        r = Results()
        acnt = self.d_acnt.cpu().numpy()
        r.n_contigs, r.n_seq, r.n_closed = int(acnt[0]), int(acnt[2:4].view(np.uint64)[0]), int(acnt[4])
        xerr = int(self.d_xerr[0])
        for lb in self.libs:
            c = lb.d_cnt.cpu().numpy()
            lb.counts = {"screen_hits": int(c[CNT_SCREEN]) if lb.screen else 0, "tagger_hits": int(c[CNT_TAG]), "second_hop_hits": int(c[CNT_HOP]),
                         "pool_keys": int(c[CNT_KEYS]), "pooled_reads": int(lb.d_pool_off[-1])}
            for n_, cap, what in ((lb.counts["screen_hits"], lb.hit_cap, "screen hits"), (int(c[CNT_TAG]), lb.hit_cap, "tagger hits"),
                                  (int(c[CNT_HOP]), lb.hit_cap, "second-hop hits"), (int(c[CNT_KEYS]), lb.key_cap, "pool keys"),
                                  (int(c[CNT_LOW]), lb.hit_cap, "MAPQ-0 records"), (int(c[CNT_ROWS]), lb.row_cap, "second-hop rows")):
                self._check_cap(n_, cap, what, lb)
            if int(c[CNT_POOLERR]):
                raise RuntimeError("library %s: pool buffer too small (flag %#x)" % (lb.name, int(c[CNT_POOLERR]) & 0xFFFFFFFF))
        n_err = int(self.d_gap_err.sum())
        if xerr or n_err or r.n_contigs > self.contig_cap or r.n_seq > self.seq_cap:
            raise RuntimeError("step overflow: exchange/merge flag %#x, %d gap errors, %d contigs (cap %d), %d contig bases (cap %d)"
                               % (xerr & 0xFFFFFFFF, n_err, r.n_contigs, self.contig_cap, r.n_seq, self.seq_cap))
        r.merge = None
        if self.merge_in_step and self.kk:
            ms = self.d_mstats.cpu().numpy().view(np.uint32)
            if int(ms[B.MG_ERR]):
                raise RuntimeError("merge round: capacity flags %#x" % int(ms[B.MG_ERR]))
            r.merge = {"gaps_tried": int(ms[B.MG_N_SETS]), "gaps_skipped_large": int(ms[B.MG_SKIPPED]), "gaps_skipped_graph": int(ms[B.MG_SKIPPED_GRAPH]), "candidate_pairs": int(ms[B.MG_N_PAIRS]),
                       "edges": int(ms[B.MG_N_EDGES]), "new_contigs": int(ms[B.MG_N_JOBS]), "gaps_with_new_contigs": int(ms[B.MG_SETS_WITH_JOBS]),
                       "contigs_before": int(ms[B.MG_N0])}
        r.asm_off_t = self.d_moff if self.need_merge else self.libs[0].d_pool_off
        r.asm_pool_t = self.d_merged if self.need_merge else self.d_pools
        r.asm_rows_total = int(r.asm_off_t[-1])
        r.contigs = np.frombuffer(self.d_ctg[:r.n_contigs * 32].cpu().numpy().tobytes(), dtype=B.CONTIG)
        r.seq = self.d_seq[:r.n_seq].cpu().numpy().tobytes()
        r.best = self.d_best[:self.n_gaps].cpu().numpy().view(np.uint64)
        if r.merge is not None:      # gaps whose winning contig is a merged one
            idx = 0x7FFFFFFF - ((r.best >> np.uint64(1)) & np.uint64(0x7FFFFFFF)).astype(np.int64)
            r.merge["gaps_closed_by_merging"] = int(((r.best != 0) & (idx >= r.merge["contigs_before"])).sum())
        if pools:
            r.pool_off = r.asm_off_t.cpu().numpy().astype(np.int64)
            r.pool_rows = r.asm_pool_t[:r.asm_rows_total * self.rb].cpu().numpy().reshape(-1, self.rb)
        return r

    def merge_open_gaps(self, res, max_set=128):
        """The reference merges a gap's contigs BEFORE it picks (assemble_gaps.py:301-306, 335-339: run_contigs_merge, then
        pick_full_constructed_contigs); the step picks first, so only the gaps that pick left open can gain from merging: their
        contigs go through the contig merger (MergeContigs.merge_sets: exact-containment dedup, all-pairs prefilter + overlap
        evaluation on the GPU, path search on the host) and the NEW_CONTIG_MERGE sequences through a second pick on the device.
        res: fetch()'s Results.  Returns {"gaps_tried", "gaps_with_new_contigs", "new_contigs", "closed": {gap: (anchor, span + 1,
        index into "contigs", reverse)}, "contigs": [(gap, seq)]}.  Gaps with more than max_set contigs after the dedup are left alone
        ("gaps_skipped_large"): the contig graph of a repeat-bearing gap has thousands of paths (C2 with planted repeats and mate pairs:
        53 641 merged strings for 176 gaps, 158 s on the host, and 50 of the 174 gaps they close are closed with a wrong sequence)."""
        from .MergeContigs import MAX_SET, drop_contained, merge_sets
        ctg, seq = res.contigs, res.seq
        open_gaps = np.nonzero(res.best == 0)[0]
        order = np.argsort(ctg["gap"], kind="stable")
        gs = ctg["gap"][order]
        lo, hi = np.searchsorted(gs, open_gaps), np.searchsorted(gs, open_gaps, side="right")
        sets, gaps_of, skipped = [], [], 0
        for g, a, z in zip(open_gaps, lo, hi):
            if z - a >= 2:
                recs = [("c%d" % i, seq[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])].decode()) for i in order[a:z]]
                recs = drop_contained(recs) if len(recs) <= MAX_SET else recs
                if 2 <= len(recs) <= min(MAX_SET, max_set):
                    sets.append(recs)
                    gaps_of.append(int(g))
                elif len(recs) > max_set:
                    skipped += 1
        out = {"gaps_tried": len(sets), "gaps_skipped_large": skipped, "gaps_with_new_contigs": 0, "new_contigs": 0, "closed": {}, "contigs": []}
        if not sets:
            return out
        new = []
        for g, m in zip(gaps_of, merge_sets(self.gf, sets)):
            out["gaps_with_new_contigs"] += bool(m["new"])
            new += [(g, s) for _, s, _ in m["new"]]
        out["new_contigs"], out["contigs"] = len(new), new
        if not new:
            return out
        c2 = np.zeros(len(new), dtype=B.CONTIG)
        off = 0
        for i, (g, s) in enumerate(new):
            c2[i] = (g, 0, 0, 0, len(s), 0, 0, off)
            off += len(s)
        d_c2 = torch.from_numpy(c2.view(np.uint8).copy()).to(self.dev)
        d_s2 = torch.from_numpy(np.frombuffer("".join(s for _, s in new).encode(), dtype=np.uint8).copy()).to(self.dev)
        d_b2 = torch.zeros(max(1, self.n_gaps), dtype=torch.int64, device=self.dev)
        d_n2 = torch.tensor([len(new), 0, 0, 0], dtype=torch.int32, device=self.dev)
        torch.cuda.synchronize()
        a0, a1 = self.anchors[0], (self.anchors[1] if len(self.anchors) > 1 else 0)
        self._chk(self.lib.gf_pick_anchored2_dev(self.h, d_c2.data_ptr(), d_n2.data_ptr(), len(new), d_s2.data_ptr(), a0, a1, d_b2.data_ptr(),
                                                 d_n2.data_ptr() + 8), "gf_pick_anchored2_dev")
        self.gf.sync()
        b2 = d_b2[:self.n_gaps].cpu().numpy().view(np.uint64)
        out["closed"] = {int(g): decode_best(b2[g]) for g in np.nonzero(b2)[0]}
        out["arrays"] = (c2, "".join(s for _, s in new).encode(), b2)       # the second pick's contig table, bases and pick words
        return out

    def fixed_ms(self, steps):
        out = {}
        for name, e0, e1 in self.fixed_spans:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1) / steps
        return out


def decode_best(b):
    """gap_best word (gf_pick_anchored_dev) -> (anchor length, span + 1, contig index, reverse strand?)."""
    b = int(b)
    return b >> 56, (b >> 32) & 0xFFFFFF, 0x7FFFFFFF - ((b >> 1) & 0x7FFFFFFF), b & 1
