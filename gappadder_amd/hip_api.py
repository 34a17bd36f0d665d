"""Host-side object over the C ABI (include/gapfill_hip.h): numpy in, numpy out, HIP underneath."""
import ctypes as C

import numpy as np

from . import _lib as B


class GapFill:
    """One context per device (gf_init / gf_destroy)."""

    def __init__(self, device=0):
        self._L = B.lib()
        h = C.c_void_p()
        rc = self._L.gf_init(int(device), C.byref(h))
        if rc:
            raise B.GapFillError(rc, "gf_init(device=%d)" % device)
        self._h = h
        self.n_gaps = 0

    def close(self):
        if getattr(self, "_h", None):
            self._L.gf_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc, what):
        if rc:
            raise B.GapFillError(rc, what, self._L.gf_last_error(self._h).decode())

    @property
    def handle(self):
        return self._h

    def set_option(self, name, value):
        self._chk(self._L.gf_set_option(self._h, name.encode(), int(value)), "gf_set_option")

    def set_gaps(self, gaps, n_scaffolds, flanks=None):
        """gaps: structured array (B.GAP); flanks: list of (left, right) ASCII strings or None."""
        gaps = np.ascontiguousarray(gaps, dtype=B.GAP)
        if flanks is not None:
            parts, off = [], [0]
            for l, r in flanks:
                for s in (l, r):
                    parts.append(s)
                    off.append(off[-1] + len(s))
            blob = "".join(parts).encode()
            offs = np.asarray(off, dtype=np.uint64)
            rc = self._L.gf_set_gaps(self._h, B._p(gaps), len(gaps), int(n_scaffolds), blob, B._p(offs))
        else:
            rc = self._L.gf_set_gaps(self._h, B._p(gaps), len(gaps), int(n_scaffolds), None, None)
        self._chk(rc, "gf_set_gaps")
        self.n_gaps = len(gaps)

    @staticmethod
    def pack_reads(seqs, read_len, with_mask=False):
        """seqs: list of equal-length str/bytes or one bytes blob.  Returns (packed uint8 [n, rb], n_mask or None)."""
        L = B.lib()
        blob = seqs if isinstance(seqs, (bytes, bytearray)) else "".join(seqs).encode()
        n = len(blob) // read_len
        assert n * read_len == len(blob)
        rb = L.gf_packed_read_bytes(read_len)
        packed = np.zeros((n, rb), dtype=np.uint8)
        nm = np.zeros((n, (read_len + 31) // 32), dtype=np.uint32) if with_mask else None
        rc = L.gf_pack_reads(bytes(blob), n, read_len, B._p(packed), B._p(nm))
        if rc:
            raise B.GapFillError(rc, "gf_pack_reads")
        return packed, nm

    def fastq_pack(self, text, read_len, with_mask=True):
        """FASTQ text (bytes: a whole file or a chunk starting at a record boundary) -> (packed uint8 [n, rb], n_mask or None,
        hdr_begin uint64 [n] = byte offset of every record's '@' line, status bits) — parsed and packed on the GPU."""
        text = bytes(text)
        rb = self._L.gf_packed_read_bytes(read_len)
        cap = max(1, text.count(b"\n") // 4 + 1)
        packed = np.zeros((cap, rb), dtype=np.uint8)
        nm = np.zeros((cap, (read_len + 31) // 32), dtype=np.uint32) if with_mask else None
        hdr = np.zeros(cap + 1, dtype=np.uint64)
        n, st = C.c_size_t(0), C.c_uint32(0)
        self._chk(self._L.gf_fastq_pack(self._h, text, len(text), read_len, B._p(packed), cap, B._p(nm), B._p(hdr), C.byref(n),
                                        C.byref(st)), "gf_fastq_pack")
        return packed[:n.value], (nm[:n.value] if with_mask else None), hdr[:n.value], int(st.value)

    def sam_pack(self, text, names):
        """SAM text (bytes; alignment lines) -> (records B.ALNREC [n], line_begin uint64 [n]) parsed on the GPU; names = the
        scaffold names of the .fai in order.  Same records as sam_io.decode on the same lines."""
        text = bytes(text)
        blob = "".join(names).encode()
        off = np.zeros(len(names) + 1, dtype=np.uint32)
        off[1:] = np.cumsum([len(n.encode()) for n in names])
        cap = max(1, text.count(b"\n") + 1)
        recs = np.zeros(cap, dtype=B.ALNREC)
        lb = np.zeros(cap, dtype=np.uint64)
        n = C.c_size_t(0)
        self._chk(self._L.gf_sam_pack(self._h, text, len(text), blob, B._p(off), len(names), B._p(recs), cap, B._p(lb), C.byref(n)),
                  "gf_sam_pack")
        return recs[:n.value], lb[:n.value]

    def bgzf_inflate(self, bgzf, carry=b"", want_host=True):
        """BGZF file bytes (from a block boundary) -> (inflated bytes = carry + whole blocks' contents, n_consumed); inflated
        and CRC-checked on the GPU, where the stream also stays for bam_pack(None, ...) / bam_fetch.  want_host=False returns
        the stream length instead of a host copy of it."""
        bgzf, carry = bytes(bgzf), bytes(carry)
        n, used = C.c_size_t(0), C.c_size_t(0)
        if not want_host:
            self._chk(self._L.gf_bgzf_inflate(self._h, bgzf, len(bgzf), carry, len(carry), None, 0, C.byref(n), C.byref(used)), "gf_bgzf_inflate")
            return int(n.value), int(used.value)
        out = np.zeros(1, dtype=np.uint8)
        rc = self._L.gf_bgzf_inflate(self._h, bgzf, len(bgzf), carry, len(carry), B._p(out), 0, C.byref(n), C.byref(used))
        if rc == B.GF_E_NOSPACE:
            out = np.zeros(n.value, dtype=np.uint8)
            rc = self._L.gf_bgzf_inflate(self._h, bgzf, len(bgzf), carry, len(carry), B._p(out), len(out), C.byref(n), C.byref(used))
        self._chk(rc, "gf_bgzf_inflate")
        return out[:n.value], int(used.value)

    def bam_fetch(self, begin, end):
        """Slices [begin[i], end[i]) of the inflated stream on the GPU -> one uint8 array, back to back."""
        begin = np.ascontiguousarray(begin, dtype=np.uint64)
        end = np.ascontiguousarray(end, dtype=np.uint64)
        total = int((end - begin).sum()) if len(begin) else 0
        dst = np.zeros(max(1, total), dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self._L.gf_bam_fetch(self._h, B._p(begin), B._p(end), len(begin), B._p(dst), total, C.byref(n)), "gf_bam_fetch")
        return dst[:n.value]

    def bam_pack(self, stream, first, ref_map, n_bytes=None):
        """Inflated BAM bytes (None: the n_bytes left on the GPU by bgzf_inflate) -> (records B.ALNREC [n], rec_begin uint64 [n],
        n_consumed); first = offset of the first alignment record, ref_map[refID] = .fai index or 0xFFFFFFFF."""
        if stream is not None:
            stream = np.ascontiguousarray(np.frombuffer(bytes(stream), dtype=np.uint8))
            n_bytes = len(stream)
        ref_map = np.ascontiguousarray(ref_map, dtype=np.uint32)
        cap = (n_bytes - first) // 128 + 16
        n, used = C.c_size_t(0), C.c_size_t(0)
        while True:
            recs = np.zeros(cap, dtype=B.ALNREC)
            rb = np.zeros(cap, dtype=np.uint64)
            rc = self._L.gf_bam_pack(self._h, B._p(stream) if stream is not None else None, n_bytes, first, B._p(ref_map), len(ref_map),
                                     B._p(recs), cap, B._p(rb), C.byref(n), C.byref(used))
            if rc != B.GF_E_NOSPACE:
                break
            cap = n.value   # the stream stays on the device: the retry does not upload it again
            stream = None
        self._chk(rc, "gf_bam_pack")
        return recs[:n.value], rb[:n.value], int(used.value)

    def _grow(self, call, dtype, cap):
        while True:
            out = np.zeros(max(cap, 1), dtype=dtype)
            n = C.c_size_t(0)
            rc = call(out, len(out), n)
            if rc == B.GF_E_NOSPACE:
                cap = n.value
                continue
            return rc, out[:n.value]

    def screen_reads(self, packed, read_len, k, min_hits=1, n_mask=None, cap=None):
        packed = np.ascontiguousarray(packed, dtype=np.uint8)
        n = packed.shape[0] if packed.ndim == 2 else packed.size // B.lib().gf_packed_read_bytes(read_len)
        if n_mask is not None:
            n_mask = np.ascontiguousarray(n_mask, dtype=np.uint32)
        rc, out = self._grow(lambda o, c, cnt: self._L.gf_screen_reads(self._h, B._p(packed), B._p(n_mask), n, read_len, k,
                                                                        min_hits, B._p(o), c, C.byref(cnt)),
                             B.HIT, cap if cap is not None else max(1024, n // 16))
        self._chk(rc, "gf_screen_reads")
        return out

    def tag_alignments(self, recs, insert_size, sd, clip_dist=250, anchor_mapq=30, cap=None):
        recs = np.ascontiguousarray(recs, dtype=B.ALNREC)
        rc, out = self._grow(lambda o, c, cnt: self._L.gf_tag_alignments(self._h, B._p(recs), len(recs), insert_size, sd,
                                                                          clip_dist, anchor_mapq, B._p(o), c, C.byref(cnt)),
                             B.TAGHIT, cap if cap is not None else max(1024, len(recs) // 8))
        self._chk(rc, "gf_tag_alignments")
        return out

    def tag_alignments_bam(self, n_recs, insert_size, sd, clip_dist=250, anchor_mapq=30, cap=None):
        """tag_alignments on the records the last bam_pack left on the GPU (n_recs of them: sizes the first output buffer)."""
        rc, out = self._grow(lambda o, c, cnt: self._L.gf_tag_alignments_bam(self._h, insert_size, sd, clip_dist, anchor_mapq,
                                                                              B._p(o), c, C.byref(cnt)),
                             B.TAGHIT, cap if cap is not None else max(1024, n_recs // 8))
        self._chk(rc, "gf_tag_alignments_bam")
        return out

    def tag_low_mapq_bam(self, n_recs, table, cap=None):
        table = np.ascontiguousarray(table, dtype=B.DPOS)
        rc, out = self._grow(lambda o, c, cnt: self._L.gf_tag_low_mapq_bam(self._h, B._p(table), len(table), B._p(o), c, C.byref(cnt)),
                             B.TAGHIT, cap if cap is not None else max(1024, n_recs // 8))
        self._chk(rc, "gf_tag_low_mapq_bam")
        return out

    def tag_low_mapq(self, recs, table, cap=None):
        recs = np.ascontiguousarray(recs, dtype=B.ALNREC)
        table = np.ascontiguousarray(table, dtype=B.DPOS)
        rc, out = self._grow(lambda o, c, cnt: self._L.gf_tag_low_mapq(self._h, B._p(recs), len(recs), B._p(table), len(table),
                                                                        B._p(o), c, C.byref(cnt)),
                             B.TAGHIT, cap if cap is not None else max(1024, len(recs) // 8))
        self._chk(rc, "gf_tag_low_mapq")
        return out

    def assemble(self, pool_packed, pool_off, read_len, kk, min_count=2, min_contig=40, n_mask=None):
        """pool_packed: uint8 [total_reads, rb]; pool_off: n_pools+1 read offsets; kk: [(k, kv), ...].
        Returns (contigs structured array (B.CONTIG), sequence bytes)."""
        pool_packed = np.ascontiguousarray(pool_packed, dtype=np.uint8)
        pool_off = np.ascontiguousarray(pool_off, dtype=np.uint64)
        ks = np.asarray([a for a, _ in kk], dtype=np.int32)
        kvs = np.asarray([b for _, b in kk], dtype=np.int32)
        if n_mask is not None:
            n_mask = np.ascontiguousarray(n_mask, dtype=np.uint32)
            if not n_mask.any():      # no N, no ragged read: the launches specialised for mask-free pools (k as a compile-time constant) apply
                n_mask = None
        ccap, scap = 1024, 1 << 16
        while True:
            ctg = np.zeros(ccap, dtype=B.CONTIG)
            seq = np.zeros(scap, dtype=np.uint8)
            nc, sl = C.c_size_t(0), C.c_size_t(0)
            rc = self._L.gf_assemble(self._h, B._p(pool_packed), B._p(n_mask), B._p(pool_off), len(pool_off) - 1, read_len,
                                     B._p(ks), B._p(kvs), len(kk), min_count, min_contig, B._p(ctg), ccap, C.byref(nc),
                                     B._p(seq), scap, C.byref(sl))
            if rc == B.GF_E_NOSPACE:
                ccap, scap = max(ccap, nc.value), max(scap, sl.value)
                continue
            self._chk(rc, "gf_assemble")
            return ctg[:nc.value], seq[:sl.value].tobytes()

    def count_kmers(self, pool_packed, read_len, k, min_count=2, n_mask=None):
        """(kmers uint64 [n, 2] (hi, lo), counts uint32 [n]) ascending — the `kmc -k{k}` | `kmc_dump` listing."""
        pool_packed = np.ascontiguousarray(pool_packed, dtype=np.uint8)
        n = pool_packed.shape[0]
        if n_mask is not None:
            n_mask = np.ascontiguousarray(n_mask, dtype=np.uint32)
        cap = 4096
        while True:
            km = np.zeros((cap, 2), dtype=np.uint64)
            cn = np.zeros(cap, dtype=np.uint32)
            m = C.c_size_t(0)
            rc = self._L.gf_count_kmers(self._h, B._p(pool_packed), B._p(n_mask), n, read_len, k, min_count, B._p(km), B._p(cn),
                                        cap, C.byref(m))
            if rc == B.GF_E_NOSPACE:
                cap = m.value
                continue
            self._chk(rc, "gf_count_kmers")
            return km[:m.value], cn[:m.value]

    def quick_check(self, contig_sets, k=10):
        """The contig merger's all-pairs k-mer prefilter (ContigsCompactor.cpp:1982-2095) for a batch of contig sets (one per
        gap): contig_sets = [[seq, ...], ...].  Returns B.QCPAIR triples (set, i, j) sorted, nodes = 2 * contig + strand."""
        flat = [c for cs in contig_sets for c in cs]
        blob = "".join(flat).encode()
        coff = np.zeros(len(flat) + 1, dtype=np.uint64)
        coff[1:] = np.cumsum([len(c) for c in flat])
        soff = np.zeros(len(contig_sets) + 1, dtype=np.uint64)
        soff[1:] = np.cumsum([len(cs) for cs in contig_sets])
        cap = 4096
        while True:
            out = np.zeros(cap, dtype=B.QCPAIR)
            n = C.c_size_t(0)
            rc = self._L.gf_quick_check(self._h, blob, B._p(coff), B._p(soff), len(contig_sets), k, B._p(out), cap, C.byref(n))
            if rc == B.GF_E_NOSPACE:
                cap = n.value
                continue
            self._chk(rc, "gf_quick_check")
            return out[:n.value]

    # ContigsMerger's options as GAPPadder sets them (MergeContigs.py:75: -s 0.4 -i1 -2.0 -i2 -2.0 -x 12 -y 50) + its defaults (main.cpp:24-27)
    MERGER_PARAMS = (-2.0, -2.0, 50.0, 0.005, 0.4, 12.0, 6.0)

    def overlap_evaluate(self, contig_sets, pairs, params=None, relax=False):
        """The contig merger's pairwise overlap evaluation (ContigsCompactor::Evaluate, ContigsCompactor.cpp:1572-1976) of the
        ordered node pairs `pairs` (B.QCPAIR triples: set, i, j; nodes = 2 * contig + strand) -> B.OVL_RESULT array, same order."""
        pairs = np.ascontiguousarray(pairs, dtype=B.QCPAIR)
        out = np.zeros(len(pairs), dtype=B.OVL_RESULT)
        if not len(pairs):
            return out
        flat = [c for cs in contig_sets for c in cs]
        blob = "".join(flat).encode()
        coff = np.zeros(len(flat) + 1, dtype=np.uint64)
        coff[1:] = np.cumsum([len(c) for c in flat])
        soff = np.zeros(len(contig_sets) + 1, dtype=np.uint64)
        soff[1:] = np.cumsum([len(cs) for cs in contig_sets])
        pr = np.zeros(1, dtype=B.OVL_PARAMS)
        p7 = tuple(self.MERGER_PARAMS if params is None else params)
        pr[0] = p7[:7] + (float(p7[7]) if len(p7) > 7 else (1.0 if relax else 0.0),)
        self._chk(self._L.gf_overlap_evaluate(self._h, blob, B._p(coff), B._p(soff), len(contig_sets), B._p(pairs), len(pairs), B._p(pr), B._p(out)),
                  "gf_overlap_evaluate")
        return out

    def merge_round(self, contig_sets, params=None, kmer_len_quick=10, max_set=128, open_gaps=None, k_pairs=None):
        """The contig-merge round of the step (gf_merge_open_gaps_dev) on contig sets given from the host: contig_sets[g] = the contigs
        of gap g; every gap counts as open unless open_gaps (booleans) says otherwise.  Returns (per gap the merged sequences in record
        order, stats dict).  k_pairs = [(k, kv)]: the contigs are (k, kv, sequence) triples in ANY record order and the round takes them in
        the order of the gap's contigs.fa (pairs in list order, inside a pair by length descending, then sequence); without it: the order
        given.  A test / tool entry: the pipeline calls the device function on the step's own contig list."""
        n_gaps = len(contig_sets)
        kinfo = None
        if k_pairs is not None:
            kinfo = [(int(k), int(kv)) for g, cs in enumerate(contig_sets) for (k, kv, _) in cs]
            contig_sets = [[c for (_, _, c) in cs] for cs in contig_sets]
        flat = [(g, c) for g, cs in enumerate(contig_sets) for c in cs]
        n0 = len(flat)
        cap = n0 + 4096 + 4 * n0
        ctg = np.zeros(cap, dtype=B.CONTIG)
        off = 0
        for i, (g, c) in enumerate(flat):
            ctg[i] = (g, kinfo[i][0] if kinfo else 31, kinfo[i][1] if kinfo else 29, max(1, len(c) - 28), len(c), 0, 0, off)
            off += len(c)
        blob = np.frombuffer("".join(c for _, c in flat).encode(), dtype=np.uint8)
        seq_cap = off + (1 << 20) + 8 * off
        best = np.zeros(max(1, n_gaps), dtype=np.uint64)
        if open_gaps is not None:
            best[:n_gaps] = [0 if o else 1 for o in open_gaps]
        cnt = np.zeros(4, dtype=np.uint32)
        cnt[0] = n0
        cnt[2:4] = np.array([off], dtype=np.uint64).view(np.uint32)
        pr = np.zeros(1, dtype=B.OVL_PARAMS)
        p7 = tuple(self.MERGER_PARAMS if params is None else params)
        pr[0] = p7[:7] + (0.0,)
        bufs = []

        def dev(nbytes, src=None):
            p_ = C.c_void_p()
            self._chk(self._L.gf_dev_alloc(self._h, max(64, int(nbytes)), C.byref(p_)), "gf_dev_alloc")
            bufs.append(p_)
            if src is not None and src.nbytes:
                self._chk(self._L.gf_memcpy_h2d(self._h, p_, B._p(src), src.nbytes), "gf_memcpy_h2d")
            return p_
        try:
            d_ctg, d_seq = dev(cap * 32, ctg[:max(1, n0)]), dev(seq_cap, blob)
            d_cnt, d_best, d_stats = dev(16, cnt), dev(best.nbytes, best), dev(4 * B.MG_WORDS)
            self._chk(self._L.gf_merge_open_gaps_dev(self._h, d_ctg, d_cnt, cap, d_seq, C.c_void_p(d_cnt.value + 8), seq_cap, d_best, n_gaps, B._p(pr),
                                                     kmer_len_quick, max_set,
                                                     (C.c_int * len(k_pairs))(*[int(k) for k, _ in k_pairs]) if k_pairs else None,
                                                     (C.c_int * len(k_pairs))(*[int(kv) for _, kv in k_pairs]) if k_pairs else None,
                                                     len(k_pairs) if k_pairs else 0, d_stats), "gf_merge_open_gaps_dev")
            self.sync()
            st = np.zeros(B.MG_WORDS, dtype=np.uint32)
            self._chk(self._L.gf_memcpy_d2h(self._h, B._p(st), d_stats, st.nbytes), "gf_memcpy_d2h")
            self._chk(self._L.gf_memcpy_d2h(self._h, B._p(cnt), d_cnt, cnt.nbytes), "gf_memcpy_d2h")
            n1, sl = int(cnt[0]), int(cnt[2:4].view(np.uint64)[0])
            stats = {"gaps_tried": int(st[B.MG_N_SETS]), "gaps_skipped_large": int(st[B.MG_SKIPPED]), "gaps_skipped_graph": int(st[B.MG_SKIPPED_GRAPH]), "pairs": int(st[B.MG_N_PAIRS]), "edges": int(st[B.MG_N_EDGES]),
                     "new_contigs": int(st[B.MG_N_JOBS]), "gaps_with_new_contigs": int(st[B.MG_SETS_WITH_JOBS]), "error_bits": int(st[B.MG_ERR]),
                     "prefilter_flags": int(st[B.MG_QC_FLAGS]), "contigs_before": int(st[B.MG_N0])}
            if st[B.MG_ERR] or n1 > cap or sl > seq_cap:
                raise RuntimeError("merge round: capacity flags %#x (contigs %d / %d, bases %d / %d)" % (int(st[B.MG_ERR]), n1, cap, sl, seq_cap))
            out = [[] for _ in range(n_gaps)]
            if n1 > n0:
                new = np.zeros(n1 - n0, dtype=B.CONTIG)
                self._chk(self._L.gf_memcpy_d2h(self._h, B._p(new), C.c_void_p(d_ctg.value + 32 * n0), new.nbytes), "gf_memcpy_d2h")
                sq = np.zeros(max(1, sl), dtype=np.uint8)
                self._chk(self._L.gf_memcpy_d2h(self._h, B._p(sq), d_seq, sl), "gf_memcpy_d2h")
                sb = sq.tobytes()
                for c in new:
                    assert int(c["k"]) == 0 and int(c["kv"]) == 0
                    out[int(c["gap"])].append(sb[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode())
            return out, stats
        finally:
            for p_ in bufs:
                self._L.gf_dev_free(self._h, p_)

    # ---- timing -------------------------------------------------------------------------------------
    def timing(self, on=True):
        self._chk(self._L.gf_timing_enable(self._h, 1 if on else 0), "gf_timing_enable")
        self._chk(self._L.gf_timing_reset(self._h), "gf_timing_reset")

    def kernel_time(self, which):
        ms, n = C.c_double(0), C.c_uint64(0)
        self._chk(self._L.gf_timing_read(self._h, which, C.byref(ms), C.byref(n)), "gf_timing_read")
        return ms.value, n.value

    def sync(self):
        self._chk(self._L.gf_sync(self._h), "gf_sync")

    # ---- synthetic workload (include/gf_synth.h) ----------------------------------------------------
    @staticmethod
    def synth_cfg(seed=20260002, scaffold_len=5_000_000, n_scaffolds=50, gaps_per_scaffold=20, gap_len=2000,
                  read_len=150, insert_mean=300, insert_sd=30, err=0.005, mapq0=0.02, chimeric=0.01, flank_len=300, library=0,
                  repeat_period=0, repeat_copies=50):
        """repeat_period P >= 4: the stress workload of include/gf_synth.h — every P-th gap sits at a copy of a repeat family shared
        by repeat_copies gaps, two more of every P share a 2-copy repeat, one carries a low-complexity run in its flank."""
        c = np.zeros(1, dtype=B.SYNTH_CFG)
        c[0] = (seed, scaffold_len, n_scaffolds, gaps_per_scaffold, gap_len, read_len, insert_mean, insert_sd,
                int(round(err * 65536)), int(round(mapq0 * 65536)), int(round(chimeric * 65536)), flank_len, library,
                (int(repeat_period) & 0xFF) | ((int(repeat_copies) & 0xFF) << 8) if repeat_period else 0)
        return c

    @staticmethod
    def synth_layout(cfg):
        n = int(cfg["n_scaffolds"][0]) * int(cfg["gaps_per_scaffold"][0])
        fl = int(cfg["flank_len"][0]) - 5
        gaps = np.zeros(n, dtype=B.GAP)
        blob = np.zeros(2 * n * fl, dtype=np.uint8)
        off = np.zeros(2 * n + 1, dtype=np.uint64)
        rc = B.lib().gf_synth_layout(B._p(cfg), B._p(gaps), B._p(blob), B._p(off))
        if rc:
            raise B.GapFillError(rc, "gf_synth_layout")
        b = blob.tobytes().decode()
        flanks = [(b[int(off[2 * g]):int(off[2 * g + 1])], b[int(off[2 * g + 1]):int(off[2 * g + 2])]) for g in range(n)]
        return gaps, flanks

    @staticmethod
    def synth_truth(cfg, scaffold, start, n):
        """The true bases of [start, start + n) of a scaffold of the synthetic draft (the planted gaps' interiors included)."""
        out = np.zeros(int(n), dtype=np.uint8)
        rc = B.lib().gf_synth_truth(B._p(cfg), int(scaffold), int(start), int(n), B._p(out))
        if rc:
            raise B.GapFillError(rc, "gf_synth_truth")
        return out.tobytes().decode()

    def synth_pairs_dev(self, cfg, first_pair, n_pairs, d_packed, d_recs=None):
        self._chk(self._L.gf_synth_pairs_dev(self._h, B._p(cfg), int(first_pair), int(n_pairs), d_packed, d_recs),
                  "gf_synth_pairs_dev")
