"""`-c Collect` / `-c All` of the CLI on libraries that stay in HBM (software_path.samtools = "builtin").

The reference's Collect stage (main.py:226-270) is, per library: one `samtools view | collect_reads_for_gaps.py` pipe per scaffold,
the discordant inversion + sort, a second pipe per scaffold for the MAPQ-0 hop, then FOUR single-threaded CPython passes over
the whole FASTQ pair that look every record's id up in a dictionary (run_multi_threads_discordant.py:141-317, 452-594), then a
`cat` per gap over the libraries (merge_reads.py:12-56).  Here a library is read ONCE:

    BAM    file bytes -> HBM -> gf_bgzf_inflate -> gf_bam_append_dev      32-byte records + QNAME hashes + QNAME arena, resident
    FASTQ  file bytes -> HBM -> gf_fastq_pack_dev + gf_fastq_index_dev    2-bit reads (+ N masks), id hashes, record offsets
    join   gf_read_join_dev                                                record.read = 2 * FASTQ record number + mate

and gappadder_amd/pipeline.py runs the same chain of launches bench.py times: tagger + second hop (+ the flank-k-mer screen when
`parameters.kmer_screen` is set) -> per-gap pools -> library merge -> assembly of every (k, k_velvet) -> flank anchoring.
The reference's files are then WRITTEN FROM the results — they are no longer the data path:
  * `{lib}/scaffold_reads_list_all/*`, `discordant_reads_pos.txt(.sorted.txt)`, `discordant_temp/*`, `discordant_reads_list/*`,
    `left_reads.list`, `right_reads.list`: from the hits (a few per cent of the records); only THEIR names cross PCIe;
  * `{lib}/gap_reads/{id}.fastq`, `gap_reads_high_quality/{id}.fastq`, `merged/*/{id}.fastq`: the FASTQ records of the pooled
    read ids (one D2H of ids + offsets), cut from the input files at the offsets the ingest kernels recorded;
  * `-c All`: `merged/velvet_temp/{id}/contigs*.fa` of the first assembly round from the device contigs (assemble_gaps.py).
The join is by 64-bit hash of the names; the names of every record that produces a hit are compared exactly against the FASTQ ids
they were joined to, and a mismatch raises.  Inputs this path does not take (duplicate FASTQ ids, mate files that are not in the
same order, reads longer than 1 000 bases) raise `DeviceCollectUnsupported`: main.py then runs the per-scaffold path."""
import ctypes as C
import mmap
import os
import sys
import time

import numpy as np
import torch

from . import _lib as B
from . import textio
from . import bam_io
from . import sam_io
from .pipeline import CNT_HOP, CNT_ROWS, CNT_SCREEN, CNT_TAG, DeviceLibrary, Pipeline

NO_READ = 0xFFFFFFFF


class DeviceCollectUnsupported(Exception):
    pass


class _Restart(Exception):
    def __init__(self, read_len):
        self.read_len = read_len


def _guess_read_len(paths, n_records=None):
    """Longest sequence line among the first records of every file (GF_INGEST_GUESS_RECORDS, default 4 096): the packed row size; a longer
    read further down restarts the ingest at its length."""
    n_records = int(os.environ.get("GF_INGEST_GUESS_RECORDS", 4096)) if n_records is None else n_records
    L = 0
    for p in paths:
        with open(p, "rb") as f:
            for i, line in enumerate(f):
                if i >= 4 * n_records:
                    break
                if i % 4 == 1:
                    L = max(L, len(line.rstrip(b"\r\n")))
    return L


class _RecordFiles:
    """The FASTQ files of a library for gf_fastq_records_text: as open files (one positioned read per record) when the records asked for
    are a small part of the files, as mappings when they cover much of them — measured on a 7.9-GB file pair with 0.5 % of the bytes
    asked for, mapping + unmapping cost three times the reads; on a 0.8-GB pair with 5 % asked for, half of them."""

    def __init__(self, paths):
        self.files = [open(p, "rb", buffering=0) for p in paths]
        self.size = sum(os.fstat(f.fileno()).st_size for f in self.files)
        self.maps = None

    def pick(self, begin, end):
        asked = int((np.asarray(end, dtype=np.int64) - np.asarray(begin, dtype=np.int64)).sum())
        how = os.environ.get("GF_RECORD_READS", "")        # "pread" / "mmap": tests run both
        if how == "pread" or (how != "mmap" and asked * 64 < self.size):
            return self.files
        if self.maps is None:
            self.maps = [mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) for f in self.files]
        return self.maps

    def close(self):
        for m in self.maps or []:
            m.close()
        for f in self.files:
            f.close()


class ResidentLibrary(DeviceLibrary):
    """A DeviceLibrary read from files, with what the file contract needs on top: QNAME arena, FASTQ record offsets, the files."""
    pass


class DeviceCollector:
    def __init__(self, gf, cfg, sf_fai, sf_gap_pos, anchor_mapq=30, clip_dist=250, kmers=None, chunk_bytes=256 << 20, log=None):
        """cfg: main.parse_configuration's dictionary.  kmers: the (k, k_velvet) pairs to assemble right away (`-c All`), or None
        (`-c Collect`: pools and files only)."""
        self.gf, self.lib, self.h = gf, B.lib(), gf.handle
        self.cfg, self.sf_fai, self.sf_gap_pos = cfg, sf_fai, sf_gap_pos
        self.anchor_mapq, self.clip_dist = anchor_mapq, clip_dist
        self.kmers = list(kmers) if kmers else []
        self.chunk_bytes = int(os.environ.get("GF_INGEST_CHUNK_BYTES", chunk_bytes))
        self.dev = torch.device("cuda", torch.cuda.current_device())
        self.t = {}
        self.log = log or (lambda s: None)

    def _chk(self, rc, what):
        if rc:
            raise B.GapFillError(rc, what, self.lib.gf_last_error(self.h).decode())

    def _tick(self, name, t0):
        self.gf.sync()
        torch.cuda.synchronize()
        self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0

    # ---- FASTQ pair -> packed reads [n_pairs][2][rb] (+ masks), id hashes, record offsets -----------------------------------
    def _ingest_fastq(self, path, L):
        """One mate file -> (records, packed reads, N masks, id hashes, header offsets (+ the file size)), all on the device.  The file
        is read in pieces straight into a pinned buffer and copied to HBM as it is; the pack kernel itself finds where the last whole
        4-line record of a piece ends (hdr_begin[n]), the few bytes behind it are carried in front of the next piece — no host pass
        over the text."""
        lib, h, dev, gf = self.lib, self.h, self.dev, self.gf
        rb, nmw = lib.gf_packed_read_bytes(L), (L + 31) // 32
        packed, masks, hashes, hdrs = [], [], [], []
        d_cnt = torch.zeros(4, dtype=torch.int64, device=dev)        # [0] n_reads (u64), [1] status (u32), [2] max_len (u32)
        size = os.path.getsize(path)
        chunk = max(4096, min(self.chunk_bytes, size))
        slack = 1 << 16                                             # room for the carried tail: a record longer than this is no short read
        pin = self._pinned(chunk + slack)
        view = memoryview(pin.numpy())
        d_text = torch.empty(chunk + slack + 64, dtype=torch.uint8, device=dev)
        n_total, off, carry = 0, 0, 0
        with open(path, "rb", buffering=0) as f:
            file_at = 0
            while True:
                got = self._read_piece(f.fileno(), view[carry:carry + min(chunk, size - file_at)], file_at)
                file_at += got
                total = carry + (got or 0)
                if total == 0:
                    break
                last = not got or off + total >= size
                d_text[:total].copy_(pin[:total], non_blocking=True)
                cap = total // max(8, L) + 16
                while True:
                    d_p = torch.empty(cap * rb, dtype=torch.uint8, device=dev)
                    d_m = torch.empty(cap * nmw, dtype=torch.int32, device=dev)
                    d_h = torch.empty(cap + 1, dtype=torch.int64, device=dev)
                    d_cnt.zero_()
                    torch.cuda.synchronize()
                    self._chk(lib.gf_fastq_pack_dev(h, d_text.data_ptr(), total, L, d_p.data_ptr(), cap, d_m.data_ptr(), d_h.data_ptr(),
                                                    d_cnt.data_ptr(), d_cnt.data_ptr() + 8), "gf_fastq_pack_dev")
                    gf.sync()
                    n, st = int(d_cnt[0]), int(d_cnt[1])
                    if n <= cap and not (st & 4):
                        break
                    cap = n + 16                                    # reads shorter than promised: more records than estimated
                if not last and (st & 8):
                    n -= 1                                          # the piece ends inside the quality line of its last record
                if not last:
                    if n == 0:
                        raise DeviceCollectUnsupported("a FASTQ record of %s is longer than %d bytes" % (path, chunk))
                    end = int(d_h[n])                               # where the last whole record of the piece ends
                else:
                    end = total
                d_id = torch.empty(max(1, n), dtype=torch.int64, device=dev)
                self._chk(lib.gf_fastq_index_dev(h, d_text.data_ptr(), end, d_h.data_ptr(), n, d_id.data_ptr(), d_cnt.data_ptr() + 16),
                          "gf_fastq_index_dev")
                gf.sync()
                mx = int(d_cnt[2]) & 0xFFFFFFFF
                if mx > L:
                    raise _Restart(mx)
                packed.append(d_p[:n * rb])
                masks.append(d_m[:n * nmw])
                hashes.append(d_id[:n])
                hdrs.append(d_h[:n] + off)
                n_total += n
                carry = total - end
                if carry > slack:
                    raise DeviceCollectUnsupported("a FASTQ record of %s is longer than %d bytes" % (path, slack))
                if carry:
                    view[:carry] = bytes(view[end:total])
                off += end
                if last:
                    break
        hdr = torch.cat(hdrs + [torch.tensor([size], dtype=torch.int64, device=dev)]) if hdrs else torch.tensor([size], dtype=torch.int64, device=dev)
        cat = lambda xs, dt: torch.cat(xs) if xs else torch.empty(0, dtype=dt, device=dev)
        return n_total, cat(packed, torch.uint8), cat(masks, torch.int32), cat(hashes, torch.int64), hdr

    def _read_piece(self, fd, dst, file_at):
        """The next len(dst) bytes of the file into the pinned buffer, read by a few threads at once (preadv releases the interpreter
        lock; one thread copies out of the page cache at a fraction of what the memory system and the PCIe link behind it take)."""
        n = len(dst)
        n_thr = int(os.environ.get("GF_INGEST_READ_THREADS", "4"))
        if n_thr <= 1 or n < (8 << 20):
            parts = [(0, n)]
        else:
            step = -(-n // n_thr) + 4095 & ~4095
            parts = [(a, min(n, a + step)) for a in range(0, n, step)]

        def read(a, z):
            at = a
            while at < z:
                r = os.preadv(fd, [dst[at:z]], file_at + at)
                if r <= 0:
                    break
                at += r
            return at - a
        if len(parts) == 1:
            return read(0, n)
        if getattr(self, "_readers", None) is None:
            from concurrent.futures import ThreadPoolExecutor
            self._readers = ThreadPoolExecutor(n_thr)
        got = list(self._readers.map(lambda p: read(*p), parts))
        for (a, z), g in zip(parts, got):       # a short part (the file shrank under us) ends the piece there
            if g < z - a:
                return a + g
        return n

    def _pinned(self, n):
        if getattr(self, "_pin", None) is None or self._pin.numel() < n:
            self._pin = torch.empty(n, dtype=torch.uint8, pin_memory=True)
        return self._pin

    def _ingest_pair(self, left, right, L):
        dev, rb, nmw = self.dev, self.lib.gf_packed_read_bytes(L), (L + 31) // 32
        nl, pl, ml, hl, hdr_l = self._ingest_fastq(left, L)
        nr, pr, mr, hr, hdr_r = self._ingest_fastq(right, L)
        if nl != nr:
            raise DeviceCollectUnsupported("%s holds %d records, %s %d" % (left, nl, right, nr))
        if nl == 0:
            raise DeviceCollectUnsupported("%s holds no reads" % left)
        if not bool((hl == hr).all()):
            raise DeviceCollectUnsupported("the records of %s and %s do not carry the same ids in the same order" % (left, right))
        d_reads = torch.empty(nl * 2 * rb + 64, dtype=torch.uint8, device=dev)
        v = d_reads[:nl * 2 * rb].view(nl, 2, rb)
        v[:, 0, :] = pl.view(nl, rb)
        v[:, 1, :] = pr.view(nl, rb)
        d_nmask = None
        if bool((ml != 0).any()) or bool((mr != 0).any()):      # N or reads shorter than L somewhere: the windows touching them are skipped
            d_nmask = torch.empty(nl * 2 * nmw + 16, dtype=torch.int32, device=dev)
            mv = d_nmask[:nl * 2 * nmw].view(nl, 2, nmw)
            mv[:, 0, :] = ml.view(nl, nmw)
            mv[:, 1, :] = mr.view(nl, nmw)
        torch.cuda.synchronize()
        return nl, d_reads, d_nmask, hl, hdr_l, hdr_r

    # ---- BAM -> resident records + QNAME hashes + QNAME arena --------------------------------------------------------------
    def _ingest_bam(self, path, fai_names, both_unmapped=False):
        """both_unmapped: the records with both mates unmapped (FLAG bits 4 and 8) are written to `{path}.both_unmapped.sam` / `.fq` on the
        way — the files collect_both_unmapped_reads.run_collect_both_unmapped makes from a second pass over the BAM; a run that goes on
        to the assembly rounds (`-c All`) then has them already."""
        lib, h, dev, gf = self.lib, self.h, self.dev, self.gf
        f_sam = f_fq = None
        if both_unmapped:
            try:
                f_sam = open(path + ".both_unmapped.sam", "wb")
                f_fq = open(path + ".both_unmapped.fq", "wb")
            except OSError:        # the BAM's folder cannot be written: the second round, if it comes to one, reports that itself
                if f_sam is not None:
                    f_sam.close()
                f_sam = f_fq = None
        self.both_unmapped_written = f_sam is not None
        try:
            return self._ingest_bam_pieces(path, fai_names, f_sam, f_fq)
        finally:
            for f in (f_sam, f_fq):
                if f is not None:
                    f.close()

    def _bam_pieces(self, path):
        """The file in pieces of chunk_bytes, each read by several threads into the pinned buffer behind what the caller left unconsumed of
        the piece before (the bytes of a BGZF block cut by the piece's end: the caller's `file_carry`, < 64 KiB)."""
        size = os.path.getsize(path)
        chunk = max(4096, min(self.chunk_bytes, size))
        slack = 1 << 17
        view = memoryview(self._pinned(chunk + slack).numpy())
        self._bam_left = b""
        with open(path, "rb", buffering=0) as f:
            at = 0
            while at < size:
                left = self._bam_left
                if len(left) > slack:
                    raise ValueError("BGZF block of more than %d bytes in %s" % (slack, path))
                view[slack - len(left):slack] = left
                got = self._read_piece(f.fileno(), view[slack:slack + min(chunk, size - at)], at)
                if got <= 0:
                    break
                at += got
                yield view[slack - len(left):slack + got]

    def _inflate(self, data, rec_carry):
        """gf_bgzf_inflate of a piece that lies in the pinned buffer (no copy into a bytes object on the way) -> (stream length, bytes consumed);
        what is not consumed is handed to the next piece."""
        n, used = C.c_size_t(0), C.c_size_t(0)
        arr = np.frombuffer(data, dtype=np.uint8)
        self._chk(self.lib.gf_bgzf_inflate(self.h, C.cast(arr.ctypes.data, C.c_char_p), len(arr), rec_carry, len(rec_carry), None, 0, C.byref(n), C.byref(used)),
                  "gf_bgzf_inflate")
        self._bam_left = bytes(data[used.value:])
        return int(n.value), int(used.value)

    def _ingest_bam_pieces(self, path, fai_names, f_sam, f_fq):
        lib, h, dev, gf = self.lib, self.h, self.dev, self.gf
        index = {n: i for i, n in enumerate(fai_names)}
        size = os.path.getsize(path)
        rec_cap, name_cap = max(1 << 16, size // 24), max(1 << 20, size // 2)
        d_recs = torch.empty(rec_cap * 4, dtype=torch.int64, device=dev)
        d_qh = torch.empty(rec_cap, dtype=torch.int64, device=dev)
        d_noff = torch.empty(rec_cap + 1, dtype=torch.int64, device=dev)
        d_names = torch.empty(name_cap, dtype=torch.uint8, device=dev)
        d_seen = torch.zeros(max(1, len(fai_names)), dtype=torch.int32, device=dev)
        n_recs = n_name = 0
        ref_names, ref_map = None, None
        file_carry, rec_carry = b"", b""
        nr, nb, used = C.c_size_t(0), C.c_size_t(0), C.c_size_t(0)
        torch.cuda.synchronize()
        for data in self._bam_pieces(path):     # (a view of the pinned buffer: what the last piece left over + the next bytes of the file)
            n_stream, consumed_file = self._inflate(data, rec_carry)
            file_carry = self._bam_left
            first = 0
            if ref_names is None:
                hdr, want = None, 1 << 16
                while hdr is None and n_stream:
                    got = min(want, n_stream)
                    hdr = bam_io.parse_header(gf.bam_fetch([0], [got]).tobytes())
                    if got == n_stream:
                        break
                    want *= 4
                if hdr is None:          # header longer than the pieces so far: keep everything inflated and go on
                    rec_carry = gf.bam_fetch([0], [n_stream]).tobytes() if n_stream else b""
                    continue
                ref_names, first = hdr
                ref_map = np.array([index.get(n, bam_io.NO_REF) for n in ref_names], dtype=np.uint32)
            d_rb = torch.empty(n_stream // 36 + 2, dtype=torch.int64, device=dev) if f_sam is not None else None   # (a record takes 36 bytes at least)
            while True:
                rc = lib.gf_bam_append_dev(h, n_stream, first, B._p(ref_map), len(ref_map), d_recs.data_ptr(), n_recs, rec_cap, d_qh.data_ptr(),
                                           d_names.data_ptr(), n_name, name_cap, d_noff.data_ptr(), d_seen.data_ptr(), len(fai_names),
                                           d_rb.data_ptr() if d_rb is not None else None, C.byref(nr), C.byref(nb), C.byref(used))
                if rc != B.GF_E_NOSPACE:
                    break
                if n_recs + nr.value > rec_cap:
                    rec_cap = int(1.5 * (n_recs + nr.value)) + 1024
                    d_recs = torch.cat([d_recs[:4 * n_recs], torch.empty(4 * (rec_cap - n_recs), dtype=torch.int64, device=dev)])
                    d_qh = torch.cat([d_qh[:n_recs], torch.empty(rec_cap - n_recs, dtype=torch.int64, device=dev)])
                    d_noff = torch.cat([d_noff[:n_recs + 1], torch.empty(rec_cap - n_recs, dtype=torch.int64, device=dev)])
                if n_name + nb.value > name_cap:
                    name_cap = int(1.5 * (n_name + nb.value)) + 1024
                    d_names = torch.cat([d_names[:n_name], torch.empty(name_cap - n_name, dtype=torch.uint8, device=dev)])
                torch.cuda.synchronize()
            self._chk(rc, "gf_bam_append_dev")
            if d_rb is not None and nr.value:       # `samtools view -f 12` of this piece, while its inflated bytes are still on the device
                flag = d_recs[4 * n_recs:4 * (n_recs + nr.value)].view(-1, 4)[:, 2] >> 32        # (gf_alnrec: FLAG at byte 20)
                sel = ((flag & 12) == 12).nonzero().flatten()
                if sel.numel():
                    rb = torch.cat([d_rb[:nr.value], torch.tensor([used.value], dtype=torch.int64, device=dev)])
                    begin, end = rb[sel].cpu().numpy().astype(np.uint64), rb[sel + 1].cpu().numpy().astype(np.uint64)
                    at = np.concatenate([[0], np.cumsum((end - begin).astype(np.int64))[:-1]]).astype(np.uint64)
                    sam, fq = textio.bam_records_text(gf.bam_fetch(begin, end), at, ref_names, h)
                    f_sam.write(sam)
                    f_fq.write(fq)
            del d_rb
            n_recs += nr.value
            n_name += nb.value
            rec_carry = gf.bam_fetch([used.value], [n_stream]).tobytes() if used.value < n_stream else b""
        if file_carry:
            raise ValueError("BAM file ends inside a BGZF block (%d stray bytes)" % len(file_carry))
        if rec_carry and ref_names is not None:
            raise ValueError("BAM file ends inside an alignment record (%d stray bytes)" % len(rec_carry))
        return n_recs, d_recs, d_qh, d_names, n_name, d_noff, d_seen

    def ingest_library(self, name, bam, left, right, is_mean, is_sd, fai_names, L, screen):
        t0 = time.perf_counter()
        n_pairs, d_reads, d_nmask, d_idh, hdr_l, hdr_r = self._ingest_pair(left, right, L)
        self._tick("ingest_fastq", t0)
        t0 = time.perf_counter()
        n_recs, d_recs, d_qh, d_names, n_name, d_noff, d_seen = self._ingest_bam(bam, fai_names, both_unmapped=bool(self.kmers))
        if self.kmers and self.both_unmapped_written:
            from . import collect_both_unmapped_reads
            collect_both_unmapped_reads.PREPARED.add(os.path.abspath(bam))
        self._tick("ingest_bam", t0)
        t0 = time.perf_counter()
        d_stats = torch.zeros(4, dtype=torch.int32, device=self.dev)
        torch.cuda.synchronize()
        self._chk(self.lib.gf_read_join_dev(self.h, d_idh.data_ptr(), n_pairs, d_recs.data_ptr(), d_qh.data_ptr(), n_recs, d_stats.data_ptr()),
                  "gf_read_join_dev")
        self.gf.sync()
        dup, orphan = int(d_stats[0]), int(d_stats[1])
        if dup:
            raise DeviceCollectUnsupported("%d read ids occur more than once in %s" % (dup, left))
        del d_qh, d_idh
        self._tick("join", t0)
        lb = ResidentLibrary(name, is_mean, is_sd, 2 * n_pairs, d_reads, d_recs, n_recs=n_recs, pull_mates=1, d_nmask=d_nmask, screen=screen)
        lb.bam, lb.left, lb.right = bam, left, right
        lb.d_names, lb.n_name, lb.d_noff, lb.seen = d_names, n_name, d_noff, d_seen.cpu().numpy()
        lb.hdr = (hdr_l, hdr_r)
        lb.records_without_a_read = orphan
        return lb

    # ---- the run --------------------------------------------------------------------------------------------------------
    def run(self, folders, merge_folder, write_files=True):
        """folders: the per-library working folders (main.prepare_folders); merge_folder: `{wf}merged/`.  Returns the Pipeline's
        Results of the step (contigs + picks when kmers were given), with `.keys` = the gap ids."""
        cfg, gf = self.cfg, self.gf
        names = sam_io.read_fai(self.sf_fai)
        sidx = {n: i for i, n in enumerate(names)}
        gaps, keys = sam_io.read_gap_positions(self.sf_gap_pos, sidx)
        if not len(gaps):
            raise DeviceCollectUnsupported("no gaps")
        k_screen = int(cfg.get("kmer_screen", 0))
        flanks = None
        if k_screen or self.kmers:
            from .kmer_recruit import flank_table
            flanks = flank_table(cfg["wf"], keys)
        gf.set_gaps(gaps, len(names), flanks)
        paths = [p for pair in cfg["raw_reads"] for p in pair]
        L = max(16, _guess_read_len(paths))
        self.check_footprint(L)
        while True:
            if L > 1000:
                raise DeviceCollectUnsupported("reads of %d bases" % L)
            try:
                libs = [self.ingest_library("%d_is%d" % (n + 1, is_), bam, left, right, is_, sd, names, L, bool(k_screen))
                        for n, ((bam, is_, sd), (left, right)) in enumerate(zip(cfg["alignments"], cfg["raw_reads"]))]
                break
            except _Restart as r:           # a read longer than the first ones promised: pack again at that length
                L = r.read_len
                libs = None
                torch.cuda.empty_cache()
        kk = self._usable_pairs(L)
        pipe = Pipeline(gf, len(gaps), L, kk, device=self.dev, anchor_mapq=self.anchor_mapq, clip_dist=self.clip_dist,
                        k_screen=k_screen or None, keep_read_ids=True)
        pipe.assemble_in_step = False
        for lb in libs:
            pipe.add_library(lb)
        t0 = time.perf_counter()
        pipe.prepare()                       # recruits every library once (hits, second-hop rows, pool keys) and sizes the buffers from that
        self._tick("recruit_and_sizing", t0)
        t0 = time.perf_counter()
        pipe.finish()                        # ... the pools from the keys it left (no second recruit: a one-shot run)
        self._tick("pools", t0)
        d_mask = None
        if kk:
            t0 = time.perf_counter()
            d_mask = self._pool_masks(pipe)
            pipe.assemble(d_mask.data_ptr() if d_mask is not None else None)
            self._tick("assemble_and_pick", t0)
        res = pipe.fetch()
        # the context goes on to the reference's later rounds (host entry points, pools that GROW: assemble_gaps.py:349-351): the bounds the
        # pipeline derived from ITS pools must not outlive it — a stale asm_max_pool_reads sent most second-round pools of a C3-sized run to
        # the eight workgroups of the deep-pool launch (3 x 27 ms instead of 3 ms)
        gf.set_option("asm_max_pool_reads", 0)
        gf.set_option("asm_big_pool_reads", 131072)
        res.keys, res.k_pairs, res.read_len = keys, kk, L
        self.pipe, self.libs = pipe, libs
        if write_files:
            t0 = time.perf_counter()
            self._write_files(pipe, libs, names, gaps, keys, folders, merge_folder)
            self.t["write_files"] = time.perf_counter() - t0
        return res

    def footprint_bytes(self, L):
        """Device bytes a run on this configuration's files needs, from the file sizes alone (an upper estimate, before anything is read):
        per BAM the buffers of _ingest_bam_pieces (32-byte record + name hash + name offset per record at rec_cap = size / 24, the name arena
        at size / 2) + the tagger's 8-byte key column; per FASTQ pair the packed reads, N masks, id hashes and record offsets; the
        recruit buffers of Pipeline.add_library (hit_cap = an eighth of the reads: 8 + 12 + 12 + 12 + 4 x 8 bytes each); the streaming
        buffers (text / compressed / inflated pieces).  Pools, contigs and the assembly workspace follow what is recruited (1-2 % of that)."""
        rb, nmw = (L + 3) // 4, (L + 31) // 32
        total = 6 * (self.chunk_bytes + (64 << 20))
        for (bam, _, _), (left, right) in zip(self.cfg["alignments"], self.cfg["raw_reads"]):
            bsz = os.path.getsize(bam)
            n_rec = max(1 << 16, bsz // 24)
            fq = os.path.getsize(left) + os.path.getsize(right)
            n_reads = fq // (2 * 16 + 6) if L <= 16 else fq // (2 * L + 6)        # a record: id line, L bases, '+', L qualities (ids of 1+ characters)
            total += n_rec * (32 + 8 + 8 + 8) + max(1 << 20, bsz // 2) + n_reads * (rb + 4 * nmw + 8 + 8)
            total += (max(n_rec, n_reads) // 8 + (1 << 20)) * (8 + 12 + 12 + 12 + 32)
        return int(1.1 * total)

    def check_footprint(self, L):
        """The device path keeps every library whole in HBM; inputs that cannot fit take the streaming per-scaffold path instead of
        dying in an allocation halfway through the files."""
        need = self.footprint_bytes(L)
        free, _ = torch.cuda.mem_get_info(self.dev)
        limit = int(os.environ.get("GF_DEVICE_COLLECT_MAX_BYTES", "0")) or free
        if need > limit:
            raise DeviceCollectUnsupported("the libraries need about %.1f GB of device memory, %.1f GB are free" % (need / 1e9, limit / 1e9))

    def _usable_pairs(self, L):
        from .assemble_gaps import velvet_kv
        out = []
        for k, kv in self.kmers:
            p = (int(k), velvet_kv(int(kv)))
            if 16 <= p[0] <= min(64, L) and 15 <= p[1] < p[0] and p not in out:
                out.append(p)
        return out

    def _pool_masks(self, pipe):
        """N masks of the pooled reads, in pool order (None when no library has any): per library a gather by the pooled read ids,
        over the libraries the same merge as the bases (gf_pools_merge_dev on rows of 4 * ceil(L / 32) bytes)."""
        if all(lb.d_nmask is None for lb in pipe.libs):
            return None
        lib, h, dev, n_gaps = self.lib, self.h, self.dev, pipe.n_gaps
        nmw = (pipe.L + 31) // 32
        n_lib = len(pipe.libs)
        d_lm = torch.zeros(n_lib * pipe.lib_cap * nmw, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        for l, lb in enumerate(pipe.libs):
            if lb.d_nmask is None:
                continue
            self._chk(lib.gf_gather_rows_dev(h, lb.d_nmask.data_ptr(), lb.n_reads, 4 * nmw, lb.d_ids.data_ptr(), lb.d_pool_off.data_ptr() + 8 * n_gaps,
                                             pipe.lib_cap, d_lm.data_ptr() + 4 * l * pipe.lib_cap * nmw), "gf_gather_rows_dev")
        if not pipe.need_merge:
            return d_lm
        d_mm = torch.zeros(pipe.merged_cap * nmw + 16, dtype=torch.int32, device=dev)
        d_off = torch.zeros(n_gaps + 1, dtype=torch.int64, device=dev)
        d_err = torch.zeros(4, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        # rows of 4 * nmw bytes = the packed rows of a "read" of 16 * nmw bases
        self._chk(lib.gf_pools_merge_dev(h, d_lm.data_ptr(), pipe.lib_cap, pipe.d_libcnt.data_ptr(), n_lib, 1, n_gaps, 16 * nmw, 0, 1, pipe.batch,
                                         d_mm.data_ptr(), pipe.merged_cap, d_off.data_ptr(), d_err.data_ptr()), "gf_pools_merge_dev")
        self.gf.sync()
        assert int(d_err[0]) == 0 and bool((d_off == pipe.d_moff).all())
        return d_mm

    # ---- the reference's files, written from the results -------------------------------------------------------------------
    def _fetch_names(self, lb, recs_idx):
        """QNAMEs of the records `recs_idx` (sorted unique int64) from the library's arena -> list of str."""
        if not len(recs_idx):
            return []
        idx = torch.from_numpy(recs_idx).to(self.dev)
        b = lb.d_noff[idx].cpu().numpy().astype(np.uint64)
        e = lb.d_noff[idx + 1].cpu().numpy().astype(np.uint64)
        total = int((e - b).sum())
        dst = np.zeros(max(1, total), dtype=np.uint8)
        n = C.c_size_t(0)
        self._chk(self.lib.gf_fetch_slices(self.h, lb.d_names.data_ptr(), lb.n_name, B._p(b), B._p(e), len(b), B._p(dst), total, C.byref(n)),
                  "gf_fetch_slices")
        blob = dst[:total].tobytes()
        ends = np.cumsum((e - b).astype(np.int64))
        return [blob[int(a):int(z)].decode() for a, z in zip(ends - (e - b).astype(np.int64), ends)]

    def _write_files(self, pipe, libs, names, gaps, keys, folders, merge_folder):
        from .run_multi_threads_discordant import DiscordantReadsCollector
        n_gaps = len(gaps)
        with_gaps = set(int(g) for g in gaps["scaffold"])
        merged = {"gap_reads": [[] for _ in range(n_gaps)], "gap_reads_high_quality": [[] for _ in range(n_gaps)]}
        for lb, folder in zip(libs, folders):
            c = lb.d_cnt.cpu().numpy()
            n_th, n_lh, n_rows = int(c[CNT_TAG]), int(c[CNT_HOP]), int(c[CNT_ROWS])
            th = np.frombuffer(lb.d_thits[:n_th * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
            lh = np.frombuffer(lb.d_lhits[:n_lh * 12].cpu().numpy().tobytes(), dtype=B.TAGHIT)
            rows = np.frombuffer(lb.d_rows[:n_rows * 16].cpu().numpy().tobytes(), dtype=B.DPOS)
            th = th[np.lexsort((th["kind"], th["gap"], th["rec"]))]                       # the host entry point's order: (rec, gap, kind)
            hit_recs = np.unique(np.concatenate([th["rec"], lh["rec"]]).astype(np.int64))
            recs = np.frombuffer(lb.d_recs.view(-1, 4)[torch.from_numpy(hit_recs).to(self.dev)].cpu().numpy().tobytes(), dtype=B.ALNREC) \
                if len(hit_recs) else np.zeros(0, dtype=B.ALNREC)
            qn = self._fetch_names(lb, hit_recs)
            at_th = np.searchsorted(hit_recs, th["rec"].astype(np.int64))       # row of `recs` / `qn` per tagger hit
            at_lh = np.searchsorted(hit_recs, lh["rec"].astype(np.int64))
            mm = _RecordFiles([lb.left, lb.right])
            self._verify_join(lb, recs, qn, mm)
            # -- scaffold lists (collect_reads_for_gaps.py:93-159): a scaffold with gaps gets its pair of files once a record of it was seen
            out = {names[s]: {"left": [], "right": []} for s in range(len(names)) if (lb.seen[s] & 1) and s in with_gaps}
            R, G = recs[at_th], gaps[th["gap"]]
            g_len = (G["end"].astype(np.int64) - G["start"].astype(np.int64)).tolist()
            for q, flag, ref, pos, mapq, mref, mpos, kind, to_mate, g_idx, gl in zip(
                    [qn[i] for i in at_th.tolist()], R["flag"].tolist(), R["ref"].tolist(), R["pos"].tolist(), R["mapq"].tolist(),
                    R["mate_ref"].tolist(), R["mate_pos"].tolist(), th["kind"].tolist(), th["to_mate"].tolist(), G["idx_in_scaffold"].tolist(), g_len):
                own = "left" if flag & 0x40 else "right"
                side = ("right" if own == "left" else "left") if to_mate else own
                if kind == B.KIND_DISCORDANT:
                    rnext = "*" if mref == NO_READ else "=" if mref == ref else names[mref]
                    line = "%s %d %d discordant %d %s %d %d" % (q, g_idx, mapq, pos, rnext, mpos, gl)
                else:
                    line = "%s %d %d %s" % (q, g_idx, mapq, B.KIND_NAMES[kind])
                out[names[ref]][side].append(line)
            open(folder + "cluster_by_gap_reads_left.list", "w").close()
            open(folder + "cluster_by_gap_reads_right.list", "w").close()
            for s, d in out.items():
                for side in ("left", "right"):
                    with open(folder + "scaffold_reads_list_all/%s_cluster_by_gap_reads_%s.list" % (s, side), "w") as f:
                        f.write("".join(l + "\n" for l in d[side]))
            # -- discordant inversion + sort + split: the reference's own text files, from the lists just written (host, hits only)
            drc = DiscordantReadsCollector(self.sf_fai, lb.bam, folder, self.cfg["nthreads"], self.gf, self.cfg["samtools"])
            drc.collect_discordant_regions_v2(folder + "discordant_reads_pos.txt")
            # -- second hop (collect_discordant_low_mapq_reads.py:52-79): a scaffold that has a discordant_temp list gets its pair of
            #    files once a MAPQ-0 record of it was seen; a record's lines in the order of the sorted table's rows
            open(folder + "cluster_by_discordant_reads_left.list", "w").close()
            open(folder + "cluster_by_discordant_reads_right.list", "w").close()
            hop = {s: {"left": [], "right": []} for s in range(len(names))
                   if (lb.seen[s] & 2) and os.path.exists(folder + "discordant_temp/" + names[s] + ".list")}
            if len(lh):
                rr = rows[lh["gap"]]
                order = np.lexsort((rr["src_gap"], rr["src_scaffold"], lh["rec"]))
                R = recs[at_lh[order]]
                for q, flag, mapq, ms, ss, sg in zip([qn[i] for i in at_lh[order].tolist()], R["flag"].tolist(), R["mapq"].tolist(),
                                                     rr["mate_scaffold"][order].tolist(), rr["src_scaffold"][order].tolist(), rr["src_gap"][order].tolist()):
                    hop[ms]["left" if flag & 0x40 else "right"].append("%s %d_%d %d" % (q, ss, sg, mapq))
            for s, d in hop.items():
                for side in ("left", "right"):
                    with open(folder + "discordant_reads_list/%s_cluster_by_discordant_reads_%s.list" % (names[s], side), "w") as f:
                        f.write("".join(l + "\n" for l in d[side]))
            # -- left_reads.list / right_reads.list (run_multi_threads_discordant.py:187-194, 262-268)
            screen_map = self._screen_hit_map(lb, keys, mm) if lb.screen else {}
            for side in ("left", "right"):
                m = drc._read_gap_map(names, side, False)
                for rid, ks in screen_map.items():        # the hit read and its mate: both files (kmer_recruit.py)
                    m.setdefault(rid, {}).update(ks)
                with open(folder + side + "_reads.list", "w") as f:
                    f.write("".join("%s %s\n" % (g, r) for r, gs in m.items() for g in gs))
            # -- per-gap FASTQ files from the pooled read ids (run_multi_threads_discordant.py:209-241, 283-316: left file's
            #    stream order, then the right file's = ascending (mate, record) — the order gf_build_pools_dev sorts a gap's keys in)
            off = lb.d_pool_off.cpu().numpy()
            ids = lb.d_ids[:int(off[-1])].cpu().numpy().astype(np.int64)
            self._write_pool_fastq(lb, folder + "gap_reads/", keys, off, ids, mm, merged["gap_reads"])
            # -- high-quality subset: the scaffold-list lines with MAPQ == 60, no second hop (run_multi_threads_discordant.py:476, 548)
            rd = recs["read"][at_th].astype(np.int64) if len(th) else np.zeros(0, dtype=np.int64)
            mq = recs["mapq"][at_th] if len(th) else np.zeros(0, dtype=np.uint8)
            ok = (mq == 60) & ((rd & 0xFFFFFFFF) != NO_READ)
            tgt = (rd[ok] & 0xFFFFFFFF) ^ th["to_mate"][ok].astype(np.int64)
            hq = np.unique(np.stack([th["gap"][ok].astype(np.int64), tgt & 1, tgt >> 1], axis=1), axis=0) if ok.any() else np.zeros((0, 3), dtype=np.int64)
            hq_off = np.searchsorted(hq[:, 0], np.arange(n_gaps + 1))
            self._write_pool_fastq(lb, folder + "gap_reads_high_quality/", keys, hq_off, hq[:, 2] * 2 + hq[:, 1], mm, merged["gap_reads_high_quality"])
            mm.close()
        for name, per_gap in merged.items():            # merge_reads.py:43-51: `cat` of the libraries' files in library order
            d = "%s%s/" % (merge_folder, name)
            os.makedirs(d, exist_ok=True)
            for g, parts in enumerate(per_gap):
                if parts:
                    with open(d + keys[g] + ".fastq", "wb") as f:
                        f.write(b"".join(parts))
        os.makedirs(merge_folder + "gap_reads_alignment", exist_ok=True)      # requested by main.py:265, filled by no stage

    @staticmethod
    def _record(mm, b, e, suffix):
        """One FASTQ record as the reference re-writes it (run_multi_threads_discordant.py:212-221): `@{id}{suffix}`, the sequence,
        a bare `+`, the qualities; (id, text).  The definition of what gf_fastq_records_text does for many records at once
        (tests/test_textio_host.py compares the two); the run itself goes through that call."""
        h, s, _, q = (mm[b:e].split(b"\n") + [b"", b"", b""])[:4]
        f = h.split()
        rid = f[0].split(b"/")[0][1:].rstrip() if f else b""
        return rid, b"@" + rid + suffix + b"\n" + s.rstrip() + b"\n+\n" + q.rstrip() + b"\n"

    def _offsets(self, lb, mate, pairs):
        idx = torch.from_numpy(np.ascontiguousarray(pairs)).to(self.dev)
        hdr = lb.hdr[mate]
        return hdr[idx].cpu().numpy(), hdr[idx + 1].cpu().numpy()

    def _verify_join(self, lb, recs, qn, mm_left):
        """Every record that produced a hit: its QNAME against the id of the FASTQ record it was joined to (exact)."""
        rd = recs["read"].astype(np.int64) & 0xFFFFFFFF
        have = np.nonzero(rd != NO_READ)[0]
        if not len(have):
            return
        b, e = self._offsets(lb, 0, rd[have] >> 1)
        _, _, ids, ids_end = textio.fastq_records_text(mm_left.pick(b, e), b, e, np.zeros(len(b), dtype=np.uint8), (b"", b""), want_ids=True, handle=self.h)
        want = [qn[i].encode() for i in have.tolist()]
        if ids.tobytes() == b"".join(want) and np.array_equal(ids_end.astype(np.int64), np.cumsum([len(w) for w in want])):
            return
        ids, at = ids.tobytes(), 0
        for i, z in zip(have.tolist(), ids_end.tolist()):
            if ids[at:z] != qn[i].encode():
                raise RuntimeError("read-name join: alignment record %r was joined to FASTQ record %r of %s (64-bit hash collision)"
                                   % (qn[i], ids[at:z].decode(), lb.left))
            at = z

    def _screen_hit_map(self, lb, keys, mm):
        """k-mer-screen recruits (parameters.kmer_screen) as {readId: {gapKey: 1}}, for left_reads.list / right_reads.list (kmer_recruit.py)."""
        c = lb.d_cnt.cpu().numpy()
        n = int(c[CNT_SCREEN])
        if not n:
            return {}
        hits = np.frombuffer(lb.d_hits[:n * 8].cpu().numpy().tobytes(), dtype=B.HIT)
        pair_of = hits["read"].astype(np.int64) >> 1
        pairs, inv = np.unique(pair_of, return_inverse=True)
        b, e = self._offsets(lb, 0, pairs)
        _, _, ids, ids_end = textio.fastq_records_text(mm.pick(b, e), b, e, np.zeros(len(b), dtype=np.uint8), (b"", b""), want_ids=True, handle=self.h)
        ids = ids.tobytes().decode()
        ends = ids_end.tolist()
        rids = [ids[a:z] for a, z in zip([0] + ends[:-1], ends)]
        out = {}
        for i, g in zip(inv.tolist(), hits["gap"].tolist()):
            out.setdefault(rids[i], {})[keys[g]] = 1
        return out

    def _write_pool_fastq(self, lb, d, keys, off, ids, mm, merged):
        os.makedirs(d, exist_ok=True)
        for fn in os.listdir(d):          # the reference wipes the folder with rsync --delete (:199)
            os.remove(os.path.join(d, fn))
        if not len(ids):
            return
        mate = ids & 1
        b = np.zeros(len(ids), dtype=np.int64)
        e = np.zeros(len(ids), dtype=np.int64)
        for m_ in (0, 1):
            sel = np.nonzero(mate == m_)[0]
            if len(sel):
                b[sel], e[sel] = self._offsets(lb, m_, ids[sel] >> 1)
        # every pooled record re-written in one host pass (gf_fastq_records_text); a gap's file is a slice of that text
        text, text_end = textio.fastq_records_text(mm.pick(b, e), b, e, mate.astype(np.uint8), (b"_1", b"_2"), handle=self.h)
        at = np.concatenate([[0], text_end.astype(np.int64)])
        for g in range(len(keys)):
            a, z = int(off[g]), int(off[g + 1])
            if z > a:
                txt = text[int(at[a]):int(at[z])].tobytes()
                with open(d + keys[g] + ".fastq", "wb") as f:
                    f.write(txt)
                merged[g].append(txt)
