"""Assembly stage, first round (mirrors assemble_gaps.py:82-136, 244-299): for every gap that has a read pool and every
(k, k_velvet) pair, count canonical k-mers (KMC's role) and assemble the surviving k-mers at hash length k_velvet
(Velvet's role); write contigs_{k}_{kv}.fa and the merged contigs.fa with '>{k}_{kv}_' headers.  All gaps of a batch go
through ONE gf_assemble call per pair instead of >= 5 process launches per (gap, k, kv).

assemble_pipeline follows the reference's round structure (assemble_gaps.py:328-368): first-round assembly + contig merging
(MergeContigs.py) -> pick -> both-unmapped recruitment for the gaps still open (collect_both_unmapped_reads.py) -> second-round
assembly of those gaps -> pick -> merge -> pick -> high-quality bridging reads + merge -> pick at anchor 15 -> extended fills."""
import os

from . import fastq_io
from . import sam_io
from .hip_api import GapFill

kmer_len_list = []
working_folder = ""
min_count = 2        # KMC's default -ci (the reference passes none, assemble_gaps.py:96)
min_contig = 40      # velvetg -min_contig_lgth 40 (:117)
_gf = None
_first_round = None  # `-c All` on resident libraries: {gap id: {(k, kv): [(seq, n_nodes, cov_sum)]}} of the device pipeline (set_first_round)


def set_first_round(res):
    """The first assembly round already ran on the device, on the pools the Collect stage left in HBM (device_collect.py): the next
    assemble_ids call writes ITS contigs instead of reading the per-gap FASTQ files back and assembling them again.  Later rounds
    (pools grown by the both-unmapped recruitment, assemble_gaps.py:349-351) go through the files as before.  res: pipeline.Results
    with .keys (gap ids) and .k_pairs."""
    global _first_round
    per = {}
    ctg, seq = res.contigs, res.seq
    order = sorted(range(len(ctg)), key=lambda i: (int(ctg[i]["gap"]), int(ctg[i]["k"]), int(ctg[i]["kv"]), -int(ctg[i]["length"]),
                                                    seq[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])]))
    for i in order:          # the host entry point's order: length descending, then sequence
        c = ctg[i]
        per.setdefault(res.keys[int(c["gap"])], {}).setdefault((int(c["k"]), int(c["kv"])), []).append(
            (seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode(), int(c["n_nodes"]), int(c["cov_sum"])))
    _first_round = {"contigs": per, "pairs": set(res.k_pairs), "read_len": res.read_len}


def _ctx():
    global _gf
    if _gf is None:
        _gf = GapFill(int(os.environ.get("GF_DEVICE", "0")))
    return _gf


def velvet_kv(kv):
    """velveth runs at an odd hash length: an even value is lowered by one (Velvet's documented behaviour)."""
    return kv if kv & 1 else kv - 1


def format_contigs(contigs):
    """[(seq, n_nodes, cov_sum)] -> Velvet-style FASTA: NODE_{n}_length_{kmers}_cov_{cov:.6f}, 60 columns."""
    out = []
    for n, (seq, nodes, cov) in enumerate(contigs, 1):
        out.append(">NODE_%d_length_%d_cov_%.6f\n" % (n, nodes, cov / float(nodes)))
        out.extend(seq[i:i + 60] + "\n" for i in range(0, len(seq), 60))
    return "".join(out)


def assemble_ids(ids, gf=None):
    """run_assembly for a batch of gap ids (assemble_gaps.py:82-136)."""
    global _first_round
    gf = gf or _ctx()
    ids = [i for i in ids if os.path.exists("%sgap_reads/%s.fastq" % (working_folder, i))]   # :272-274
    if not ids:
        return
    pairs = [(int(k), int(kv)) for k, kv in kmer_len_list]
    per = {}
    first, _first_round = _first_round, None
    if first is not None:       # contigs of the device pipeline (same pools, same kernel, no trip through the files)
        L = first["read_len"]
        usable = [(k, velvet_kv(kv)) for k, kv in pairs if 16 <= k <= min(64, L) and 15 <= velvet_kv(kv) < k]
        assert set(usable) == first["pairs"], (usable, first["pairs"])
        for g, gid in enumerate(ids):
            for pair, lst in first["contigs"].get(gid, {}).items():
                per[(g,) + pair] = lst
    else:
        pools = [fastq_io.read_fastq_seqs("%sgap_reads/%s.fastq" % (working_folder, i)) for i in ids]
        packed, nm, off, L = fastq_io.pack_pools(pools)
        usable = [(k, velvet_kv(kv)) for k, kv in pairs if 16 <= k <= min(64, L) and 15 <= velvet_kv(kv) < k]
    if usable and first is None:
        ctg, seq = gf.assemble(packed, off, L, usable, min_count, min_contig, n_mask=nm)
        for c in ctg:
            per.setdefault((int(c["gap"]), int(c["k"]), int(c["kv"])), []).append(
                (seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode(), int(c["n_nodes"]), int(c["cov_sum"])))
    for g, gid in enumerate(ids):
        d = "%svelvet_temp/%s" % (working_folder, gid)
        os.makedirs(d, exist_ok=True)
        merged = []
        for (k, kv) in pairs:
            txt = format_contigs(per.get((g, k, velvet_kv(kv)), [])) if (k, velvet_kv(kv)) in usable else ""
            with open("%s/contigs_%d_%d.fa" % (d, k, kv), "w") as f:   # always created: a missing file kills the reference (:130)
                f.write(txt)
            merged.append("".join((">%d_%d_%s" % (k, kv, line[1:]) if line.startswith(">") else line)
                                  for line in txt.splitlines(True)))
        with open(d + "/contigs.fa", "w") as f:
            f.write("".join(merged))


def run_assembly(id):
    assemble_ids([id])


def _clipped_at(read, i, contig, j, seed_len, budget=2):
    """The ungapped alignment of `read` to `contig` through the exact seed read[i:i+seed_len] == contig[j:j+seed_len]: clipped when a read
    end lies beyond the contig's end or more than `budget` mismatches separate it from the seed."""
    start = j - i                                        # contig offset of the read's first base
    if start < 0 or start + len(read) > len(contig):
        return True
    left = sum(1 for a, b in zip(read[:i], contig[start:j]) if a != b)
    right = sum(1 for a, b in zip(read[i + seed_len:], contig[j + seed_len:start + len(read)]) if a != b)
    return left > budget or right > budget


MAX_SEED_OCC = 8       # occurrences of a seed kept per contig and strand: a seed repeated inside a contig makes every placement a candidate


def _placements_by_lookup(strands, reads, seed_len):
    """The definition: {read index: {contig: {(strand, contig offset of the read's first base): (i, j) of the FIRST seed found there}}}
    — every window of every read looked up among the windows of every contig strand (at most MAX_SEED_OCC occurrences per contig
    and strand, in offset order)."""
    seeds = {}
    for sid, strand in enumerate(strands):
        seen_here = {}
        for j in range(len(strand) - seed_len + 1):
            w = strand[j:j + seed_len]
            n_here = seen_here.get(w, 0)
            if n_here < MAX_SEED_OCC:
                seen_here[w] = n_here + 1
                seeds.setdefault(w, []).append((sid, j))
    out = {}
    for r, su in enumerate(reads):
        placed = {}
        for i in range(len(su) - seed_len + 1):
            for sid, j in seeds.get(su[i:i + seed_len], ()):
                placed.setdefault(sid >> 1, {}).setdefault((sid & 1, j - i), (i, j))
        if placed:
            out[r] = placed
    return out


_CODE = None


def _windows60(texts, seed_len):
    """All seed_len-base windows (seed_len <= 32) of the ACGT strings `texts`, 2 bits per base: (value u64, text index, offset)."""
    import numpy as np
    global _CODE
    if _CODE is None:
        _CODE = np.zeros(256, dtype=np.uint64)
        for c, v in zip(b"ACGT", range(4)):
            _CODE[c] = v
    lens = np.fromiter(map(len, texts), dtype=np.int64, count=len(texts))
    flat = _CODE[np.frombuffer("".join(texts).encode(), dtype=np.uint8)]
    n = len(flat) - seed_len + 1
    if n <= 0:
        z = np.zeros(0, dtype=np.int64)
        return np.zeros(0, dtype=np.uint64), z, z
    val = np.zeros(n, dtype=np.uint64)
    for t in range(seed_len):                     # (thirty shifted adds over the whole gap's text: no per-window work in the interpreter)
        val = (val << np.uint64(2)) | flat[t:t + n]
    start = np.concatenate([[0], np.cumsum(lens)[:-1]])
    owner = np.repeat(np.arange(len(texts)), lens)[:n]
    off = np.arange(n) - start[owner]
    ok = off + seed_len <= lens[owner]            # windows that run into the next text are none
    return val[ok], owner[ok], off[ok]


def _placements_by_sort(strands, reads, seed_len):
    """_placements_by_lookup for ACGT-only texts: the windows as 2-bit values, joined by sorting."""
    import numpy as np
    cv, cs, cj = _windows60(strands, seed_len)
    rv, rr, ri = _windows60(reads, seed_len)
    if not len(cv) or not len(rv):
        return {}
    # at most MAX_SEED_OCC occurrences per (strand, value), lowest offsets first: the windows come in (strand, offset) order, so ONE stable
    # sort by value leaves every value's entries in that order
    o = np.argsort(cv, kind="stable")
    cv, cs, cj = cv[o], cs[o], cj[o]
    first = np.ones(len(cv), dtype=bool)
    first[1:] = (cv[1:] != cv[:-1]) | (cs[1:] != cs[:-1])
    grp_start = np.maximum.accumulate(np.where(first, np.arange(len(cv)), 0))
    keep = np.arange(len(cv)) - grp_start < MAX_SEED_OCC
    cv, cs, cj = cv[keep], cs[keep], cj[keep]
    # every read window's run of equal contig values: the windows are looked up in sorted order (a binary search that starts where the last
    # one ended), the run's end comes from the contig side's own run lengths, and both go back to the windows' (read, offset) order
    run_first = np.ones(len(cv), dtype=bool)
    run_first[1:] = cv[1:] != cv[:-1]
    starts = np.flatnonzero(run_first)
    run_end = np.repeat(np.append(starts[1:], len(cv)), np.diff(np.append(starts, len(cv))))
    o = np.argsort(rv, kind="stable")
    lo_s = np.searchsorted(cv, rv[o], "left")
    at = np.minimum(lo_s, len(cv) - 1)
    hi_s = np.where((lo_s < len(cv)) & (cv[at] == rv[o]), run_end[at], lo_s)
    lo, hi = np.empty_like(lo_s), np.empty_like(lo_s)
    lo[o], hi[o] = lo_s, hi_s
    cnt = hi - lo
    if not cnt.any():
        return {}
    w = np.repeat(np.arange(len(rv)), cnt)                       # read window of every (window, occurrence) pair
    e = np.arange(int(cnt.sum())) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(lo, cnt)
    r, i, sid, j = rr[w], ri[w], cs[e], cj[e]
    # per (read, strand, diagonal) the lowest read offset: the pairs come in (read, read offset) order, so a stable sort by the
    # combined key keeps the first seed of every diagonal in front
    n_sid = len(strands)
    key = ((r * n_sid + sid).astype(np.uint64) << np.uint64(32)) | (j - i + (1 << 31)).astype(np.uint64)
    o = np.argsort(key, kind="stable")
    key, r, i, sid, j = key[o], r[o], i[o], sid[o], j[o]
    head = np.ones(len(r), dtype=bool)
    head[1:] = key[1:] != key[:-1]
    out = {}
    for r_, i_, sid_, j_ in zip(r[head].tolist(), i[head].tolist(), sid[head].tolist(), j[head].tolist()):
        out.setdefault(r_, {}).setdefault(sid_ >> 1, {})[(sid_ & 1, j_ - i_)] = (i_, j_)
    return out


def bridging_reads(contigs, reads, seed_len=30):
    """contigs: [(name, SEQUENCE)]; reads: {id: sequence} -> [(id, sequence)] of the reads that align CLIPPED to at least two contigs
    (see GapAssembler.collect_high_quality_unmap_to_contigs_reads), in the order of `reads`."""
    from .pick_contigs import revcomp
    strands = []
    for _, s in contigs:
        strands += [s, revcomp(s)]
    if len(contigs) < 2:          # (a bridge is clipped at two contigs at least)
        return []
    items = list(reads.items())
    ups = [seq.upper() for _, seq in items]
    plain = seed_len <= 32 and not (set("".join(strands)) | set("".join(ups))) - set("ACGT")
    placements = (_placements_by_sort if plain else _placements_by_lookup)(strands, ups, seed_len)
    bridges = []
    for r, placed in sorted(placements.items()):
        su = ups[r]
        # clipped at a contig = it shares a seed with it and NO candidate placement aligns end to end (bwa reports the
        # best alignment: a read that fits somewhere in the contig is no bridge, whatever its other seed hits look like)
        clipped = [ci for ci, pl in placed.items()
                   if all(_clipped_at(su, i, strands[2 * ci + st], j, seed_len) for (st, _), (i, j) in pl.items())]
        if len(clipped) >= 2:                                                             # clipped at two contigs at least (:213)
            bridges.append(items[r])
    return bridges


def bridging_reads_batch(items, seed_len=30, budget=2):
    """bridging_reads for many gaps in ONE host call of the library (gf_bridging_reads, csrc/textio.hip: the definition above in C++,
    hashing instead of per-gap numpy sorts).  items = [(contigs [(name, SEQUENCE)], reads {id: sequence})] -> per item [(id, sequence)] of
    the bridging reads, in the order of `reads`."""
    import ctypes as C

    import numpy as np

    from . import _lib as B
    ctext, coff, cset, rtext, roff, rset = [], [0], [0], [], [0], [0]
    read_items = []
    for contigs, reads in items:
        for _, s_ in contigs:
            ctext.append(s_)
            coff.append(coff[-1] + len(s_))
        cset.append(len(coff) - 1)
        it = list(reads.items())
        read_items.append(it)
        for _, s_ in it:
            rtext.append(s_)
            roff.append(roff[-1] + len(s_))
        rset.append(len(roff) - 1)
    out = np.zeros(max(1, len(roff) - 1), dtype=np.uint8)
    a = lambda x: np.asarray(x, dtype=np.uint64)
    co, cs, ro, rs = a(coff), a(cset), a(roff), a(rset)
    rc = B.lib().gf_bridging_reads(None, "".join(ctext).encode(), B._p(co), B._p(cs), "".join(rtext).encode(), B._p(ro), B._p(rs), len(items),
                                   int(seed_len), int(budget), B._p(out))
    if rc:
        raise B.GapFillError(rc, "gf_bridging_reads")
    res, at = [], 0
    for it in read_items:
        res.append([it[q] for q in range(len(it)) if out[at + q]])
        at += len(it)
    return res


class GapAssembler:
    def __init__(self, sf_fai, sf_pos, n_jobs, working_space, kmer_list=None, gf=None, bam_list=None, samtools_path=None):
        global kmer_len_list, working_folder, _gf
        self.bam_list = list(bam_list or [])
        self.samtools_path = samtools_path
        if kmer_list is not None:
            kmer_len_list = list(kmer_list)
        self.sf_fai = sf_fai
        self.sf_pos = sf_pos
        self.n_jobs = int(n_jobs)
        working_folder = working_space
        if gf is not None:
            _gf = gf

    def prepare_list(self):
        sidx = {n: i for i, n in enumerate(sam_io.read_fai(self.sf_fai))}
        _, keys = sam_io.read_gap_positions(self.sf_pos, sidx)
        return [k for k in keys if os.path.exists("%sgap_reads/%s.fastq" % (working_folder, k))]

    def assembly(self, id_list):
        for sub in ("kmc_temp", "temp", "kmers", "velvet_temp"):
            os.makedirs(working_folder + sub, exist_ok=True)
        assemble_ids(id_list)

    assembly_given_list = assembly

    def run_contigs_merge(self, fa_list):
        """assemble_gaps.py:301-306 / run_merge :138-145: dedup + ContigsMerger + dedup per gap (MergeContigs.merge_contigs: contigs.fa
        becomes the merged set, the assembly's own contigs move to original_contigs_before_merging.fa).  The reference's merge is
        a best-effort step — each gap is its own Pool task and a failed one leaves its contigs.fa alone —, so a failure here is
        reported and the pipeline goes on with the contigs as they are."""
        from .MergeContigs import merge_contigs
        from ._lib import GapFillError
        try:
            return merge_contigs(_ctx(), working_folder, fa_list)
        except GapFillError:            # a faulted kernel / HIP error: the context is not fit to go on picking and assembling with
            raise
        except (OSError, ValueError) as e:      # host-side trouble with one gap's files: the optional step is skipped, the picks go on
            import sys
            sys.stderr.write("contig merging skipped for %d gaps: %r\n" % (len(fa_list), e))
            return {}

    def collect_high_quality_unmap_to_contigs_reads(self, id_list, seed_len=30):
        """run_collect_high_quality_unmap_to_contig_reads (assemble_gaps.py:166-217): the gap's high-quality reads
        (gap_reads_high_quality/{id}.fastq, MAPQ 60 only) that align CLIPPED to at least two of its merged contigs are bridges
        between them; the merged contigs.fa is dropped, the assembly's own contigs come back (original_contigs_before_merging.fa)
        and the bridging reads are appended as FASTA records, so that the next merge can chain through them.  `bwa mem` is replaced
        by seed and extend: a read aligns to a contig when they share an exact stretch of seed_len bases (bwa's default output
        threshold -T 30) on either strand, and the alignment is CLIPPED (is_qualified_clipped(cigar, 1), :196-203) when, extended from
        the seed without gaps, a read end runs off the contig or gathers more than two mismatches on its way there — a read that lies
        inside a contig with a sequencing error or two is an end-to-end alignment for bwa and no bridge.  One deviation in
        the file handling: the reference removes contigs.fa when original_contigs_before_merging.fa is missing (:206-210); here the
        merged file stays.  Returns the number of reads appended."""
        from .pick_contigs import read_fasta
        work = []                                                                         # (folder, contigs, reads) of the gaps that take part
        for gid in id_list:
            d = "%svelvet_temp/%s/" % (working_folder, gid)
            sf_reads = "%sgap_reads_high_quality/%s.fastq" % (working_folder, gid)
            if not os.path.exists(d + "contigs.fa") or not os.path.exists(sf_reads):      # (:170-179)
                continue
            contigs = [(n, s.upper()) for n, s in read_fasta(d + "contigs.fa")]
            reads = {}                                                                    # first record of an id counts (:188-191)
            with open(sf_reads) as f:
                lines = f.read().split("\n")
            for q in range(0, len(lines) - 1, 4):
                if lines[q]:
                    reads.setdefault(lines[q][1:].split()[0], lines[q + 1].strip() if q + 1 < len(lines) else "")
            work.append((d, contigs, reads))
        # all gaps of the round through ONE library call (the reference runs bwa per gap; the numpy form of this module took 2 ms per gap)
        bridges_of = bridging_reads_batch([(c, r) for _, c, r in work], seed_len) if work else []
        n_added = 0
        for (d, _, _), bridges in zip(work, bridges_of):
            if os.path.exists(d + "original_contigs_before_merging.fa"):                  # (:206-210)
                os.replace(d + "original_contigs_before_merging.fa", d + "contigs.fa")
            with open(d + "contigs.fa", "a") as f:
                for rid, seq in bridges:
                    f.write(">%s\n%s\n" % (rid, seq))
            n_added += len(bridges)
        return n_added

    def pick_already_constructed(self, contigs_select, fa_list, sf_picked):
        picked = contigs_select.get_already_picked(sf_picked)
        return [k for k in fa_list if k not in picked]

    def assemble_pipeline(self):
        """The reference's rounds (assemble_gaps.py:328-368): assemble + merge; pick the gaps whose contigs are anchored by both
        flanks (anchor length 30 = the reference's first bwa_min_score); for the gaps still open recruit the both-unmapped pairs that
        share k-mers with their contigs and assemble again (:344-351), pick; merge those that are still open, pick (:353-356); bring
        in the high-quality reads that bridge two contigs and merge once more (:358-361); pick at 15 (:364-366); what is still open
        gets the extended (partial, 'NN'-joined) fill (:367-368)."""
        from .pick_contigs import ContigsSelection
        fa_list = self.prepare_list()
        self.assembly(fa_list)
        merged = sum(1 for v in self.run_contigs_merge(fa_list).values() if v)
        sf_picked = working_folder + "../picked_seqs.fa"
        cs = ContigsSelection(working_folder)
        closed = cs.pick_full_constructed_contigs(30, fa_list, sf_picked)
        remain = self.pick_already_constructed(cs, fa_list, sf_picked)
        recruited = 0
        if self.bam_list and self.samtools_path and remain:
            from .collect_both_unmapped_reads import BothUnmappedReadsCollector
            ks = [int(k) for k, _ in kmer_len_list if 16 <= int(k) <= 64]
            burc = BothUnmappedReadsCollector(working_folder, self.samtools_path, _ctx(), min(ks) if ks else 31)
            burc.collect_both_unmapped_reads(self.bam_list, remain)
            recruited = sum(1 for key in remain if os.path.exists("%sunmapped_reads/%s.fastq" % (working_folder, key)))
        self.assembly_given_list(remain)                                     # second-round assembly (:349-351)
        closed += cs.pick_full_constructed_contigs(30, remain, sf_picked)
        remain = self.pick_already_constructed(cs, remain, sf_picked)
        merged += sum(1 for v in self.run_contigs_merge(remain).values() if v)             # second-round merging (:353-356)
        closed += cs.pick_full_constructed_contigs(30, remain, sf_picked)
        remain = self.pick_already_constructed(cs, remain, sf_picked)
        bridges = self.collect_high_quality_unmap_to_contigs_reads(remain)   # (:358-361)
        merged += sum(1 for v in self.run_contigs_merge(remain).values() if v)
        closed += cs.pick_full_constructed_contigs(15, remain, sf_picked)    # (:364-366)
        remain = self.pick_already_constructed(cs, remain, sf_picked)
        extended = cs.pick_extended_contigs(15, remain, sf_picked)           # partial fills, left + 'NN' + right (:367-368)
        return {"gaps": len(fa_list), "closed": closed, "extended": extended, "second_round_gaps": recruited,
                "gaps_with_merged_contigs": merged, "bridging_reads": bridges}
