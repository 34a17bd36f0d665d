"""Second-round recruitment of pairs with BOTH mates unmapped (mirrors collect_both_unmapped_reads.py).

File side — identical to the reference and pinned by the reference-generated fixture tests/golden/twolib/round2.json.gz:
  run_collect_both_unmapped            `samtools view -f 12 BAM` -> BAM.both_unmapped.sam and BAM.both_unmapped.fq with names
                                       `@{QNAME}_2` when FLAG > 128 (sic: a comparison, not a bit test, :26) else `_1`
                                       (collect_both_unmapped_reads.py:14-34)
  collect_both_unmapped_reads          cat of the per-BAM files -> both_unmapped.fq; both_unmapped_1.fq / _2.fq list, in order
                                       of the `_1` records, `@{QNAME}` + the mate-1 resp. mate-2 record (:187-236)
  align_unmapped_to_contigs            gap_contigs_all.fa = every contig of every listed gap as `>{gapKey}-{contigId}` + one
                                       sequence line (:136-153); recruited records are APPENDED to gap_reads/{gapKey}.fastq
                                       and written to unmapped_reads/{gapKey}.fastq as `@{name}_{1|2}` + seq/+/qual (:113-124)

Recruitment side — where the reference runs `bwa mem -a` of the pairs against the contigs and keeps every read with an
alignment record on a contig of the gap, plus its mate when the mate is unmapped or lies in another gap's contigs (:56-103) —
this build has no aligner: a read is recruited for a gap when it shares at least one canonical k-mer with that gap's contigs
(the same exact GPU screen that recruits against the flanks, gf_screen_reads), and its mate always comes along.  That is a
definition of this build (bwa is an unpinned third-party tool, absent here: parity unpinned for this step), checked against
the oracle's k-mer predicate."""
import os
import shutil
import subprocess

from . import bam_io
from . import fastq_io
from .gnrt_pos_true_seqs import read_fasta
from .pick_contigs import _cached_fasta
from .hip_api import GapFill


PREPARED = set()      # BAM paths whose .both_unmapped.sam / .fq this process wrote already, on its one pass over the file (device_collect.py)


def run_collect_both_unmapped(sf_bam, samtools_path, gf=None):
    sf_both_unmap = sf_bam + ".both_unmapped.sam"
    if bam_io.is_builtin(samtools_path):      # `samtools view -f 12`: both FLAG bits 4 and 8 set; the 11 mandatory columns
        if os.path.abspath(sf_bam) in PREPARED and os.path.exists(sf_both_unmap) and os.path.exists(sf_bam + ".both_unmapped.fq"):
            PREPARED.discard(os.path.abspath(sf_bam))
            return
        # the records' bytes come back from the device in one gather per piece of the file; both files are formatted from them in one
        # host pass (gf_bam_records_text) — the .fq needs no second reading of the .sam
        import numpy as np
        from . import textio
        gf = gf or GapFill(0)
        with open(sf_both_unmap, "wb") as f, open(sf_bam + ".both_unmapped.fq", "wb") as fout:
            for recs, cols in bam_io.decode_file(gf, sf_bam, []):
                both = ((recs["flag"] & 12) == 12).nonzero()[0]
                if not len(both):
                    continue
                begin = cols.rb[both]
                end = np.append(cols.rb[1:], np.uint64(cols.end))[both]
                blob = gf.bam_fetch(begin, end)
                at = np.concatenate([[0], np.cumsum((end - begin).astype(np.int64))[:-1]]).astype(np.uint64)
                sam, fq = textio.bam_records_text(blob, at, cols.names, gf.handle)
                f.write(sam)
                fout.write(fq)
        return
    with open(sf_both_unmap, "w") as f:
        subprocess.run([samtools_path, "view", "-f", "12", sf_bam], check=True, stdout=f)
    with open(sf_both_unmap) as fin, open(sf_bam + ".both_unmapped.fq", "w") as fout:
        for line in fin:
            fields = line.split()
            fout.write("@" + fields[0] + ("_2\n" if int(fields[1]) > 128 else "_1\n"))      # :26-29
            fout.write(fields[9] + "\n+\n" + fields[10] + "\n")


def kmer_recruit_unmapped(gf, gap_contigs, names, seqs, k, min_hits=1):
    """gap_contigs: [[contig sequence, ...] per gap]; names/seqs: the both-unmapped records (`{q}_1` / `{q}_2`).
    -> per gap, the list of record indices recruited (a hit recruits the record and its mate), in record order."""
    import numpy as np
    from . import _lib as B
    n_gaps = len(gap_contigs)
    out = [[] for _ in range(n_gaps)]
    if not n_gaps or not any(len(s) >= k for s in seqs):      # e.g. SEQ '*' in the SAM text: nothing to screen
        return out
    gaps = np.zeros(n_gaps, dtype=B.GAP)          # coordinates are irrelevant to the screen
    gaps["idx_in_scaffold"] = np.arange(n_gaps) + 1
    # the contigs of a gap, joined by N (k-mers touching a non-ACGT byte are not indexed), play the role of a flank
    gf.set_gaps(gaps, 1, [("N".join(c), "") for c in gap_contigs])
    mate = dict(zip(names, range(len(names))))                # (a name that occurs twice: its last record, as an assignment loop leaves it)
    lens = np.fromiter(map(len, seqs), dtype=np.int64, count=len(seqs))
    got = [set() for _ in range(n_gaps)]
    for L in np.unique(lens).tolist():
        if L < k:
            continue
        idx = np.nonzero(lens == L)[0].tolist()
        # one length per group: the sequences back to back are the packer's input as they are
        packed, nm = GapFill.pack_reads("".join([seqs[i] for i in idx]).encode(), L, with_mask=True)
        for h in gf.screen_reads(packed, L, k, min_hits, n_mask=nm):
            i = idx[int(h["read"])]
            g = int(h["gap"])
            got[g].add(i)
            other = names[i][:-1] + ("2" if names[i].endswith("1") else "1")
            if other in mate:
                got[g].add(mate[other])
    return [sorted(s) for s in got]


class BothUnmappedReadsCollector:
    def __init__(self, working_space, samtools_path="samtools", gf=None, k=31):
        self.wf = working_space
        self.samtools_path = samtools_path
        self.gf = gf
        self.k = int(k)

    def collect_both_unmapped_reads(self, bam_list, id_list):
        wf = self.wf
        for sf_bam in bam_list:
            run_collect_both_unmapped(sf_bam, self.samtools_path, self.gf)
        with open(wf + "both_unmapped.fq", "wb") as fout:                                  # `cat` of the per-BAM files (:197-202)
            for sf_bam in bam_list:
                with open(sf_bam + ".both_unmapped.fq", "rb") as f:
                    shutil.copyfileobj(f, fout, 16 << 20)
        with open(wf + "both_unmapped.fq") as fin:
            lines = fin.read().split("\n")
        # head -> "seq\n+\nqual\n" (:205-220), one entry per four lines while three more lines follow; a head seen again keeps its place
        # and takes the later record (what assigning in a loop does) — built column-wise: a C2-sized run holds 1.6 M of these records
        n = max(0, (len(lines) - 3 + 3) // 4)
        self.reads = dict(zip([h.rstrip()[1:] for h in lines[0:4 * n:4]],
                              [a.rstrip() + "\n" + b.rstrip() + "\n" + c.rstrip() + "\n"
                               for a, b, c in zip(lines[1:4 * n:4], lines[2:4 * n:4], lines[3:4 * n:4])]))
        del lines
        reads = self.reads
        with open(wf + "both_unmapped_1.fq", "w") as f_left, open(wf + "both_unmapped_2.fq", "w") as f_right:
            left, right = [], []
            for key, rec in reads.items():                                                 # insertion order
                if key[-1] == "1":
                    read_id = key[:-2]
                    left.append("@" + read_id + "\n" + rec)
                    right.append("@" + read_id + "\n" + reads[read_id + "_2"])            # KeyError if absent, as in the reference
                    if len(left) >= 65536:
                        f_left.write("".join(left))
                        f_right.write("".join(right))
                        left, right = [], []
            f_left.write("".join(left))
            f_right.write("".join(right))
        self.align_unmapped_to_contigs(id_list)

    def align_unmapped_to_contigs(self, fa_list):
        wf = self.wf
        keys, contigs = [], []
        with open(wf + "gap_contigs_all.fa", "w") as fout:
            for key in fa_list:
                sf_ctg = wf + "velvet_temp/%s/contigs.fa" % key
                if not os.path.exists(sf_ctg):
                    continue
                recs = _cached_fasta(sf_ctg, lambda p_: list(read_fasta(p_)))
                for name, seq in recs:
                    fout.write(">" + key + "-" + name + "\n" + seq + "\n")
                if recs:
                    keys.append(key)
                    contigs.append([s for _, s in recs])
        if not keys or not self.reads:
            return {}
        gf = self.gf or GapFill(int(os.environ.get("GF_DEVICE", "0")))
        names = list(self.reads)
        seqs = [self.reads[n].split("\n", 1)[0] for n in names]
        picked = kmer_recruit_unmapped(gf, contigs, names, seqs, self.k)
        os.makedirs(wf + "unmapped_reads", exist_ok=True)
        out = {}
        for key, idx in zip(keys, picked):
            if not idx:
                continue
            text = "".join("@" + names[i] + "\n" + self.reads[names[i]] for i in idx)
            with open(wf + "gap_reads/%s.fastq" % key, "a") as f:                          # :113-117
                f.write(text)
            with open(wf + "unmapped_reads/%s.fastq" % key, "w") as f:                     # :119-123
                f.write(text)
            out[key] = [names[i] for i in idx]
        return out
