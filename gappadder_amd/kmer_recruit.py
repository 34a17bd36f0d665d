"""North-star recruitment mode: pull candidate read PAIRS out of whole FASTQ files by flank k-mers (no alignment needed).
Every read of the left and the right file is parsed and packed to 2 bits ON the GPU (gf_fastq_pack) and screened there against the canonical k-mers of each
gap's flanks (gf_screen_reads); a hit recruits the read and its mate.  The result is a {readId -> set(gapKey)} map per mate
file that DiscordantReadsCollector unions with the alignment-derived lists before it writes gap_reads/{gapKey}.fastq.
Off by default (the reference recruits by alignment only); enabled with "parameters": {"kmer_screen": K} in the JSON config."""
import os

from . import fastq_io
from . import sam_io
from .gnrt_pos_true_seqs import read_fasta
from .hip_api import GapFill


def read_fastq_ids_seqs(path):
    ids, seqs = [], []
    with open(path) as f:
        while True:
            h = f.readline()
            if not h:
                break
            ids.append(h.split()[0].split("/")[0][1:].rstrip())     # run_multi_threads_discordant.py:212-214
            seqs.append(f.readline().strip())
            f.readline()
            f.readline()
    return ids, seqs


def flank_table(working_folder, keys):
    out = []
    for key in keys:
        p = "%sflank_regions/%s.fa" % (working_folder, key)
        fl = dict(read_fasta(p)) if os.path.exists(p) else {}
        out.append((fl.get(key + "_left", ""), fl.get(key + "_right", "")))
    return out


def screen_fastq_pair(gf, sf_fai, sf_gap_pos, working_folder, sf_left, sf_right, k, min_hits=1):
    """-> (extra_left, extra_right): {readId: set(gapKey)}.  working_folder = the run's top folder (holds flank_regions/)."""
    names = sam_io.read_fai(sf_fai)
    sidx = {n: i for i, n in enumerate(names)}
    gaps, keys = sam_io.read_gap_positions(sf_gap_pos, sidx)
    gf.set_gaps(gaps, len(names), flank_table(working_folder, keys))
    extra = ({}, {})
    for m, path in enumerate((sf_left, sf_right)):
        with open(path, "rb") as f:
            text = f.read()
        if not text:
            continue
        # FASTQ text -> packed reads on the GPU (gf_fastq_pack); ids are cut from the text only for the reads that hit
        L = len(text.split(b"\n", 2)[1].rstrip(b"\r")) if text.count(b"\n") >= 1 else 0
        if L < 1:
            continue
        packed, nm, hdr, st = gf.fastq_pack(text, L)
        if st & 1:      # some read is longer than the first one: size to the longest (shorter ones are padded with masked N)
            L = max(len(l.rstrip(b"\r")) for l in text.split(b"\n")[1::4])
            packed, nm, hdr, st = gf.fastq_pack(text, L)
        if L < k or not len(packed):
            continue
        for h in gf.screen_reads(packed, L, k, min_hits, n_mask=nm):
            o = int(hdr[int(h["read"])])
            rid = text[o:text.index(b"\n", o)].split()[0].split(b"/")[0][1:].rstrip().decode()   # run_multi_threads_discordant.py:212-214
            key = keys[int(h["gap"])]
            extra[m].setdefault(rid, set()).add(key)
            extra[1 - m].setdefault(rid, set()).add(key)            # the mate comes along ("candidate read pairs")
    return extra
