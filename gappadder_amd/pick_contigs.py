"""Flank anchoring -> gap sequence selection (SURVEY.md §8f "next" rank 1; mirrors ContigsSelection, pick_contigs.py:64-358,
542-581).  The reference aligns the two flanks to the gap's contigs with `bwa mem -T {score} -a` and keeps the contig both
flanks hit on the same strand; bwa is out of scope here, so the anchors are EXACT matches: the last `score` bases of the left
flank and the first `score` bases of the right flank (score = the reference's bwa_min_score: 30, later 15).  Among the
qualifying contigs — every occurrence of the anchors, both orientations — the longest span wins (pick_contigs.py:300-321);
the same rule runs on the device as gf_pick_anchored_dev (csrc/pick.hip); the picked slice is contig[left_end : right_start + 1]
in flank orientation — the +1 reproduces the reference's 1-based/0-based slice (:341-349); header '>{gapId}_{contigName}'
(:352).  A gap with a picked sequence is what this build reports as "closed"."""
import os

_COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def revcomp(s):
    return s.translate(_COMP)[::-1]


def read_fasta(path):
    out, name, chunks = [], None, []
    with open(path) as f:
        for line in f:
            line = line.rstrip("\n")
            if line.startswith(">"):
                if name is not None:
                    out.append((name, "".join(chunks)))
                name, chunks = line[1:].split()[0], []
            elif line:
                chunks.append(line)
    if name is not None:
        out.append((name, "".join(chunks)))
    return out


def anchor_spans(seq, la, ra):
    """Both orientations of one contig: [(span, left_end, right_start, oriented)] for every orientation in which the left anchor
    occurs and the right anchor occurs at or behind its end — leftmost left anchor, rightmost right anchor, i.e. the longest
    span the anchors allow (the reference keeps the longest span among its bwa hits, pick_contigs.py:300-321)."""
    out = []
    a = len(la)
    for oriented in (seq, revcomp(seq)):
        i = oriented.find(la)
        if i < 0:
            continue
        j = oriented.rfind(ra)
        if j < i + a:
            continue
        out.append((j - (i + a), i + a, j, oriented))
    return out


def pick_gap_sequence(contigs, left_flank, right_flank, anchor_len):
    """contigs: [(name, seq)].  Returns (name, gap_seq, oriented_contig) or None.  Longest span over all contigs and both
    orientations; ties go to the first contig of the list, forward orientation first."""
    if len(left_flank) < anchor_len or len(right_flank) < anchor_len:
        return None
    la, ra = left_flank[-anchor_len:], right_flank[:anchor_len]
    if any(c not in "ACGT" for c in la + ra):
        return None
    best = None
    for name, seq in contigs:
        for span, left_end, j, oriented in anchor_spans(seq, la, ra):
            if best is None or span > best[0]:
                best = (span, name, oriented[left_end:j + 1], oriented)
    return None if best is None else best[1:]


def pick_extended_sequence(contigs, left_flank, right_flank, anchor_len):
    """The fallback of the last round (run_pick_extended_contig, pick_contigs.py:361-539): no contig carries both anchors in
    order, so the gap is filled from each side as far as a contig reaches — left part = what follows the left anchor in the
    contig that reaches furthest into the gap, right part = what precedes the right anchor — joined by 'NN' (:513-520).
    The reference takes, per side, the bwa hit with the longest match and breaks ties by a comparison that is constant in
    Python 2 (int > str, :444, :457), i.e. by dict order; with exact anchors every match has the same length, so this build
    DEFINES the tie: the longest extension wins, then the first contig of the list, forward orientation first.  When both sides
    pick the same contig the reference keeps only the side with the longer match, the right side on a tie (:468-486): here always
    the right side.  Returns (left_name, right_name, sequence, contig_text) or None when neither anchor occurs."""
    if len(left_flank) < anchor_len or len(right_flank) < anchor_len:
        return None
    la, ra = left_flank[-anchor_len:], right_flank[:anchor_len]
    if any(c not in "ACGT" for c in la + ra):
        return None
    best_l = best_r = None        # (extension length, name, extension, contig as written)
    for name, seq in contigs:
        for oriented in (seq, revcomp(seq)):
            i = oriented.find(la)
            if i >= 0:
                ext = oriented[i + anchor_len:]
                if best_l is None or len(ext) > best_l[0]:
                    best_l = (len(ext), name, ext, seq)
            j = oriented.rfind(ra)
            if j >= 0:
                ext = oriented[:j]
                if best_r is None or len(ext) > best_r[0]:
                    best_r = (len(ext), name, ext, seq)
    if best_l is None and best_r is None:
        return None
    if best_l is not None and best_r is not None and best_l[1] == best_r[1]:
        best_l = None
    left_name, left_seq = (best_l[1], best_l[2]) if best_l else ("", "")
    right_name, right_seq = (best_r[1], best_r[2]) if best_r else ("", "")
    seq = left_seq + "NN" + right_seq
    if best_l and best_r:
        contig_text = best_l[3] + "NN" + best_r[3]
    else:
        contig_text = (best_l or best_r)[3]
    if seq == "NN":
        return None
    return left_name, right_name, seq, contig_text


class ContigsSelection:
    def __init__(self, working_space):
        self.working_folder = working_space

    def _pick_one(self, gid, anchor_len):
        wf = self.working_folder
        sf_flank = wf + "../flank_regions/%s.fa" % gid
        sf_contig = wf + "velvet_temp/%s/contigs.fa" % gid
        for p in (wf + "velvet_temp/%s/picked_seqs.fa" % gid, wf + "velvet_temp/%s/picked_contigs.fa" % gid):
            if os.path.exists(p):
                os.remove(p)
        if not (os.path.exists(sf_flank) and os.path.exists(sf_contig)):
            return False
        fl = dict(read_fasta(sf_flank))
        res = pick_gap_sequence(read_fasta(sf_contig), fl.get(gid + "_left", ""), fl.get(gid + "_right", ""), anchor_len)
        if res is None:
            return False
        name, gap_seq, oriented = res
        if gap_seq:
            with open(wf + "velvet_temp/%s/picked_seqs.fa" % gid, "w") as f:
                f.write(">%s_%s\n%s\n" % (gid, name, gap_seq))
        with open(wf + "velvet_temp/%s/picked_contigs.fa" % gid, "w") as f:
            f.write(">%s_%s\n%s\n" % (gid, name, oriented))
        return bool(gap_seq)

    def pick_full_constructed_contigs(self, bwa_score, fa_list, sf_picked):
        n = 0
        for gid in fa_list:
            if self._pick_one(gid, int(bwa_score)):
                n += 1
            for src, dst in (("picked_seqs.fa", sf_picked), ("picked_contigs.fa", sf_picked + "_ori.txt")):
                p = self.working_folder + "velvet_temp/%s/%s" % (gid, src)
                if os.path.exists(p):
                    with open(dst, "a") as out, open(p) as f:   # the reference appends with `cat >>` (:564-572)
                        out.write(f.read())
        return n

    def pick_extended_contigs(self, bwa_score, fa_list, sf_picked):
        """pick_contigs.py:583-603: per gap velvet_temp/{id}/picked_seqs.fa + picked_contigs.fa with header
        '>{id}_{left contig}_{right contig}_extended', appended to the ledger like the full picks."""
        n = 0
        for gid in fa_list:
            wf = self.working_folder
            sf_flank = wf + "../flank_regions/%s.fa" % gid
            sf_contig = wf + "velvet_temp/%s/contigs.fa" % gid
            if not (os.path.exists(sf_flank) and os.path.exists(sf_contig)):
                continue
            fl = dict(read_fasta(sf_flank))
            res = pick_extended_sequence(read_fasta(sf_contig), fl.get(gid + "_left", ""), fl.get(gid + "_right", ""), int(bwa_score))
            if res is None:
                continue
            left_name, right_name, seq, contig_text = res
            hdr = ">%s_%s_%s_extended\n" % (gid, left_name, right_name)
            for fn, body, dst in (("picked_seqs.fa", seq, sf_picked), ("picked_contigs.fa", contig_text, sf_picked + "_ori.txt")):
                with open(wf + "velvet_temp/%s/%s" % (gid, fn), "w") as f:
                    f.write(hdr + body + "\n")
                with open(dst, "a") as out:
                    out.write(hdr + body + "\n")
            n += 1
        return n

    def get_already_picked(self, sf_picked):
        picked = {}
        if os.path.exists(sf_picked):
            with open(sf_picked) as f:
                for line in f:
                    if line[0] == ">":
                        fl = line[1:].split("_")
                        picked[fl[0] + "_" + fl[1]] = 1
        return picked
