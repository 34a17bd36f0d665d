"""Flank anchoring -> gap sequence selection (SURVEY.md §8f "next" rank 1; mirrors ContigsSelection, pick_contigs.py:64-358,
361-539, 542-603).  The reference aligns the two flanks to the gap's contigs with `bwa mem -T {score} -a`, and everything after
that is its own code: per contig and side the longest hit of each clip type (:97-146), the best same-strand pair of a left and a
right hit (:149-297), over the contigs the longest span (:300-321), the slice (:341-349) and the header (:352).  bwa is out of scope
here; its place is taken by EXACT anchors (`anchor_hits`): the last `score` bases of the left flank and the first `score` bases of
the right flank (score = the reference's bwa_min_score: 30, later 15).  The selection itself follows the reference hit for hit —
including what its coordinates do on the reverse strand (the slice then keeps the last base of the LEFT anchor instead of the
first base of the right one) — and is pinned on the reference's own answers (tests/golden/pick_kat.json.gz through
oracle/gp_oracle.py).  The same rule runs on the device as gf_pick_anchored_dev (csrc/pick.hip).  A gap with a picked sequence
is what this build reports as "closed"."""
import os

_COMP = str.maketrans("ACGTacgt", "TGCATGCA")            # gnrt_reverse_complementary, pick_contigs.py:19-33: upper-case output
_BOTH, _LEFT, _RIGHT, _NONE = 1, 2, 3, 4                 # clip types (pick_contigs.py:9-12)


def revcomp(s):
    return s.translate(_COMP)[::-1]


# The reference's rounds read a gap's small FASTA files again and again (flanks: twice per pick and round; contigs.fa: every merge, pick
# and recruit step — 23 000 parses for 1 000 gaps).  A parse is kept per path and handed out again while the file's (mtime, size, inode) stay
# what they were: one stat instead of open + read + split.  Files beyond 1 MB (a draft) are never kept.
_FASTA_CACHE, _FASTA_CACHE_BYTES = {}, [0]


def _cached_fasta(path, parse):
    st = os.stat(path)
    key = (st.st_mtime_ns, st.st_size, st.st_ino)
    hit = _FASTA_CACHE.get(path)
    if hit is not None and hit[0] == key:
        return list(hit[1])
    recs = parse(path)
    if st.st_size <= (1 << 20):
        if _FASTA_CACHE_BYTES[0] > (1 << 30):      # (a human-scale run: 20 000 gaps x a few files x tens of kB stay far below)
            _FASTA_CACHE.clear()
            _FASTA_CACHE_BYTES[0] = 0
        _FASTA_CACHE[path] = (key, recs)
        _FASTA_CACHE_BYTES[0] += st.st_size
    return list(recs)


def read_fasta(path):
    return _cached_fasta(path, _parse_fasta)


def _parse_fasta(path):
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


def anchor_hits(contigs, left_flank, right_flank, score):
    """The stand-in for `bwa mem -T {score} -a` (pick_contigs.py:79-86): [(side, reverse?, contig index, 1-based position in the
    contig, clip type, matched bases)], per contig forward left / forward right / reverse left / reverse right.  Forward: the
    LEFTMOST occurrence of the left anchor, the RIGHTMOST of the right anchor; reverse: the same in the reverse-complemented
    contig, reported in the contig's own coordinates with the clip on the other end, as SAM does."""
    a = int(score)
    if len(left_flank) < a or len(right_flank) < a:
        return []
    la, ra = left_flank[len(left_flank) - a:], right_flank[:a]
    if any(c not in "ACGT" for c in la + ra):
        return []
    l_clipped, r_clipped = len(left_flank) > a, len(right_flank) > a
    out = []
    for ci, (_, seq) in enumerate(contigs):
        n = len(seq)
        for rev, s in ((False, seq), (True, revcomp(seq))):
            i, j = s.find(la), s.rfind(ra)
            if i >= 0:       # left flank = [clipped part][anchor]: clip in front on the forward strand, behind on the reverse strand
                out.append(("left", rev, ci, n - i - a + 1 if rev else i + 1, _NONE if not l_clipped else _RIGHT if rev else _LEFT, a))
            if j >= 0:
                out.append(("right", rev, ci, n - j - a + 1 if rev else j + 1, _NONE if not r_clipped else _LEFT if rev else _RIGHT, a))
    return out


_PAIRS = ((_NONE, _NONE), (_NONE, _LEFT), (_NONE, _RIGHT), (_LEFT, _NONE), (_LEFT, _RIGHT), (_RIGHT, _NONE), (_RIGHT, _LEFT))


def select_full(hits):
    """pick_contigs.py:97-321 on hit tuples: (contig index, left pos, right pos, left match, right match, reverse?) or None."""
    table = {}                                           # contig -> side -> clip type -> (reverse?, match, pos)
    for side, rev, ci, pos, ct, m in hits:
        if ct == _BOTH:
            continue
        slot = table.setdefault(ci, {}).setdefault(side, {})
        if ct not in slot or m > slot[ct][1]:
            slot[ct] = (rev, m, pos)
    best = None
    for ci, sides in table.items():
        if len(sides) != 2:
            continue
        top, sel, rc = -1, None, False
        for lt, rt in _PAIRS:                            # the reference's seven pairs in its order (:173-291)
            l, r = sides["left"].get(lt), sides["right"].get(rt)
            if l is None or r is None or l[0] != r[0] or not top < l[1] + r[1]:
                continue
            top, sel = l[1] + r[1], (l[2], r[2], l[1], r[1])
            rc = rc or l[0]                              # (:176-177: set by any winning reverse pair, never cleared)
        if sel is None:
            continue
        lp, rp, lm, rm = sel
        span = (lp - (rp + rm)) if rc else (rp - (lp + lm))
        if span > (-1 if best is None else best[0]):     # longest span, the earlier contig on ties (:313-321)
            best = (span, ci, lp, rp, lm, rm, rc)
    return None if best is None else best[1:]


def pick_gap_sequence(contigs, left_flank, right_flank, anchor_len):
    """contigs: [(name, seq)].  Returns (name, gap_seq, contig as written to picked_contigs.fa) or None (pick_contigs.py:331-358)."""
    sel = select_full(anchor_hits(contigs, left_flank, right_flank, anchor_len))
    if sel is None:
        return None
    ci, lp, rp, lm, rm, rc = sel
    name, seq = contigs[ci]
    if rc:
        return name, revcomp(seq[rp + rm - 1:lp]), revcomp(seq)
    return name, seq[lp + lm - 1:rp], seq


def pick_extended_sequence(contigs, left_flank, right_flank, anchor_len):
    """The fallback of the last round (run_pick_extended_contig, pick_contigs.py:361-539): no contig carries both anchors in
    order, so the gap is filled from each side as far as a contig reaches and the parts are joined by 'NN' (:517-525).  Hit for
    hit the reference's rule on the stand-in's hits: clipped hits only (:388-389), per side the contig with the longest match —
    the anchors all match `anchor_len` bases and the reference's tie test is constant (int > str, :444, :457), so the FIRST contig
    with a hit; when both sides pick the same contig only the right side is used, and its slice then keeps the first anchor base
    (:480-486 vs :509-512); reverse-strand slices keep one anchor base as well (:496, :474).  Returns (left_name, right_name,
    sequence or None, picked_contigs text or None)."""
    first = {"left": None, "right": None}
    for side, rev, ci, pos, ct, m in anchor_hits(contigs, left_flank, right_flank, anchor_len):
        want = (_RIGHT if rev else _LEFT) if side == "left" else (_LEFT if rev else _RIGHT)
        if ct == want and first[side] is None:
            first[side] = (ci, pos, m, rev)
    l, r = first["left"], first["right"]
    if l is None and r is None:
        return None
    l_seq = r_seq = text = ""
    rc_l = rc_r = True
    if l is not None and r is not None and l[0] == r[0]:
        l = None
        ci, pos, m, rc_r = r
        seq = contigs[ci][1]
        r_seq, text = (seq[pos + m - 1:] if rc_r else seq[:pos]), seq
        rc_l = first["left"][3]
    else:
        if l is not None:
            ci, pos, m, rc_l = l
            seq = contigs[ci][1]
            l_seq, text = (seq[:pos] if rc_l else seq[pos + m - 1:]), seq
        if r is not None:
            ci, pos, m, rc_r = r
            seq = contigs[ci][1]
            r_seq, text = (seq[pos + m - 1:] if rc_r else seq[:pos - 1]), text + "NN" + seq
    out = (revcomp(l_seq) if rc_l else l_seq) + "NN" + (revcomp(r_seq) if rc_r else r_seq)
    names = tuple(contigs[x[0]][0] if x is not None else "" for x in (first["left"], first["right"]))
    return names[0], names[1], (out if out != "NN" else None), (text if text not in ("", "NN") else None)


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
                if body is None:                          # nothing but 'NN' to report: the file is not written (:527-537)
                    continue
                with open(wf + "velvet_temp/%s/%s" % (gid, fn), "w") as f:
                    f.write(hdr + body + "\n")
                with open(dst, "a") as out:
                    out.write(hdr + body + "\n")
            n += seq is not None
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
