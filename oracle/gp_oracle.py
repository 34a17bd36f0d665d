"""TEST INFRASTRUCTURE — CPU restatement (oracle) of GAPPadder's recruit + local-assembly hot path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (gappadder_amd/) never does.  Every function cites the reference file:line it restates
(paths relative to /root/reference).

Parity status
  * a-1..a-5 (gap scan, flanks, alignment-record tagging, second hop, FASTQ join, library merge):
    PINNED — checked bit-for-bit against outputs of the reference itself (tests/golden/*/expected.tar.gz,
    produced by tests/golden/make_golden.py which runs the 2to3-converted reference in the build container).
  * a-7 (2-bit k-mer layout): PINNED — checked against known answers dumped from the reference's own
    KmerUtils.cpp compiled from /root/reference (tests/golden/kmerutils_kat.json).
  * a-6 (k-mer counting + de-Bruijn assembly): PARITY UNPINNED — the arithmetic lives in third-party KMC
    and Velvet (un-vendored, un-pinned: assemble_gaps.py:96-118 only shells out to them).  The functions
    below restate the documented behaviour for the flags the reference passes and DEFINE the tie-breaks;
    see DESIGN.md "Assembly semantics".
"""
import bisect

# ----------------------------------------------------------------------------- a-1: gaps & flanks


def scan_gaps(seq, min_gap):
    """gnrt_pos_true_seqs.py:17-56.  Gap = first 'N' .. next UPPER-case A/C/G/T; kept if len >= min_gap;
    search resumes at min_pos+2; an N-run reaching the end of the sequence is dropped (:50-51)."""
    out = []
    pos = 0
    while True:
        start = seq.find("N", pos)
        if start == -1:
            break
        pos = start + 1
        ends = [e for e in (seq.find(c, pos) for c in "ACGT") if e != -1]
        if not ends:
            break
        min_pos = min(ends)
        if min_pos - start >= min_gap:
            out.append((start, min_pos))
        pos = min_pos + 2
    return out


def gap_positions(records, min_gap):
    """gnrt_pos_true_seqs.py:12-57 -> lines 'start end len scaffold' in FASTA order."""
    return [(s, e, e - s, name) for name, seq in records for (s, e) in scan_gaps(seq, min_gap)]


def flank_seqs(seq, start, end, flank):
    """gnrt_pos_true_seqs.py:94-99 incl. the Python slice semantics of start-5 < 0."""
    left = seq[0:start - 5] if start < flank else seq[start - flank:start - 5]
    right = seq[end + 5:end + flank]
    return left, right


def gap_ids(fai_names, gaps):
    """'{scaffoldIdx}_{n}', n 1-based per scaffold (gnrt_pos_true_seqs.py:71-83, merge_reads.py:27-41,
    assemble_gaps.py:255-271).  Assumes gap lines grouped by scaffold, as the reference's scan writes them."""
    idx = {n: i for i, n in enumerate(fai_names)}
    out, cnt, pre = [], 1, None
    for (_, _, _, scf) in gaps:
        if scf != pre:
            cnt = 1
        out.append("%d_%d" % (idx[scf], cnt))
        cnt += 1
        pre = scf
    return out


# ----------------------------------------------------------------------------- a-2: alignment-record tagging


def is_clipped(cigar):
    """collect_reads_for_gaps.py:13-26: +2 if last char is S/H, +1 if the first op is S/H."""
    cnt = 2 if cigar[-1] in "SH" else 0
    for ch in cigar:
        if "0" <= ch <= "9":
            continue
        if ch in "SH":
            cnt += 1
        break
    return cnt


def focal_tags(gaps_of_scaffold, pos, dist2, clip_dist):
    """collect_reads_for_gaps.py:31-65 as a closed form: tags [(gapIdx1based, '0c'|'0d'|'1c'|'1d')] at `pos`.
    left window  start-i (i in [0,dist2), start-i >= 0), 'c' when i <= clip_dist;
    right window end+i, same rule.  0-based gap coordinates are compared with the 1-based POS un-shifted."""
    tags = []
    for j, (start, end) in enumerate(gaps_of_scaffold, 1):
        i = start - pos
        if 0 <= i < dist2 and pos >= 0:
            tags.append((j, "0c" if i <= clip_dist else "0d"))
        i = pos - end
        if 0 <= i < dist2:
            tags.append((j, "1c" if i <= clip_dist else "1d"))
    return tags


def tag_record(fields, gaps_of_scaffold, IS, sd, clip_dist, anchor_mapq):
    """collect_reads_for_gaps.py:76-159 (long-IS) / :174-259 (short-IS; switch at IS >= 750, :275).
    fields = first 9 SAM columns (strings).  Returns [(list 'left'|'right', line)]."""
    dist1, dist2 = IS - 3 * sd, IS + 3 * sd
    short_is = IS < 750
    qname, flag, ref, map_pos, mapq_s, cigar, mate_ref, mate_pos = (
        fields[0], int(fields[1]), fields[2], int(fields[3]), fields[4], fields[5], fields[6], int(fields[7]))
    bfirst = (flag & 0x40) != 0
    own, mate = ("left", "right") if bfirst else ("right", "left")
    out = []
    for (j, sflag) in focal_tags(gaps_of_scaffold, map_pos, dist2, clip_dist):
        start, end = gaps_of_scaffold[j - 1]
        gap_len = end - start
        clip_flag = is_clipped(cigar)
        if (sflag == "0c" and clip_flag >= 2) or (sflag == "1c" and clip_flag in (1, 3)):
            out.append((own, "%s %d %s clip" % (qname, j, mapq_s)))
        if (flag & 0x4) == 0 and (flag & 0x8) == 0 and int(mapq_s) >= anchor_mapq:
            line = "%s %d %s discordant %d %s %d %d" % (qname, j, mapq_s, map_pos, mate_ref, mate_pos, gap_len)
            if mate_ref != "=":
                out.append((mate, line))
            else:
                t = abs(int(fields[8]))
                if t >= dist2 or (short_is and t <= dist1):
                    out.append((mate, line))
        elif (flag & 0x4) == 0 and (flag & 0x8) != 0:
            out.append((mate, "%s %d %s unmap" % (qname, j, mapq_s)))
    return out


def collect_reads_for_gaps(sam_lines, gaps, IS, sd, clip_dist=250, anchor_mapq=30):
    """One pass over SAM text for every scaffold that has gaps (run_multi_threads_collect_reads.py:17-33).
    Returns {scaffold: {'left': [lines], 'right': [lines]}} = scaffold_reads_list_all/*."""
    by_scf = {}
    for (s, e, _, scf) in gaps:
        by_scf.setdefault(scf, []).append((s, e))
    out = {scf: {"left": [], "right": []} for scf in by_scf}
    for line in sam_lines:
        f = line.split()
        if f[2] not in by_scf:
            continue
        for side, txt in tag_record(f[:9], by_scf[f[2]], IS, sd, clip_dist, anchor_mapq):
            out[f[2]][side].append(txt)
    return out


# ----------------------------------------------------------------------------- a-3: discordant second hop


def collect_discordant_regions(fai_names, scaffold_lists):
    """run_multi_threads_discordant.py:19-122.  Returns the sorted rows [(mIdx, mPos, sIdx, gIdx)]
    (= discordant_reads_pos.txt.sorted.txt, `sort -k1n -k2n -k3n -k4n`) — duplicates are kept."""
    idx = {n: i for i, n in enumerate(fai_names)}
    rows = []
    for scf in fai_names:
        if scf not in scaffold_lists:
            continue
        for side in ("left", "right"):
            for rec in scaffold_lists[scf][side]:
                f = rec.split()
                if f[3] != "discordant":
                    continue
                m = scf if f[5] == "=" else f[5]
                rows.append((idx[m], int(f[6]), idx[scf], int(f[1])))
    rows.sort()
    return rows


def low_mapq_focal(rows_of_scaffold):
    """collect_discordant_low_mapq_reads.py:4-28 as a closed form.  rows sorted by mPos.
    focal_region[p] = LAST mPos (file order) with mPos-199 <= p <= mPos+299 (later entries overwrite, :21-25);
    m_pos_gaps[mPos] = every 'sIdx_gIdx' of that position, duplicates kept (:15-19)."""
    poss = sorted(set(r[1] for r in rows_of_scaffold))
    groups = {}
    for (_, mpos, sidx, gidx) in rows_of_scaffold:
        groups.setdefault(mpos, []).append("%d_%d" % (sidx, gidx))

    def lookup(p):
        k = bisect.bisect_right(poss, p + 199) - 1     # largest mPos <= p+199
        if k < 0:
            return None
        mpos = poss[k]
        if mpos + 299 < p or p < 0:
            return None
        return mpos
    return lookup, groups


def collect_discordant_low_mapq(sam_lines, fai_names, rows):
    """collect_discordant_low_mapq_reads.py:31-84 for every scaffold (run_multi_threads_discordant.py:125-138).
    Returns {scaffold: {'left': [...], 'right': [...]}} only for scaffolds that own a discordant_temp file
    AND saw at least one MAPQ-0 record (the file pair is opened on the first such record, :55-65)."""
    per = {}
    for r in rows:
        per.setdefault(fai_names[r[0]], []).append(r)
    cache, out = {}, {}
    for line in sam_lines:
        f = line.split()
        if int(f[4]) > 0:
            continue
        scf = f[2]
        if scf not in per:
            continue
        if scf not in cache:
            cache[scf] = low_mapq_focal(per[scf])
            out[scf] = {"left": [], "right": []}
        lookup, groups = cache[scf]
        src = lookup(int(f[3]))
        if src is None:
            continue
        side = "left" if (int(f[1]) & 0x40) else "right"
        for b in groups[src]:
            out[scf][side].append("%s %s %d" % (f[0], b, int(f[4])))
    return out


# ----------------------------------------------------------------------------- a-4: FASTQ join


def read_gap_map(fai_names, scaffold_lists, discordant_lists, side, high_quality=False):
    """run_multi_threads_discordant.py:153-185 (all) / :464-485 (MAPQ == 60 only, no discordant lists).
    {readId: set(gapKey)}."""
    idx = {n: i for i, n in enumerate(fai_names)}
    m = {}
    if not high_quality:
        for scf in fai_names:
            for line in discordant_lists.get(scf, {}).get(side, []):
                f = line.split()
                m.setdefault(f[0], set()).add(f[1])
    for scf in fai_names:
        for line in scaffold_lists.get(scf, {}).get(side, []):
            f = line.split()
            if high_quality and int(f[2]) != 60:
                continue
            m.setdefault(f[0], set()).add("%d_%s" % (idx[scf], f[1]))
    return m


def fastq_records(text):
    lines = text.split("\n")
    for i in range(0, len(lines) - 3, 4):
        yield lines[i], lines[i + 1], lines[i + 3]


def fastq_read_id(header):
    """run_multi_threads_discordant.py:212-214: first whitespace token, text before the first '/', minus '@'."""
    return header.split()[0].split("/")[0][1:].rstrip()


def dispatch_reads(fq_left, fq_right, left_map, right_map):
    """run_multi_threads_discordant.py:209-241, 283-316 -> {gapKey: fastq text}: left-file stream order
    then right-file stream order; header '@{id}_1' / '@{id}_2', bare '+' line."""
    out = {}
    for (text, m, suffix) in ((fq_left, left_map, "_1"), (fq_right, right_map, "_2")):
        for h, s, q in fastq_records(text):
            rid = fastq_read_id(h)
            if rid in m:
                rec = "@%s%s\n%s\n+\n%s\n" % (rid, suffix, s.rstrip(), q.rstrip())
                for key in m[rid]:
                    out.setdefault(key, []).append(rec)
    return {k: "".join(v) for k, v in out.items()}


def collect_library(sam_lines, fq_left, fq_right, fai_names, gaps, IS, sd, clip_dist=250, anchor_mapq=30):
    """main.py:243-259 for one library.  Returns dict with every intermediate the reference writes."""
    lists = collect_reads_for_gaps(sam_lines, gaps, IS, sd, clip_dist, anchor_mapq)
    rows = collect_discordant_regions(fai_names, lists)
    dlists = collect_discordant_low_mapq(sam_lines, fai_names, rows)
    gap_reads = dispatch_reads(fq_left, fq_right, read_gap_map(fai_names, lists, dlists, "left"),
                               read_gap_map(fai_names, lists, dlists, "right"))
    hq = dispatch_reads(fq_left, fq_right, read_gap_map(fai_names, lists, dlists, "left", True),
                        read_gap_map(fai_names, lists, dlists, "right", True))
    return {"lists": lists, "rows": rows, "dlists": dlists, "gap_reads": gap_reads, "gap_reads_high_quality": hq}


def merge_libraries(per_library, ids):
    """merge_reads.py:12-56: per gap id, `cat` the per-library files in library order."""
    out = {}
    for gid in ids:
        parts = [lib[gid] for lib in per_library if gid in lib]
        if parts:
            out[gid] = "".join(parts)
    return out


# ----------------------------------------------------------------------------- a-7: 2-bit k-mers

_CODE = {"C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3}


def base_code(ch):
    """KmerUtils.cpp:25-41: A=00 C=01 G=10 T=11, anything else -> A."""
    return _CODE.get(ch, 0)


def pack_kmer64(seq, pos, k):
    """KmerUtils.cpp:61-69 (FormKmerTypeShortSeg): base i at bits 63-2i,62-2i (MSB-first, left-aligned)."""
    v = 0
    for i in range(k):
        v |= base_code(seq[pos + i]) << (62 - 2 * i)
    return v


def all_kmers64(seq, k):
    """KmerUtils.cpp:90-115 (GetAllKmersFromSeq) by shift-and-set (:72-87)."""
    out = [pack_kmer64(seq, 0, k)]
    for i in range(1, len(seq) - k + 1):
        v = (out[-1] << 2) & 0xFFFFFFFFFFFFFFFF
        v &= ~(3 << (62 - 2 * (k - 1))) & 0xFFFFFFFFFFFFFFFF
        v |= base_code(seq[i + k - 1]) << (62 - 2 * (k - 1))
        out.append(v)
    return out


def kmer_to_string(v, k):
    """KmerUtils.cpp:127-169 (ConvKmerToString)."""
    return "".join("ACGT"[(v >> (62 - 2 * i)) & 3] for i in range(k))


def read_contains_freq_kmers(src_kmers, read, k, thr):
    """KmerUtils.cpp:215-241 (IsReadContainingFreqKmers)."""
    s = set(src_kmers)
    return sum(1 for v in all_kmers64(read, k) if v in s) >= thr


# wide (k <= 64) left-aligned 128-bit value, same layout extended: base i at bits 127-2i,126-2i
def pack_kmer128(seq, pos, k):
    v = 0
    for i in range(k):
        v |= base_code(seq[pos + i]) << (126 - 2 * i)
    return v


def revcomp_kmer128(v, k):
    r = 0
    for i in range(k):
        b = (v >> (126 - 2 * i)) & 3
        r |= (3 - b) << (126 - 2 * (k - 1 - i))
    return r


def canonical_kmers(seq, k, skip_n=True):
    """[(pos, canonical 128-bit value)] of every k-mer of `seq` whose bases are all ACGT (KMC skips k-mers
    with non-ACGT symbols); canonical = min(fwd, revcomp) on the left-aligned value (A<C<G<T)."""
    out = []
    bad = -1
    for i, ch in enumerate(seq):
        if ch not in "ACGT":
            bad = i
        p = i - k + 1
        if p < 0:
            continue
        if skip_n and bad >= p:
            continue
        f = pack_kmer128(seq, p, k)
        out.append((p, min(f, revcomp_kmer128(f, k))))
    return out


# ----------------------------------------------------------------------------- north-star k-mer screen


def flank_kmer_index(flanks, k):
    """{canonical k-mer: set(gap index)} over the left+right flank of every gap (north_star: 'canonical k-mer
    extract of flanking contigs').  flanks = [(left, right)] upper-case; k-mers touching N are skipped."""
    idx = {}
    for g, (l, r) in enumerate(flanks):
        for s in (l, r):
            for _, c in canonical_kmers(s, k):
                idx.setdefault(c, set()).add(g)
    return idx


def screen_reads(reads, flanks, k, min_hits=1):
    """Sorted [(gap, read idx)] such that >= min_hits k-mer POSITIONS of the read are in the gap's flank set."""
    idx = flank_kmer_index(flanks, k)
    hits = []
    for r, seq in enumerate(reads):
        cnt = {}
        for _, c in canonical_kmers(seq, k):
            for g in idx.get(c, ()):
                cnt[g] = cnt.get(g, 0) + 1
        hits.extend((g, r) for g, n in cnt.items() if n >= min_hits)
    hits.sort()
    return hits


# ----------------------------------------------------------------------------- a-6: count + de Bruijn walk


def count_kmers(reads, k, min_count=2, max_count=10000000):
    """KMC as invoked at assemble_gaps.py:96-102: canonical k-mers, k-mers with non-ACGT skipped, counts kept
    when >= min_count (KMC default -ci2), capped at -cs; dump sorted ascending on the canonical form."""
    cnt = {}
    for seq in reads:
        for _, c in canonical_kmers(seq, k):
            cnt[c] = cnt.get(c, 0) + 1
    return sorted((c, min(n, max_count)) for c, n in cnt.items() if n >= min_count)


def kmer128_to_string(v, k):
    return "".join("ACGT"[(v >> (126 - 2 * i)) & 3] for i in range(k))


def _rc(s):
    return s[::-1].translate(str.maketrans("ACGT", "TGCA"))


# Stage 2 (the kv-mer graph, unitigs, Velvet's default tip clipping + bubble popping) is restated in C only:
# oracle/gp_oracle.c `or_assemble_pool2` (bound as oracle.c_oracle.assemble_pool) — the definition the GPU kernel is checked against.


# ----------------------------------------------------------------------------- f-1: flank anchoring -> picked_seqs.fa
# PINNED: tests/golden/pick_kat.json.gz holds flanks.sam texts + contigs and the files the reference's own pick_contigs.py wrote for
# them (make_golden.py::pick_kat runs the 2to3-converted reference with a `bwa` stand-in that serves the prepared SAM text).  What is
# NOT pinnable is which hits bwa itself would report (bwa is absent): `exact_anchor_sam` below DEFINES the stand-in.

BOTH_CLIP, LEFT_CLIP, RIGHT_CLIP, UNCLIP = 1, 2, 3, 4


def clip_type_length(cigar):
    """pick_contigs.py:35-62: (clip type, total length of the M operations)."""
    ops, temp, total_m = [], "", 0
    for ch in cigar:
        if "0" <= ch <= "9":
            temp += ch
        else:
            if ch == "M":
                total_m += int(temp)
            ops.append(ch)
            temp = ""
    first, last = ops[0] in "SH", ops[-1] in "SH"
    if first and last:
        return BOTH_CLIP, total_m
    if first:
        return LEFT_CLIP, total_m
    if last:
        return RIGHT_CLIP, total_m
    return UNCLIP, total_m


def pick_revcomp(s):
    """pick_contigs.py:19-33: complements ACGT of either case to UPPER case, keeps every other symbol."""
    m = {"A": "T", "a": "T", "T": "A", "t": "A", "C": "G", "c": "G", "G": "C", "g": "C"}
    return "".join(m.get(ch, ch) for ch in reversed(s))


def pick_full_from_sam(gid, sam_text, contigs):
    """run_pick_full_constructed_contig, pick_contigs.py:97-358, from the point where flanks.sam exists.  contigs = [(id, seq)]
    in file order.  Returns (picked_seqs.fa text or None, picked_contigs.fa text or None): None = the file is not written.
    Dicts iterate in insertion order (what the 2to3-converted reference does under Python 3; Python 2 iterates in hash order, so
    among contigs of EQUAL span the reference's choice is arbitrary)."""
    hits = {}                                            # rname -> side -> clip type -> (flag, map_length, map_pos)   :97-146
    for line in sam_text.splitlines():
        f = line.split()
        if len(f) < 6:
            continue
        cigar, flag = f[5], int(f[1])
        if cigar == "*":
            continue
        ct, ml = clip_type_length(cigar)
        if ct == BOTH_CLIP:
            continue
        side = f[0].split("_")[-1]
        if side not in ("left", "right"):
            continue
        slot = hits.setdefault(f[2], {}).setdefault(side, {})
        if ct not in slot or ml > slot[ct][1]:           # first hit of a type stays on equal match length (:125-129)
            slot[ct] = (flag, ml, int(f[3]))
    picked = {}
    for rname, h in hits.items():                        # :149-297
        if len(h) != 2:
            continue
        best, sel, b_rc = -1, None, False
        for lt, rt in ((UNCLIP, UNCLIP), (UNCLIP, LEFT_CLIP), (UNCLIP, RIGHT_CLIP), (LEFT_CLIP, UNCLIP), (LEFT_CLIP, RIGHT_CLIP),
                       (RIGHT_CLIP, UNCLIP), (RIGHT_CLIP, LEFT_CLIP)):
            if lt in h["left"] and rt in h["right"]:
                lf, ll, lp = h["left"][lt]
                rf, rl, rp = h["right"][rt]
                if best < ll + rl and (lf & 16) == (rf & 16):
                    best, sel = ll + rl, (lp, rp, ll, rl)
                    if lf & 16:
                        b_rc = True                      # never reset when a later forward pair wins (:176-177 ...)
        if sel is not None:
            picked[rname] = sel + (b_rc,)
    s_picked, min_len = "", -1
    for key, (lp, rp, ll, rl, rc) in picked.items():     # longest span, the first one on ties (:300-321)
        start, end = (rp + rl, lp) if rc else (lp + ll, rp)
        if end - start > min_len:
            s_picked, min_len = key, end - start
    if s_picked == "":
        return None, None
    lp, rp, ll, rl, rc = picked[s_picked]
    seqs, ctgs = "", ""
    for cid, s_ori in contigs:                           # :331-358
        if cid != s_picked:
            continue
        if rc:
            s_gap = pick_revcomp(s_ori[rp + rl - 1:lp])
            s_contig = pick_revcomp(s_ori)
        else:
            s_gap = s_ori[lp + ll - 1:rp]
            s_contig = s_ori
        if s_gap != "":
            seqs += ">" + gid + "_" + cid + "\n" + s_gap + "\n"
        if s_contig != "":
            ctgs += ">" + gid + "_" + cid + "\n" + s_contig + "\n"
    return seqs, ctgs


def pick_extended_from_sam(gid, sam_text, contigs):
    """run_pick_extended_contig, pick_contigs.py:361-539 (reads the flanks.sam the last full pick left behind).  Returns
    (picked_seqs.fa text or None, picked_contigs.fa text or None).  Quirks kept: strand = `flag*16 != 0` (:383-384: ANY non-zero
    flag counts as reverse); on equal match lengths the tie test compares an int with a str (:444, :457) — constant False in
    Python 2 (a TypeError in Python 3: the fixtures avoid such ties), so the first contig stays."""
    m_contigs = dict(contigs)
    hits = {}                                            # qname -> side -> rname -> (map_pos, map_length, b_rc)
    for line in sam_text.splitlines():
        f = line.split()
        if len(f) < 6:
            continue
        cigar, flag = f[5], int(f[1])
        if cigar == "*":
            continue
        b_rc = flag * 16 != 0
        ct, ml = clip_type_length(cigar)
        if ct in (UNCLIP, BOTH_CLIP):
            continue
        qname, rname, side = f[0], f[2], f[0].split("_")[-1]
        if side == "left":
            if (b_rc and ct == LEFT_CLIP) or (not b_rc and ct == RIGHT_CLIP):
                continue
        elif side == "right":
            if (b_rc and ct == RIGHT_CLIP) or (not b_rc and ct == LEFT_CLIP):
                continue
        else:
            continue
        if qname not in hits:
            hits[qname] = {side: {rname: (int(f[3]), ml, b_rc)}}
        else:                                            # (a qname's other side would raise KeyError in the reference: not reachable,
            slot = hits[qname][side]                     #  '{id}_left' only ever carries side 'left')
            if rname not in slot or ml > slot[rname][1]:
                slot[rname] = (int(f[3]), ml, b_rc)
    s_left, s_right = "", ""
    for qname, h in hits.items():                        # :430-459
        for side in ("left", "right"):
            best = 0
            for rname, (_, ml, _) in h.get(side, {}).items():
                if ml > best:
                    best = ml
                    if side == "left":
                        s_left = rname
                    else:
                        s_right = rname
    l_seq = r_seq = s_contig = ""
    rc_l = rc_r = True
    if s_left != "" and s_left == s_right:               # :468-491
        lp, ll, rc_l = hits[gid + "_left"]["left"][s_left]
        rp, rl, rc_r = hits[gid + "_right"]["right"][s_right]
        if ll > rl:
            l_seq = m_contigs[s_left][0:lp] if rc_l else m_contigs[s_left][lp + ll - 1:]
            s_contig = m_contigs[s_left]
        else:
            r_seq = m_contigs[s_right][0:rp] if not rc_r else m_contigs[s_right][rp + rl - 1:]
            s_contig = m_contigs[s_right]
    else:                                                # :492-515
        if s_left != "":
            lp, ll, rc_l = hits[gid + "_left"]["left"][s_left]
            l_seq = m_contigs[s_left][0:lp] if rc_l else m_contigs[s_left][lp + ll - 1:]
            s_contig = m_contigs[s_left]
        if s_right != "":
            rp, rl, rc_r = hits[gid + "_right"]["right"][s_right]
            r_seq = m_contigs[s_right][0:rp - 1] if not rc_r else m_contigs[s_right][rp + rl - 1:]
            s_contig = s_contig + "NN" + m_contigs[s_right]
    s_seq = (pick_revcomp(l_seq) if rc_l else l_seq) + "NN" + (pick_revcomp(r_seq) if rc_r else r_seq)   # :517-525
    hdr = ">" + gid + "_" + s_left + "_" + s_right + "_extended\n"
    return (hdr + s_seq + "\n" if s_seq not in ("", "NN") else None,
            hdr + s_contig + "\n" if s_contig not in ("", "NN") else None)


def exact_anchor_sam(gid, contigs, left_flank, right_flank, score):
    """DEFINITION (bwa is absent: parity of this step unpinned) of the stand-in for `bwa mem -T {score} -a contigs.fa flanks.fa`
    (pick_contigs.py:79-86): the only alignments reported are EXACT matches of the `score` flank bases next to the gap — the last
    `score` bases of the left flank, the first `score` of the right flank, upper-case ACGT only — never extended.  Per contig, in
    file order: forward strand left (its LEFTMOST occurrence), forward right (RIGHTMOST), reverse strand left (the occurrence that
    is leftmost in the reverse-complemented contig), reverse right (rightmost there); flag 0 / 16 (no secondary flags), the rest of
    the flank soft-clipped, reverse-strand lines in the contig's coordinates with the CIGAR reversed, as SAM has them."""
    a = int(score)
    if len(left_flank) < a or len(right_flank) < a:
        return ""
    la, ra = left_flank[len(left_flank) - a:], right_flank[:a]
    if any(c not in "ACGT" for c in la + ra):
        return ""
    lclip, rclip = len(left_flank) - a, len(right_flank) - a
    out = []
    for name, seq in contigs:
        n = len(seq)
        rcs = _rc(seq) if all(c in "ACGT" for c in seq) else pick_revcomp(seq)
        for flag, s in ((0, seq), (16, rcs)):
            i, j = s.find(la), s.rfind(ra)
            for qn, p, clip, clip_first in (("left", i, lclip, True), ("right", j, rclip, False)):
                if p < 0:
                    continue
                pos = p + 1 if flag == 0 else n - p - a + 1
                parts = ["%dM" % a]
                if clip:
                    parts = ["%dS" % clip] + parts if clip_first == (flag == 0) else parts + ["%dS" % clip]
                out.append("%s_%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*" % (gid, qn, flag, name, pos, "".join(parts)))
    return "".join(l + "\n" for l in out)


def pick_gap(gid, contigs, left_flank, right_flank, score):
    """The picker as this build runs it: the reference's selection on the stand-in's hits.  Returns (seqs text, contigs text)."""
    return pick_full_from_sam(gid, exact_anchor_sam(gid, contigs, left_flank, right_flank, score), contigs)


def pick_gap_extended(gid, contigs, left_flank, right_flank, score):
    return pick_extended_from_sam(gid, exact_anchor_sam(gid, contigs, left_flank, right_flank, score), contigs)


# ----------------------------------------------------------------------------- f-3: ContigsMerger, from the overlap edges to the merged contigs
# PINNED: tests/golden/merger_kat.json.gz = contig sets and what the reference's own ContigsMerger (built from its sources into
# oracle/_ref/contigs_merger, run with -t 1) printed for them.  The reference orders several containers by POINTER value (sets of
# graph nodes, GraphUtils.cpp:713-741, 1258-1344; with -t > 1 also the edge lists by thread timing); this restatement uses the
# allocation order — node index, insertion order — which is what the single-threaded reference does under glibc, and the tests
# compare the merged sequences as the reference lists them.

def merger_edges(contigs, params, k=10):
    """CompactVer3 up to addEdges (ContigsCompactor.cpp:773-870): nodes [c0, rc(c0), c1, ...]; every feasible pair (i <= j, the
    prefilter's order) is evaluated; an overlap of class 2 without containment is an edge i -> j (mode 12) or j -> i (mode 21) of
    length -overlap (threadMergeContigV2 :624-690, addEdges :724-770).  Returns (nodes, adj) with adj[v] = [(w, length)] in
    insertion order."""
    from oracle import c_oracle as CO
    nodes = CO.merger_nodes(contigs)
    adj = [[] for _ in nodes]
    for i, j in CO.quick_check(contigs, k):
        r = CO.overlap_evaluate(nodes[i], nodes[j], params)
        if r["res"] == 2 and not r["containment"]:
            if r["first_goes_first"]:
                adj[i].append((j, -float(r["overlap"])))
            else:
                adj[j].append((i, -float(r["overlap"])))
    return nodes, adj


def merger_scc(adj):
    """AbstractGraph::SCC (GraphUtils.cpp:1028-1178): Tarjan from node 0 upwards, neighbours in edge order; the components come
    out in reverse finishing order = a topological order of the condensation.  Each component as a sorted list (a std::set of
    node pointers)."""
    n = len(adj)
    index, low, on = [-1] * n, [0] * n, [False] * n
    stack, out, nxt = [], [], [1]

    def visit(v):                      # (recursive like the reference; contig sets are small)
        index[v] = low[v] = nxt[0]
        nxt[0] += 1
        stack.append(v)
        on[v] = True
        for w, _ in adj[v]:
            if index[w] < 0:
                visit(w)
                low[v] = min(low[v], low[w])
            elif on[w]:
                low[v] = min(low[v], index[w])
        if low[v] == index[v]:
            comp = []
            while True:
                w = stack.pop()
                on[w] = False
                comp.append(w)
                if w == v:
                    break
            out.append(sorted(comp))
    import sys
    sys.setrecursionlimit(max(sys.getrecursionlimit(), 4 * n + 100))
    for v in range(n):
        if index[v] < 0:
            visit(v)
    return out[::-1]


def merger_path_ends(sccs, adj, start):
    """FindSimplePathsTopSortStart (GraphUtils.cpp:1258-1344): candidate path roots (start) / ends (not start)."""
    comp_of = {}
    for ci, comp in enumerate(sccs):
        for v in comp:
            comp_of[v] = ci
    cand = set(comp_of)
    for comp in sccs:
        for v in comp:
            for w, _ in adj[v]:
                if comp_of[w] != comp_of[v]:
                    if start:
                        cand.discard(w)
                    else:
                        cand.discard(v)
                        break
    for comp in sccs:
        if len(comp) > 1:
            all_in = all(v in cand for v in comp)
            keep = comp[0] if start else comp[-1]
            for v in comp:
                if v != keep:
                    cand.discard(v)
            if not all_in:
                cand.discard(keep)
    return sorted(cand)


def merger_paths(adj, max_per_root=20):
    """FindSimplePathsTopSort + FindSimplePathsTopSortFrom (GraphUtils.cpp:625-859): per candidate root a shortest-path DP over
    the topologically listed nodes (edges that point backwards in the list are ignored; lengths are -overlap, so the path with the
    largest total overlap wins; a later path must be strictly shorter to replace an earlier one), the paths to the candidate ends,
    the longest max_per_root + 1 of them by node count.  Returns the set of paths as sorted tuples (std::set order)."""
    sccs = merger_scc(adj)
    order = [v for comp in sccs for v in comp]
    rank = {v: i for i, v in enumerate(order)}
    roots, ends = merger_path_ends(sccs, adj, True), merger_path_ends(sccs, adj, False)
    found = set()
    for root in roots:
        dist = [None] * len(order)
        path = [None] * len(order)
        dist[rank[root]], path[rank[root]] = 0.0, (root,)
        for i in range(rank[root], len(order)):
            if dist[i] is None:
                continue
            v = order[i]
            for w, _ in adj[v]:
                pw = rank[w]
                if pw < i:
                    continue
                length = next(l for x, l in adj[v] if x == w)           # GetEdgeTo: the first edge to that node
                if dist[pw] is None or dist[i] + length < dist[pw]:
                    dist[pw], path[pw] = dist[i] + length, path[i] + (w,)
        cur = []
        for e in ends:
            if dist[rank[e]] is not None and path[rank[e]] not in cur:
                cur.append(path[rank[e]])
        by_len = sorted(range(len(cur)), key=lambda q: (-len(cur[q]), q))       # longest first, then as found
        for n_out, q in enumerate(by_len):
            found.add(cur[q])
            if n_out + 1 > max_per_root:
                break
    return sorted(found)


def merger_merge_path(nodes, path, params):
    """FormMergedSeqFromPath (ContigsCompactor.cpp:1456-1520): the running merged string against the next node, Evaluate in its
    relaxed mode, joined by SetMergedStringConcat (:108-153)."""
    from oracle import c_oracle as CO
    s1 = nodes[path[0]]
    for v in path[1:]:
        s2 = nodes[v]
        r = CO.overlap_evaluate(s1, s2, params, relax=True)
        n1, n2, re_, ce, nc = len(s1), len(s2), r["row_end"], r["col_end"], r["nclip"]
        if r["contained"] and re_ + nc == n1 and n1 < n2:
            s1 = s2
        elif r["contained"] and ce + nc == n2 and n2 < n1:
            pass
        elif re_ + nc == n1:
            s1 = s1[:n1 - nc] + s2[ce:]
        else:
            s1 = s2[:n2 - nc] + s1[re_:]
    return s1


def merger_new_contigs(contigs, params, k=10, max_per_root=20):
    """ContigsMerger's NEW_CONTIG_MERGE_n records for one contig set (CompactVer3, ContigsCompactor.cpp:773-983): [(path as node
    indices, merged sequence)] in output order; of a path and its reverse-complement twin the one listed first stays
    (RemoveDupRevCompPaths :1422-1454); single-node paths produce nothing."""
    nodes, adj = merger_edges(contigs, params, k)
    paths = merger_paths(adj, max_per_root)
    kept = []
    for i, p in enumerate(paths):
        twin = tuple(v ^ 1 for v in reversed(p))
        if twin not in paths[:i]:
            kept.append(p)
    return [(p, merger_merge_path(nodes, p, params)) for p in kept if len(p) > 1]
