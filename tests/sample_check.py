"""The full-size checker: does what the GPU left in its output buffers equal the oracle's answers on the sample the oracle
computed?  bench.py's `cpu_baseline` leg calls these after the timed region (`parity_on_sample`), tests/test_gpu_scale.py
asserts on that at every BASELINE configuration's full size — and tests/test_sample_check.py plants a wrong hit, a wrong contig
base and a wrong pick in such buffers and requires every one of these functions to go red (the comparison
code must not be able to hide a bug of its own).  Test infrastructure: the product never imports this."""
import numpy as np

HIT = np.dtype([("gap", "<u4"), ("read", "<u4")])
TAGHIT = np.dtype([("rec", "<u4"), ("gap", "<u4"), ("kind", "<u2"), ("to_mate", "<u2")])


def stripes(n_items, n_sample, strides=(), shard_worlds=(2, 4, 8), seed=1, n_random=4):
    """Where the full-size sample lies: disjoint ranges [(first, n)] of ITEMS (read pairs) that together hold about n_sample items and
    cover the whole array instead of its prefix: the first and the last items, one stripe across every place where a
    byte offset `item * stride` passes 4 GiB (strides = bytes per item of the arrays the kernels index: packed reads, 32-byte records,
    8-byte keys), one across every boundary between the shards of a `shard_worlds`-rank run (gappadder_amd/sharding.shard_range), and
    n_random more at seeded places.  n_sample >= n_items: the whole array, one range."""
    n_items, n_sample = int(n_items), int(n_sample)
    if n_sample >= n_items:
        return [(0, n_items)] if n_items else []
    centres = []
    for st in strides:
        if n_items * int(st) > (1 << 32):
            centres.append((1 << 32) // int(st))
    for w in shard_worlds:
        base, extra = divmod(n_items, w)
        centres += [r * base + min(r, extra) for r in range(1, w)]
    rng = np.random.RandomState(seed)
    centres += [int(x) for x in rng.randint(0, n_items, n_random)]
    centres = sorted(set(centres))
    w = max(1, n_sample // (len(centres) + 2))
    spans = [(0, min(w, n_items)), (max(0, n_items - w), n_items)] + [(max(0, c - w // 2), min(n_items, c - w // 2 + w)) for c in centres]
    spans.sort()
    merged = [list(spans[0])]
    for a, b in spans[1:]:
        if a <= merged[-1][1]:
            merged[-1][1] = max(merged[-1][1], b)
        else:
            merged.append([a, b])
    return [(a, b - a) for a, b in merged if b > a]


def chunks_of(ranges, max_items):
    """ranges cut into groups of at most max_items items (one oracle call per group: bounds the host memory of a complete comparison)."""
    out, cur, n_cur = [], [], 0
    for a, n in ranges:
        while n:
            take = min(n, max_items - n_cur)
            cur.append((a, take))
            a, n, n_cur = a + take, n - take, n_cur + take
            if n_cur == max_items:
                out.append(cur)
                cur, n_cur = [], 0
    if cur:
        out.append(cur)
    return out


def to_sample(ids, sample):
    """sample: an int n (the prefix [0, n)) or ranges [(first, n)] (sorted, disjoint) whose members the oracle saw back to back.
    -> (mask of the ids inside the sample, their index in the oracle's numbering)."""
    ids = np.asarray(ids).astype(np.int64)
    if isinstance(sample, (int, np.integer)):
        m = ids < int(sample)
        return m, ids[m]
    starts = np.array([a for a, _ in sample], dtype=np.int64)
    lens = np.array([n for _, n in sample], dtype=np.int64)
    assert (starts[1:] >= (starts + lens)[:-1]).all(), "sample ranges must be sorted and disjoint"
    base = np.concatenate([[0], np.cumsum(lens)[:-1]])
    j = np.searchsorted(starts, ids, side="right") - 1
    jj = np.maximum(j, 0)
    m = (j >= 0) & (ids < starts[jj] + lens[jj])
    return m, (ids - starts[jj] + base[jj])[m]


def hits_equal(gpu_hits, oracle_hits, sample):
    """Screen hits (gap, read) of the sampled reads: the GPU's list (any order, all reads, global read numbers) against the oracle's
    (sample only, reads numbered as the oracle saw them).  sample: prefix length or read ranges (to_sample)."""
    m, idx = to_sample(gpu_hits["read"], sample)
    sub = np.ascontiguousarray(gpu_hits[m]).astype(HIT)
    sub["read"] = idx
    sub = np.sort(sub, order=["gap", "read"])
    want = np.sort(np.ascontiguousarray(oracle_hits).astype(HIT), order=["gap", "read"])
    return len(sub) == len(want) and sub.tobytes() == want.tobytes()


def taghits_equal(gpu_tags, oracle_tags, sample):
    """Tagger hits of the sampled records (prefix length or record ranges)."""
    order = ["rec", "gap", "kind", "to_mate"]
    m, idx = to_sample(gpu_tags["rec"], sample)
    sub = np.ascontiguousarray(gpu_tags[m]).astype(TAGHIT)
    sub["rec"] = idx
    sub = np.sort(sub, order=order)
    want = np.sort(np.ascontiguousarray(oracle_tags).astype(TAGHIT), order=order)
    return len(sub) == len(want) and sub.tobytes() == want.tobytes()


def sample_gaps(n_gaps, n_sample, seed=1):
    """The gaps whose assembly and pick meet the oracle at full size: all of them when n_sample >= n_gaps, else the first, the last
    and one at a seeded place inside each of n_sample - 2 equal parts of the gap list (every region of every scaffold range is drawn from)."""
    n_gaps, n_sample = int(n_gaps), int(n_sample)
    if n_sample >= n_gaps:
        return list(range(n_gaps))
    rng = np.random.RandomState(seed)
    m = max(1, n_sample - 2)
    edges = np.linspace(0, n_gaps, m + 1).astype(np.int64)
    picks = [int(a + rng.randint(0, max(1, b - a))) for a, b in zip(edges[:-1], edges[1:])]
    return sorted(set([0, n_gaps - 1] + picks))


def _contig_text(ctg, seq, i):
    return seq[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])].decode()


def _gap_list(gaps):
    return list(range(gaps)) if isinstance(gaps, (int, np.integer)) else [int(g) for g in gaps]


def contigs_equal(ctg, seq, expected, kk, gaps):
    """The device's contig records of the sampled gaps (any order) against expected[j][i] = the oracle's [(sequence, n_nodes,
    cov_sum)] of the j-th sampled gap at kk[i].  gaps: n (the gaps [0, n)) or a list of gap numbers."""
    for j, g in enumerate(_gap_list(gaps)):
        for (k, kv), e in zip(kk, expected[j]):
            rows = np.nonzero((ctg["gap"] == g) & (ctg["k"] == k) & (ctg["kv"] == kv))[0]
            mine = sorted((_contig_text(ctg, seq, i), int(ctg[i]["n_nodes"]), int(ctg[i]["cov_sum"])) for i in rows)
            if mine != sorted(e):
                return False
    return True


def expected_pick_word(ctg, seq, g, flanks, kk):
    """The word gf_pick_anchored2_dev must leave for gap g — anchor length << 56 | span + 1 << 32 | (0x7FFFFFFF - contig) << 1 | strand —
    from oracle/gp_oracle.py::pick_gap = the reference's selection (pick_contigs.py:97-358, pinned on its own answers) on the
    exact-anchor stand-in's hits; scores 30 then 15 (assemble_gaps.py:336, 365).  The picker sees the gap's contigs in the order the
    device listed them (ties between equal spans go to the earlier contig)."""
    from oracle import gp_oracle as PO
    own = sorted(int(i) for (k, kv) in kk for i in np.nonzero((ctg["gap"] == g) & (ctg["k"] == k) & (ctg["kv"] == kv))[0])
    # the contigs a merge round appended for the gap (k = kv = 0): picked from only when the gap's own contigs leave it open — the step
    # merges the open gaps' contigs and picks a second time over the merged ones (Pipeline.assemble, assemble_gaps.py:301-306, 335-339)
    merged = sorted(int(i) for i in np.nonzero((ctg["gap"] == g) & (ctg["k"] == 0) & (ctg["kv"] == 0))[0])
    for idx in (own, merged):
        want = [("c%d" % i, _contig_text(ctg, seq, i)) for i in idx]
        for a_len in (30, 15):
            seqs, ctgs_txt = PO.pick_gap("0_1", want, flanks[g][0], flanks[g][1], a_len)
            if seqs:
                hdr, body = seqs.split("\n")[:2]
                ci = idx[[n for n, _ in want].index(hdr[len(">0_1_"):])]
                rev = int(ctgs_txt.split("\n")[1] != dict(want)["c%d" % ci])
                return (a_len << 56) | (len(body) << 32) | ((0x7FFFFFFF - ci) << 1) | rev
    return 0


def expected_merged_contigs(own, max_set=128):
    """What the step's merge round must append for a gap whose own contigs (all (k, kv), record order) the first pick left open:
    exact-containment dedup (a contig that occurs, on either strand, inside another one goes; of identical ones the first stays), then —
    for 2 .. max_set contigs left — ContigsMerger's NEW_CONTIG_MERGE sequences by oracle/gp_oracle.py::merger_new_contigs (prefilter,
    overlap evaluation, path search and merged strings restated from the reference and pinned on its binary's answers) over the
    contigs of 30 .. 8190 bases."""
    from oracle import c_oracle as CO
    from oracle import gp_oracle as PO
    comp = str.maketrans("ACGT", "TGCA")
    if not 2 <= len(own) <= 1024:
        return []
    order = sorted(range(len(own)), key=lambda i: (-len(own[i]), i))
    kept = []
    for i in order:
        q = own[i]
        if not any(q in own[j] or q in own[j].translate(comp)[::-1] for j in kept):
            kept.append(i)
    kept.sort()
    if not 2 <= len(kept) <= max_set:
        return []
    nodes = [own[i] for i in kept if 30 <= len(own[i]) <= 8190]
    return [s_ for _, s_ in PO.merger_new_contigs(nodes, CO.GAPPADDER_OVL)] if len(nodes) >= 2 else []


def merged_equal(ctg, seq, gaps, n0, kk=None):
    """The merged contigs (k = kv = 0, records from n0 on) the device appended for the listed gaps against expected_merged_contigs of
    the gaps' own contigs (records before n0) in the order of the gap's contigs.fa (assemble_gaps.py:124-135): the (k, kv) pairs in list
    order, inside a pair by (length descending, sequence); kk = None: record order."""
    for g in _gap_list(gaps):
        rows = np.nonzero(ctg["gap"] == g)[0]
        mine = [i for i in rows if i < n0]
        if kk is not None:
            pair = {(int(k), int(kv)): q for q, (k, kv) in enumerate(kk)}
            mine.sort(key=lambda i: (pair.get((int(ctg[i]["k"]), int(ctg[i]["kv"])), 0xFFFF), -int(ctg[i]["length"]), _contig_text(ctg, seq, i), i))
        own = [_contig_text(ctg, seq, i) for i in mine]
        got = [_contig_text(ctg, seq, i) for i in rows if i >= n0]
        if got != expected_merged_contigs(own):
            return False
    return True


def picks_equal(ctg, seq, best, flanks, kk, gaps):
    return all(expected_pick_word(ctg, seq, g, flanks, kk) == int(best[g]) for g in _gap_list(gaps))
