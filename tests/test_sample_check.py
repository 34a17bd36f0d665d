"""Mutation tests of the full-size checker (tests/sample_check.py): bench.py's `parity_on_sample` and
`test_full_size_config_sample_parity` rest on these comparisons, so each of them must go red when the "GPU" buffers hold a wrong
hit, a wrong contig base or a wrong pick.  The buffers here are built from the oracle's own answers (no GPU): first the checker
must accept them in any order, then reject every planted defect."""
import numpy as np

from gappadder_amd import _lib as B
from oracle import c_oracle as CO
import sample_check as SC
import synth_small as S

KK = [(31, 29), (41, 39)]


def _case():
    c = S.small_case(seed=12, n_pairs=6000)
    hits = CO.screen_reads(c["reads_blob"], c["L"], c["flanks"], 31)
    tags = CO.tag_alignments(c["recs"], c["gaps"], 300, 30)
    return c, hits, tags


def _device_like_contigs(c, hits, rng):
    """Contig records + sequence blob as the device leaves them: one list for all gaps and (k, kv), in arbitrary order."""
    L = c["L"]
    recs, expected = [], []
    for g in range(len(c["gaps"])):
        ids = sorted(set(int(h["read"]) for h in hits if h["gap"] == g) | set(int(h["read"]) ^ 1 for h in hits if h["gap"] == g))
        pool = b"".join(c["reads_blob"][i * L:(i + 1) * L] for i in ids)
        expected.append([CO.assemble_pool(pool, L, k, kv) for k, kv in KK])
        for (k, kv), e in zip(KK, expected[-1]):
            recs += [(g, k, kv, s, n, cov) for s, n, cov in e]
    order = rng.permutation(len(recs))
    ctg = np.zeros(len(recs), dtype=B.CONTIG)
    seq = bytearray()
    for j, i in enumerate(order):
        g, k, kv, s, n, cov = recs[i]
        ctg[j] = (g, k, kv, n, len(s), cov, 0, len(seq))
        seq += s.encode()
    return ctg, bytes(seq), expected


def test_hit_checks_accept_any_order_and_catch_a_wrong_missing_or_extra_hit():
    c, hits, tags = _case()
    rng = np.random.RandomState(1)
    n_s = 8000                                                  # the oracle looked at reads [0, n_s) only
    want = hits[hits["read"] < n_s]
    gpu = hits[rng.permutation(len(hits))].astype(B.HIT)        # the GPU reports all reads, in any order
    assert len(want) > 50 and len(gpu) > len(want)
    assert SC.hits_equal(gpu, want, n_s)
    i = int(np.nonzero(gpu["read"] < n_s)[0][0])
    bad = gpu.copy(); bad["gap"][i] ^= 1
    assert not SC.hits_equal(bad, want, n_s)                    # a hit on the wrong gap
    assert not SC.hits_equal(np.delete(gpu, i), want, n_s)      # a missing hit
    assert not SC.hits_equal(np.concatenate([gpu, gpu[i:i + 1]]), want, n_s)   # a hit reported twice
    extra = gpu[i:i + 1].copy(); extra["read"] = n_s - 1; extra["gap"] = 0
    if not ((want["read"] == n_s - 1) & (want["gap"] == 0)).any():
        assert not SC.hits_equal(np.concatenate([gpu, extra]), want, n_s)      # a read that is no hit
    beyond = gpu[i:i + 1].copy(); beyond["read"] = n_s + 5
    assert SC.hits_equal(np.concatenate([gpu, beyond]), want, n_s)             # (outside the sample: not the checker's business)
    wt = tags[tags["rec"] < n_s]
    gt = tags[rng.permutation(len(tags))].astype(B.TAGHIT)
    assert len(wt) > 20 and SC.taghits_equal(gt, wt, n_s)
    j = int(np.nonzero(gt["rec"] < n_s)[0][0])
    for field, delta in (("gap", 1), ("kind", 1), ("to_mate", 1), ("rec", 1)):
        bad = gt.copy(); bad[field][j] ^= delta
        assert not SC.taghits_equal(bad, wt, n_s), field
    assert not SC.taghits_equal(np.delete(gt, j), wt, n_s)


def test_contig_check_catches_a_wrong_base_count_or_missing_contig():
    c, hits, _ = _case()
    ctg, seq, expected = _device_like_contigs(c, hits, np.random.RandomState(2))
    n_g = len(c["gaps"])
    assert len(ctg) > 10 and SC.contigs_equal(ctg, seq, expected, KK, n_g)
    mid = int(ctg[0]["seq_off"]) + int(ctg[0]["length"]) // 2
    wrong = bytearray(seq); wrong[mid] = ord("A") if wrong[mid] != ord("A") else ord("C")
    assert not SC.contigs_equal(ctg, bytes(wrong), expected, KK, n_g)          # ONE wrong contig base
    for field in ("n_nodes", "cov_sum", "length"):
        bad = ctg.copy(); bad[field][0] -= 1
        assert not SC.contigs_equal(bad, seq, expected, KK, n_g), field
    assert not SC.contigs_equal(ctg[1:], seq, expected, KK, n_g)               # a contig the device did not report
    assert not SC.contigs_equal(np.concatenate([ctg, ctg[:1]]), seq, expected, KK, n_g)   # ... or reported twice
    bad = ctg.copy(); bad["k"][0], bad["kv"][0] = (41, 39) if int(ctg[0]["k"]) == 31 else (31, 29)
    assert not SC.contigs_equal(bad, seq, expected, KK, n_g)                   # filed under the wrong (k, kv)


def test_pick_check_knows_the_reference_slice_and_catches_a_wrong_pick():
    rng = np.random.RandomState(3)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.randint(0, 4, n)].tobytes().decode()
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[x] for x in reversed(s))
    flanks = [(rnd(295), rnd(295)) for _ in range(3)]
    fill = [rnd(120), rnd(77), rnd(60)]
    # gap 0: closed on the forward strand by contig 1; gap 1: closed by a reverse-complemented contig; gap 2: both anchors, but on
    # different contigs (open)
    texts = [(0, rnd(200)),
             (0, rnd(30) + flanks[0][0][-60:] + fill[0] + flanks[0][1][:60] + rnd(25)),
             (1, rc(flanks[1][0][-45:] + fill[1] + flanks[1][1][:45])),
             (2, rnd(10) + flanks[2][0][-50:] + rnd(40)),
             (2, rnd(40) + flanks[2][1][:50])]
    ctg = np.zeros(len(texts), dtype=B.CONTIG)
    seq = bytearray()
    for i, (g, s) in enumerate(texts):
        ctg[i] = (g, 31, 29, len(s) - 28, len(s), 100, 0, len(seq))
        seq += s.encode()
    seq = bytes(seq)
    kk = [(31, 29)]
    # known answers, by hand: the reference's slice contig[leftPos + leftM - 1 : rightPos] keeps ONE anchor base (pick_contigs.py:341-349),
    # so the span field is len(fill) + 1; anchor length 30; contig index; strand
    want = [(30 << 56) | ((len(fill[0]) + 1) << 32) | ((0x7FFFFFFF - 1) << 1) | 0,
            (30 << 56) | ((len(fill[1]) + 1) << 32) | ((0x7FFFFFFF - 2) << 1) | 1,
            0]
    got = [SC.expected_pick_word(ctg, seq, g, flanks, kk) for g in range(3)]
    assert got == want
    best = np.array(want, dtype=np.uint64)
    assert SC.picks_equal(ctg, seq, best, flanks, kk, 3)
    for g, word in ((0, want[0] ^ 1),                                       # wrong strand
                    (0, (want[0] & ~(0xFFFFFFFF << 1)) | ((0x7FFFFFFF - 0) << 1)),     # the wrong contig
                    (0, want[0] + (1 << 32)),                               # span off by one
                    (0, (15 << 56) | (want[0] & ((1 << 56) - 1))),          # the weaker anchor score
                    (1, 0),                                                 # a closed gap reported open
                    (2, want[0])):                                          # an open gap reported closed
        bad = best.copy(); bad[g] = word
        assert not SC.picks_equal(ctg, seq, bad, flanks, kk, 3), (g, hex(word))
    # a wrong base inside the anchor of the winning contig: the oracle no longer closes gap 0, the unchanged device word is wrong
    at = int(ctg[1]["seq_off"]) + 30 + 60 - 3
    wrong = bytearray(seq); wrong[at] = ord("A") if wrong[at] != ord("A") else ord("C")
    assert not SC.picks_equal(ctg, bytes(wrong), best, flanks, kk, 3)


def test_stripes_cover_the_ends_the_4gib_crossings_and_the_shard_boundaries():
    from gappadder_amd.sharding import shard_range
    n_pairs, rb = 450_000_000, 38            # C4: 900 M reads
    rg = SC.stripes(n_pairs, 2_000_000, strides=(2 * rb, 64, 16), seed=5)
    starts, ends = [a for a, _ in rg], [a + n for a, n in rg]
    assert all(e <= s for e, s in zip(ends[:-1], starts[1:])) and starts[0] == 0 and ends[-1] == n_pairs      # sorted, disjoint, first + last
    assert 1_800_000 <= sum(n for _, n in rg) <= 2_000_000
    inside = lambda p: any(a <= p < a + n for a, n in rg)
    for stride in (2 * rb, 64, 16):          # the pair whose bytes straddle offset 4 GiB of each array, and its neighbours
        p = (1 << 32) // stride
        assert inside(p - 1) and inside(p) and inside(p + 1), stride
    for world in (2, 4, 8):                  # both sides of every shard boundary (sharding.shard_range)
        for r in range(1, world):
            b = shard_range(n_pairs, r, world)[0]
            assert inside(b - 1) and inside(b), (world, r)
    assert SC.stripes(1000, 5000) == [(0, 1000)] and SC.stripes(1000, 1000) == [(0, 1000)]                   # complete
    small = SC.stripes(100_000, 10_000, strides=(76,), seed=1)      # no 4-GiB crossing: ends + shard boundaries + seeded places
    assert small[0][0] == 0 and small[-1][0] + small[-1][1] == 100_000 and len(small) >= 8
    assert [sum(n for _, n in g) for g in SC.chunks_of([(0, 10), (50, 25), (100, 7)], 16)] == [16, 16, 10]
    assert [x for g in SC.chunks_of([(0, 10), (50, 25), (100, 7)], 16) for x in g] == [(0, 10), (50, 6), (56, 16), (72, 3), (100, 7)]
    gs = SC.sample_gaps(19840, 256, seed=3)
    assert gs[0] == 0 and gs[-1] == 19839 and 250 <= len(gs) <= 256 and max(np.diff(gs)) < 2 * 19840 // 254 + 2
    assert SC.sample_gaps(200, 256) == list(range(200))


def test_range_samples_renumber_the_gpu_ids_and_catch_defects_in_any_stripe():
    c, hits, tags = _case()
    L, rng = c["L"], np.random.RandomState(4)
    n_reads = len(c["reads_blob"]) // L
    ranges = [(0, 1500), (4000, 2000), (n_reads - 1000, 1000)]                 # reads; the oracle sees them back to back
    blob = b"".join(c["reads_blob"][a * L:(a + n) * L] for a, n in ranges)
    want = CO.screen_reads(blob, L, c["flanks"], 31)
    gpu = hits[rng.permutation(len(hits))].astype(B.HIT)
    assert len(want) > 20 and SC.hits_equal(gpu, want, ranges)
    for a, n in ranges:                                                        # a defect in EVERY stripe is seen, first and last read included
        inside = np.nonzero((gpu["read"] >= a) & (gpu["read"] < a + n))[0]
        assert len(inside), (a, n)
        assert not SC.hits_equal(np.delete(gpu, inside[0]), want, ranges)
        for rd in (a, a + n - 1):
            extra = gpu[:1].copy(); extra["read"], extra["gap"] = rd, 1
            if not ((gpu["read"] == rd) & (gpu["gap"] == 1)).any():
                assert not SC.hits_equal(np.concatenate([gpu, extra]), want, ranges)
    outside = gpu[:1].copy(); outside["read"] = 3000                           # between the stripes: not the checker's business
    assert SC.hits_equal(np.concatenate([gpu, outside]), want, ranges)
    recs = np.concatenate([c["recs"][a:a + n] for a, n in ranges])
    wt = CO.tag_alignments(recs, c["gaps"], 300, 30)
    gt = tags[rng.permutation(len(tags))].astype(B.TAGHIT)
    assert len(wt) > 10 and SC.taghits_equal(gt, wt, ranges)
    m, _ = SC.to_sample(gt["rec"], ranges)
    j = int(np.nonzero(m)[0][-1])
    bad = gt.copy(); bad["gap"][j] ^= 1
    assert not SC.taghits_equal(bad, wt, ranges)
    # contigs / picks of a gap LIST: expected[j] belongs to gaps[j]
    ctg, seq, expected = _device_like_contigs(c, hits, np.random.RandomState(2))
    sel = [g for g in range(len(c["gaps"])) if g % 2 == 1]
    assert SC.contigs_equal(ctg, seq, [expected[g] for g in sel], KK, sel)
    bad = ctg.copy(); i = int(np.nonzero(ctg["gap"] == sel[-1])[0][0]); bad["cov_sum"][i] += 1
    assert not SC.contigs_equal(bad, seq, [expected[g] for g in sel], KK, sel)
    j = int(np.nonzero(ctg["gap"] % 2 == 0)[0][0]); ok_elsewhere = ctg.copy(); ok_elsewhere["cov_sum"][j] += 1
    assert SC.contigs_equal(ok_elsewhere, seq, [expected[g] for g in sel], KK, sel)       # a gap outside the sample


def test_merge_round_check_catches_a_wrong_missing_or_extra_merged_contig():
    """The merged contigs a step appends (k = kv = 0, behind the assembly's own) against the oracle's merger on the gap's own contigs —
    and the pick over them: a gap its own contigs leave open is picked from the merged ones."""
    rng = np.random.RandomState(9)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.randint(0, 4, n)].tobytes().decode()
    g = rnd(1400)
    flanks = [(g[100:395], g[1005:1300]), (rnd(295), rnd(295))]
    own = [(0, g[60:520]), (0, g[450:900]), (0, g[830:1340]), (0, g[450:900]), (0, g[500:560]), (1, rnd(300)), (1, rnd(200))]
    exp0 = SC.expected_merged_contigs([s for gp, s in own if gp == 0])
    assert len(exp0) == 1 and exp0[0] in (g[60:1340], g[60:1340][::-1].translate(str.maketrans("ACGT", "TGCA")))   # duplicate + contained piece dropped, the three chained
    assert SC.expected_merged_contigs([s for gp, s in own if gp == 1]) == []
    recs = own + [(0, exp0[0])]
    n0 = len(own)
    ctg = np.zeros(len(recs), dtype=B.CONTIG)
    seq = bytearray()
    for i, (gp, s) in enumerate(recs):
        ctg[i] = (gp, 31 if i < n0 else 0, 29 if i < n0 else 0, max(1, len(s) - 28), len(s), 0, 0, len(seq))
        seq += s.encode()
    seq = bytes(seq)
    assert SC.merged_equal(ctg, seq, [0, 1], n0)
    wrong = bytearray(seq); at = int(ctg[n0]["seq_off"]) + 700; wrong[at] = ord("A") if wrong[at] != ord("A") else ord("C")
    assert not SC.merged_equal(ctg, bytes(wrong), [0], n0)                       # ONE wrong base in a merged contig
    assert not SC.merged_equal(ctg[:n0], seq, [0], n0)                           # the merged contig is missing
    extra = np.concatenate([ctg, ctg[n0:]]); extra["gap"][-1] = 1
    assert not SC.merged_equal(extra, seq, [1], n0)                              # a merged contig for a gap that has nothing to merge
    # the pick: no own contig carries both anchors, the merged one does
    kk = [(31, 29)]
    w = SC.expected_pick_word(ctg, seq, 0, flanks, kk)
    assert w >> 56 == 30 and 0x7FFFFFFF - ((w >> 1) & 0x7FFFFFFF) == n0 and ((w >> 32) & 0xFFFFFF) == 1005 - 395 + 1
    assert SC.expected_pick_word(ctg[:n0], seq, 0, flanks, kk) == 0
    best = np.array([w, 0], dtype=np.uint64)
    assert SC.picks_equal(ctg, seq, best, flanks, kk, [0, 1]) and not SC.picks_equal(ctg, seq, np.array([0, 0], dtype=np.uint64), flanks, kk, [0])
