"""GPU parity of the contig-merge prefilter (SURVEY.md §8f-3 first piece: QuickCheckerContigsMatch, ContigsCompactor.cpp:1982-2095)
and of the pairwise overlap evaluation (second piece: ContigsCompactor::Evaluate, :1572-1976) through the C ABI: the reference's own answers (tests/golden/quickcheck_kat.json.gz, printed by oracle/_ref/quickcheck_kat) and the
oracle on contig sets as the assembly produces them."""
import gzip
import json
import os

import numpy as np
import pytest

from golden_util import GOLDEN
from oracle import c_oracle as CO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gf():
    from gappadder_amd.hip_api import GapFill
    g = GapFill(0)
    yield g
    g.close()


def _triples(out):
    return [(int(x["set"]), int(x["i"]), int(x["j"])) for x in out]


def test_quick_check_equals_the_reference_vectors(gf):
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "quickcheck_kat.json.gz"), "rt").read())
    for k in sorted({c["k"] for c in cases}):
        batch = [c for c in cases if c["k"] == k]
        got = _triples(gf.quick_check([c["contigs"] for c in batch], k))
        want = [(s, i, j) for s, c in enumerate(batch) for i, j in c["pairs"]]
        assert got == want, k
    assert sum(len(c["pairs"]) for c in cases) > 300


def test_quick_check_on_many_sets_matches_oracle(gf):
    """Hundreds of sets in one call (one workgroup per set, dynamic hand-out), sets larger than one table chunk (> 195 nodes),
    an empty set, lower case and N."""
    rng = np.random.default_rng(7)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    sets = []
    for s in range(150):
        g = rnd(int(rng.integers(400, 3000)))
        n = int(rng.integers(1, 40))
        cs = []
        for _ in range(n):
            a = int(rng.integers(0, len(g) - 60))
            c = g[a:a + int(rng.integers(30, 400))]
            if rng.integers(0, 2):
                c = c[::-1].translate(str.maketrans("ACGT", "TGCA"))
            if rng.integers(0, 10) == 0:
                c = c[:10] + "N" + c[11:]
            if rng.integers(0, 10) == 0:
                c = c.lower()
            cs.append(c)
        sets.append(cs)
    sets.append([])
    g = rnd(20000)
    sets.append([g[a:a + 150] for a in range(0, 19000, 60)])          # 317 contigs = 634 nodes: four table chunks
    got = _triples(gf.quick_check(sets, 10))
    want = [(s, i, j) for s, cs in enumerate(sets) for i, j in (CO.quick_check(cs, 10) if cs else [])]
    assert got == want and len(want) > 5000


def test_quick_check_rejects_short_contigs(gf):
    from gappadder_amd import _lib as B
    with pytest.raises(B.GapFillError) as e:
        gf.quick_check([["ACGT" * 10, "ACGTACGT"]], 10)
    assert e.value.code == B.GF_E_INVAL


def test_quick_check_on_assembled_contigs(gf):
    """The use the reference makes of it: the contigs.fa of a gap (all (k, kv) pairs) -> candidate pairs for the overlap merge."""
    import synth_small as S
    from gappadder_amd.hip_api import GapFill
    c = S.small_case(seed=21, n_pairs=12000)
    L = c["L"]
    hits = CO.screen_reads(c["reads_blob"], L, c["flanks"], 31)
    sets = []
    for g in range(len(c["gaps"])):
        ids = sorted(set(int(h["read"]) for h in hits if h["gap"] == g) | set(int(h["read"]) ^ 1 for h in hits if h["gap"] == g))
        pool = b"".join(c["reads_blob"][i * L:(i + 1) * L] for i in ids)
        cs = [s for k, kv in ((31, 29), (41, 39)) for s, _, _ in CO.assemble_pool(pool, L, k, kv)]
        sets.append(cs)
    got = _triples(gf.quick_check(sets, 10))
    want = [(s, i, j) for s, cs in enumerate(sets) for i, j in CO.quick_check(cs, 10)]
    assert got == want and len(want) > len(sets)
    # the same genome assembled at two k's overlaps itself: more than the diagonal is feasible
    assert any(i // 2 != j // 2 for _, i, j in want)


OVL_FIELDS = ("res", "row_end", "col_end", "nclip", "score", "contained", "merged_len", "overlap", "containment", "first_goes_first")


def _all_ordered_pairs(sets):
    from gappadder_amd import _lib as B
    tr = [(s, i, j) for s, cs in enumerate(sets) for i in range(2 * len(cs)) for j in range(2 * len(cs)) if i != j]
    out = np.zeros(len(tr), dtype=B.QCPAIR)
    out["set"], out["i"], out["j"] = [t[0] for t in tr], [t[1] for t in tr], [t[2] for t in tr]
    return out


def test_overlap_evaluate_equals_the_reference_vectors(gf):
    """Every ordered node pair of every committed contig set against what the reference's own Evaluate printed
    (tests/golden/evaluate_kat.json.gz), one GPU call per parameter set."""
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "evaluate_kat.json.gz"), "rt").read())
    n_ovl = 0
    for c in cases:
        pairs = _all_ordered_pairs([c["contigs"]])
        got = gf.overlap_evaluate([c["contigs"]], pairs, c["params"])
        assert [(int(p["i"]), int(p["j"])) for p in pairs] == [(r[0], r[1]) for r in c["results"]]
        for g, r in zip(got, c["results"]):
            assert int(g["res"]) == r[2], (r, g)
            if r[2]:
                assert [int(g[f]) for f in ("row_end", "nclip", "overlap", "merged_len", "containment")] == r[3:], (r, g)
                n_ovl += 1
    assert n_ovl >= 100


def test_overlap_evaluate_matches_oracle_on_assembly_like_sets(gf):
    """Tilings of random genomes with substitutions and indels, both strands, lower case and N, contigs up to 3 kb, several sets in
    one call; every field of every ordered pair against the oracle; and the edges merge_edges would write."""
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    sets = []
    for s in range(6):
        g = rnd(int(rng.integers(1500, 6000)))
        cs = []
        for _ in range(int(rng.integers(2, 7))):
            a = int(rng.integers(0, len(g) - 200))
            c = list(g[a:a + int(rng.integers(60, 3000 if s == 0 else 900))])
            for _ in range(int(rng.integers(0, 6))):
                i = int(rng.integers(1, len(c) - 1))
                op = int(rng.integers(0, 3))
                if op == 0: c[i] = "ACGT"[int(rng.integers(0, 4))]
                elif op == 1: del c[i]
                else: c.insert(i, "ACGT"[int(rng.integers(0, 4))])
            c = "".join(c)
            if rng.integers(0, 2):
                c = c[::-1].translate(str.maketrans("ACGT", "TGCA"))
            if rng.integers(0, 8) == 0:
                c = c[:20] + "N" + c[21:]
            if rng.integers(0, 8) == 0:
                c = c.lower()
            cs.append(c)
        sets.append(cs)
    pairs = _all_ordered_pairs(sets)
    got = gf.overlap_evaluate(sets, pairs)
    n_edge = 0
    for p, g in zip(pairs, got):
        nodes = CO.merger_nodes(sets[int(p["set"])])
        exp = CO.overlap_evaluate(nodes[int(p["i"])], nodes[int(p["j"])])
        assert {f: int(g[f]) for f in OVL_FIELDS} == exp, (int(p["set"]), int(p["i"]), int(p["j"]))
        n_edge += exp["res"] == 2 and not exp["containment"]
    assert n_edge > 10
    with pytest.raises(Exception):
        gf.overlap_evaluate(sets, pairs[:4], (-2.0, -1.5, 50.0, 0.005, 0.4, 12.0, 6.0))   # a fractional indel score is refused


def test_merge_edges_file_lists_the_graph_edges(gf, tmp_path):
    """MergeContigs.merge_edges on a working folder with two gaps: merge_edges.txt holds, per gap, exactly the edges the reference's
    threadMergeContigV2 would form from the feasible pairs (class 2, no containment; mode 12 / 21; overlap size) — here derived
    with the oracle from the oracle's own prefilter."""
    from gappadder_amd.MergeContigs import merge_edges
    rng = np.random.default_rng(5)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    rc = lambda x: x[::-1].translate(str.maketrans("ACGT", "TGCA"))
    g = rnd(1200)
    sets = {"7_1": [g[0:400], g[330:800], rc(g[720:1200]), rnd(150)], "7_2": [g[100:500], g[100:300]], "9_1": [rnd(20)]}
    wf = str(tmp_path) + "/"
    for gid, cs in sets.items():
        os.makedirs(wf + "velvet_temp/" + gid)
        with open(wf + "velvet_temp/%s/contigs.fa" % gid, "w") as f:
            f.write("".join(">NODE_%d_length_%d_cov_9.0\n%s\n" % (i + 1, len(c), c) for i, c in enumerate(cs)))
    got = merge_edges(gf, wf, ["7_1", "7_2", "9_1", "missing"])
    assert set(got) == {"7_1", "7_2"}                       # 9_1 has no contig of 30 bases, `missing` no file
    n_edges = 0
    for gid in ("7_1", "7_2"):
        nodes = CO.merger_nodes(sets[gid])
        exp = []
        for (i, j) in CO.quick_check(sets[gid], 10):
            r = CO.overlap_evaluate(nodes[i], nodes[j])
            if r["res"] == 2 and not r["containment"]:
                exp.append((i, j, "12" if r["first_goes_first"] else "21", r["overlap"]))
        assert got[gid] == exp, gid
        lines = open(wf + "velvet_temp/%s/merge_edges.txt" % gid).read().split("\n")[:-1]
        assert len(lines) == len(exp)
        for line, (i, j, mode, ov) in zip(lines, exp):
            a, sa, b, sb, m, o = line.split()
            assert (a, sa, b, sb, m, int(o)) == ("NODE_%d_length_%d_cov_9.0" % (i // 2 + 1, len(sets[gid][i // 2])), "-" if i & 1 else "+",
                                                 "NODE_%d_length_%d_cov_9.0" % (j // 2 + 1, len(sets[gid][j // 2])), "-" if j & 1 else "+", mode, ov)
        n_edges += len(exp)
    assert n_edges >= 4                                     # the tiling's overlaps in both orientations


def test_merge_contigs_on_the_device_equals_the_reference_binary(gf, tmp_path):
    """MergeContigs.merge_contigs with the GPU prefilter + overlap evaluation (strict and relaxed mode) on the contig sets the
    reference's own ContigsMerger answered (tests/golden/merger_kat.json.gz): the NEW_CONTIG_MERGE sequences and merge.info paths of
    every set without contained contigs equal the reference's; all sets equal the oracle on the de-duplicated set."""
    import gzip
    import json
    from golden_util import GOLDEN
    from gappadder_amd import MergeContigs as MC
    from oracle import gp_oracle as O
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "merger_kat.json.gz")).read())
    wf = str(tmp_path) + "/"
    ids = []
    for ci, c in enumerate(cases):
        gid = "3_%d" % (ci + 1)
        ids.append(gid)
        os.makedirs(wf + "velvet_temp/" + gid)
        with open(wf + "velvet_temp/%s/contigs.fa" % gid, "w") as f:
            f.write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(c["contigs"])))
    done = MC.merge_contigs(gf, wf, ids)
    n_direct = n_new = 0
    for gid, c in zip(ids, cases):
        d = wf + "velvet_temp/%s/" % gid
        nodup = MC.drop_contained([("c%d" % i, s) for i, s in enumerate(c["contigs"])])
        new = [s for n, s in MC.read_fasta(d + "contigs.fa_no_dup.fa.merged.fa") if n.startswith("NEW_CONTIG_MERGE_")]
        assert new == [s for _, s in O.merger_new_contigs([s for _, s in nodup], CO.GAPPADDER_OVL)] and done[gid] == len(new), gid
        if len(nodup) == len(c["contigs"]):
            assert new == [x["seq"] for x in c["new"]], gid
            assert [l.split()[1:] for l in open(d + "contigs.fa_no_dup.fa.merge.info").read().splitlines()] == [x["path"] for x in c["new"]], gid
            n_direct += 1
        n_new += len(new)
    assert n_direct >= 30 and n_new >= 40


def _one_gap_folder(root, left, right, reads, hq_reads=None):
    """A working folder with one gap '0_1': flank FASTA, read pool (and high-quality pool), .fai + gap_positions for prepare_list."""
    wf = root + "/wf/"
    for sub in ("merged/gap_reads", "merged/gap_reads_high_quality", "merged/velvet_temp", "flank_regions"):
        os.makedirs(wf + sub)
    open(wf + "flank_regions/0_1.fa", "w").write(">0_1_left\n%s\n>0_1_right\n%s\n" % (left, right))
    open(wf + "merged/gap_reads/0_1.fastq", "w").write("".join("@r%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(reads)))
    if hq_reads is not None:
        open(wf + "merged/gap_reads_high_quality/0_1.fastq", "w").write("".join("@h%d\n%s\n+\n%s\n" % (i, s, "I" * len(s)) for i, s in enumerate(hq_reads)))
    open(root + "/d.fai", "w").write("scf0\t100000\t6\t60\t61\n")
    open(wf + "gap_positions.txt", "w").write("5000 5600 600 scf0\n")
    return wf


def _tile(g, L=100, step=5):
    rc = lambda x: x[::-1].translate(str.maketrans("ACGT", "TGCA"))
    out = []
    for s in list(range(0, len(g) - L + 1, step)) + [len(g) - L]:
        out += [g[s:s + L], rc(g[s:s + L])]
    return out


def test_a_gap_that_only_the_contig_merging_closes(gf, tmp_path):
    """assemble_pipeline (assemble_gaps.py:328-368): the reads hold a long dead-end branch inside the gap (80 bases seen by several
    reads: not a tip), so the assembly stops at the junction — no contig carries both anchors; ContigsMerger's path search chains
    the unitigs across it (they overlap by kv - 1 bases) and the merged contig closes the gap with the true sequence."""
    from gappadder_amd import assemble_gaps as AG
    rng = np.random.default_rng(77)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    g = rnd(1400)                                          # left flank 100..395, gap 400..1000, right flank 1005..1300
    left, right = g[100:395], g[1005:1300]
    branch = g[560:680] + rnd(80)                          # follows the genome to 680, then leaves it for 80 bases
    reads = _tile(g[60:1340]) + _tile(branch, step=4)
    wf = _one_gap_folder(str(tmp_path), left, right, reads)
    ga = AG.GapAssembler(str(tmp_path) + "/d.fai", wf + "gap_positions.txt", 1, wf + "merged/", kmer_list=[(31, 29)], gf=gf)
    # without merging: open
    ga.assembly(["0_1"])
    from gappadder_amd.pick_contigs import ContigsSelection, read_fasta
    assert ContigsSelection(wf + "merged/").pick_full_constructed_contigs(30, ["0_1"], wf + "tmp_picked.fa") == 0
    n_before = len(read_fasta(wf + "merged/velvet_temp/0_1/contigs.fa"))
    res = ga.assemble_pipeline()
    assert res["closed"] == 1 and res["gaps_with_merged_contigs"] >= 1 and n_before >= 3
    picked = open(wf + "picked_seqs.fa").read().split("\n")
    assert picked[0].startswith(">0_1_NEW_CONTIG_MERGE_") and picked[1] in (g[395:1006], g[394:1005])
    assert os.path.exists(wf + "merged/velvet_temp/0_1/original_contigs_before_merging.fa")


def test_a_gap_that_only_the_bridging_high_quality_reads_close(gf, tmp_path):
    """The rescue round (assemble_gaps.py:357-361): the pool misses a stretch inside the gap, so two contigs end 40 bases apart with
    nothing to overlap; a high-quality read that spans the hole aligns clipped to both, is appended to the contigs, and the next
    merge chains left contig -> read -> right contig: the gap is closed at the last pick."""
    from gappadder_amd import assemble_gaps as AG
    rng = np.random.default_rng(78)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    g = rnd(1400)
    left, right = g[100:395], g[1005:1300]
    reads = _tile(g[60:700]) + _tile(g[740:1340])          # nothing covers 700..740 with a whole k-mer
    hq = [g[640:790], g[200:350], rnd(150)]                # one read across the hole, one inside a contig, one stranger
    wf = _one_gap_folder(str(tmp_path), left, right, reads, hq)
    ga = AG.GapAssembler(str(tmp_path) + "/d.fai", wf + "gap_positions.txt", 1, wf + "merged/", kmer_list=[(31, 29)], gf=gf)
    res = ga.assemble_pipeline()
    assert res["bridging_reads"] == 1 and res["closed"] == 1, res
    picked = open(wf + "picked_seqs.fa").read().split("\n")
    assert picked[1] in (g[395:1006], g[394:1005])


def _oracle_round(contig_sets, max_set=128):
    """What the merge round must append per gap: the merger's NEW_CONTIG_MERGE sequences (oracle/gp_oracle.py::merger_new_contigs, pinned on
    the reference binary's answers) of the exact-containment de-duplicated set, for sets of 2 .. max_set contigs after the dedup; contigs
    outside 30 .. 8190 bases take no part (MergeContigs.merge_sets)."""
    from gappadder_amd import MergeContigs as MC
    from oracle import gp_oracle as O
    out = []
    for cs in contig_sets:
        nodup = MC.drop_contained([("c%d" % i, s) for i, s in enumerate(cs)]) if 2 <= len(cs) <= 1024 else []
        if not 2 <= len(nodup) <= max_set:
            out.append([])
            continue
        nodes = [s.upper() for _, s in nodup if MC.MIN_NODE <= len(s) <= MC.MAX_NODE]
        out.append([s for _, s in O.merger_new_contigs(nodes, CO.GAPPADDER_OVL)] if len(nodes) >= 2 else [])
    return out


def test_device_merge_round_reproduces_the_reference_binarys_merged_contigs(gf):
    """gf_merge_open_gaps_dev — dedup, prefilter, overlap evaluation, graph, strongly connected components, roots / ends, shortest paths,
    twin removal and the merged strings, all on the device without a host synchronisation — on the contig sets the reference's own
    ContigsMerger answered (tests/golden/merger_kat.json.gz): every set's NEW_CONTIG_MERGE sequences in the reference's order
    (GraphUtils.cpp:625-859, 1028-1178, 1258-1344, 1422-1454; ContigsCompactor.cpp:773-983, 1456-1520)."""
    from gappadder_amd import MergeContigs as MC
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "merger_kat.json.gz")).read())
    sets = [c["contigs"] for c in cases]
    got, stats = gf.merge_round(sets)
    want = _oracle_round(sets)
    n_direct = 0
    for gi, c in enumerate(cases):
        assert got[gi] == want[gi], gi
        if len(MC.drop_contained([("c%d" % i, s) for i, s in enumerate(c["contigs"])])) == len(c["contigs"]):
            assert got[gi] == [x["seq"] for x in c["new"]], gi          # the reference binary's own answer
            n_direct += 1
    assert n_direct >= 30 and sum(len(g) for g in got) >= 40 and stats["error_bits"] == 0
    assert stats["new_contigs"] == sum(len(g) for g in got) and stats["gaps_with_new_contigs"] == sum(1 for g in got if g)


def test_device_merge_round_on_random_contig_sets(gf):
    """Contig sets as a fragmented assembly leaves them — overlapping pieces of a genome on either strand, duplicates, contained pieces,
    strangers, branches that make the graph fork and cycle (a repeat) — 120 gaps in one call, some of them closed (left alone):
    device round == oracle, and == the host twin MergeContigs.merge_sets (two batched GPU calls + host path search)."""
    from gappadder_amd import MergeContigs as MC
    rng = np.random.default_rng(11)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    rc = lambda x: x[::-1].translate(str.maketrans("ACGT", "TGCA"))
    sets, open_gaps = [], []
    for s in range(120):
        g = rnd(int(rng.integers(600, 4000)))
        if s % 7 == 0:                                   # a two-copy repeat: the graph forks and joins
            rep = rnd(120)
            g = g[:200] + rep + g[200:len(g) // 2] + rep + g[len(g) // 2:]
        cs, at = [], 0
        while at < len(g) - 80:
            ln = int(rng.integers(80, 700))
            c = g[at:at + ln]
            if rng.integers(0, 50) == 0:                 # a substitution inside the overlap
                q = int(rng.integers(0, len(c)))
                c = c[:q] + "ACGT"[("ACGT".index(c[q]) + 1) % 4] + c[q + 1:]
            cs.append(rc(c) if rng.integers(0, 2) else c)
            at += max(20, ln - int(rng.integers(15, 120)))
        if rng.integers(0, 3) == 0:
            cs.append(cs[0])                             # a duplicate
        if rng.integers(0, 3) == 0:
            cs.append(g[100:160])                        # contained in a longer contig
        if rng.integers(0, 4) == 0:
            cs.append(rnd(200))                          # a stranger
        order = rng.permutation(len(cs))
        sets.append([cs[i] for i in order])
        open_gaps.append(s % 10 != 3)
    sets.append([rnd(100)])                              # one contig: nothing to merge
    open_gaps.append(True)
    sets.append([])
    open_gaps.append(True)
    got, stats = gf.merge_round(sets, open_gaps=open_gaps)
    want = _oracle_round(sets)
    host = MC.merge_sets(gf, [MC.drop_contained([("c%d" % i, s) for i, s in enumerate(cs)]) for cs in sets])
    n_new = 0
    for gi in range(len(sets)):
        if not open_gaps[gi]:
            assert got[gi] == [], gi                     # a closed gap is left alone
            continue
        assert got[gi] == want[gi], gi
        if 2 <= len(MC.drop_contained([("c%d" % i, s) for i, s in enumerate(sets[gi])])) <= 128:
            assert got[gi] == [s for _, s, _ in host[gi]["new"]], gi
        n_new += len(got[gi])
    assert n_new > 100 and stats["gaps_tried"] >= 100 and stats["error_bits"] == 0 and stats["edges"] > 300


def test_device_merge_round_leaves_oversized_sets_alone(gf):
    """A gap with more than max_set contigs after the dedup is skipped and counted (the host round does the same: the contig graph of a
    repeat-bearing gap has thousands of paths); the gap beside it is merged."""
    rng = np.random.default_rng(12)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    g = rnd(900)
    sets = [[rnd(90) for _ in range(140)], [g[0:400], g[330:800], g[700:900]]]
    got, stats = gf.merge_round(sets)
    assert got[0] == [] and stats["gaps_skipped_large"] == 1 and stats["gaps_tried"] == 1
    assert got[1] == _oracle_round(sets)[1] and len(got[1]) >= 1 and g in (got[1][0], got[1][0][::-1].translate(str.maketrans("ACGT", "TGCA")))
    got8, stats8 = gf.merge_round(sets, max_set=2)
    assert got8 == [[], []] and stats8["gaps_skipped_large"] == 2


def test_device_merge_round_takes_a_gaps_contigs_in_contigs_fa_order(gf):
    """With the (k, kv) list the round orders a gap's contigs like its contigs.fa (assemble_gaps.py:124-135: pairs in list order, inside a
    pair by length descending, then sequence) whatever order the records stand in — the assembly appends them in an unspecified order —,
    so the merged contigs do not depend on it: every shuffle of the records gives the oracle's answer for the contigs.fa order."""
    rng = np.random.default_rng(21)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    kk = [(41, 39), (31, 29), (51, 49)]                      # list order, not sorted
    sets = []
    for s in range(40):
        g = rnd(int(rng.integers(800, 2500)))
        recs, at = [], 0
        while at < len(g) - 100:
            ln = int(rng.integers(100, 600))
            k, kv = kk[int(rng.integers(0, 3))]
            recs.append((k, kv, g[at:at + ln]))
            at += max(30, ln - int(rng.integers(20, 90)))
        recs.append((31, 29, recs[0][2]))                    # the same sequence under another pair: of identical contigs the FIRST in order stays
        sets.append(recs)
    order_of = {p: q for q, p in enumerate(kk)}
    canon = [[c for (_, _, c) in sorted(recs, key=lambda r: (order_of[(r[0], r[1])], -len(r[2]), r[2]))] for recs in sets]
    want = _oracle_round(canon)
    n_new = 0
    for trial in range(3):
        shuffled = [[recs[i] for i in rng.permutation(len(recs))] for recs in sets]
        got, stats = gf.merge_round(shuffled, k_pairs=kk)
        assert got == want, trial
        n_new += sum(len(x) for x in got)
    assert n_new > 60 and stats["error_bits"] == 0


def test_device_merge_round_skips_a_graph_beyond_its_limits_and_is_repeatable(gf):
    """A set whose overlap graph outgrows the round's own limits (more than 4 096 edges: a hundred windows of one short sequence, every
    pair overlapping) is left alone and counted — no error, the gap beside it is merged; and the round gives the same answer when it
    runs again on the same records (every order-dependent step follows the contigs' order, not the launch's)."""
    rng = np.random.default_rng(31)
    lut = np.frombuffer(b"ACGT", np.uint8)
    rnd = lambda n: lut[rng.integers(0, 4, n)].tobytes().decode()
    g0, g1 = rnd(700), rnd(1500)
    dense = [g0[a:a + 400] for a in range(0, 300, 3)]                  # 100 contigs, every pair overlaps by 100+ bases
    sparse = [g1[0:500], g1[430:1000], g1[930:1500]]
    got, stats = gf.merge_round([dense, sparse])
    assert got[0] == [] and stats["gaps_skipped_graph"] == 1 and stats["error_bits"] == 0 and stats["gaps_tried"] == 2
    assert got[1] == _oracle_round([dense[:2], sparse])[1] and len(got[1]) == 1
    again, stats2 = gf.merge_round([dense, sparse])
    assert again == got and {k: v for k, v in stats2.items()} == {k: v for k, v in stats.items()}
