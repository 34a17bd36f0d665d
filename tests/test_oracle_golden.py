"""The oracle (oracle/gp_oracle.py) against outputs of the REFERENCE ITSELF (tests/golden/*/expected.tar.gz)
and against known answers of the reference's own KmerUtils.cpp (tests/golden/kmerutils_kat.json)."""
import json
import os

import pytest

from golden_util import CASES, GOLDEN, Case
from oracle import gp_oracle as O


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return Case(request.param)


def _gaps(case):
    return O.gap_positions(case.fasta_records(), case.meta["min_gap"])


def test_gap_positions(case):
    got = ["%d %d %d %s" % g for g in _gaps(case)]
    assert got == case.exp_lines("gap_positions.txt")


def test_flank_regions(case):
    gaps = _gaps(case)
    ids = O.gap_ids(case.fai_names, gaps)
    seqs = dict(case.fasta_records())
    exp = case.exp_dir("flank_regions/")
    assert sorted(exp) == sorted(i + ".fa" for i in ids)
    for gid, (s, e, _, scf) in zip(ids, gaps):
        l, r = O.flank_seqs(seqs[scf], s, e, case.meta["flank"])
        assert exp[gid + ".fa"] == ">%s_left\n%s\n>%s_right\n%s\n" % (gid, l, gid, r)


def _collect(case, lib):
    return O.collect_library(lib["sam"].splitlines(), lib["fq1"], lib["fq2"], case.fai_names, _gaps(case),
                             lib["is"], lib["sd"], case.meta["clip_dist"], case.meta["anchor_mapq"])


def test_scaffold_lists_multiset(case):
    for lib in case.libs:
        got = _collect(case, lib)["lists"]
        exp = case.exp_dir(lib["folder"] + "/scaffold_reads_list_all/")
        assert sorted(exp) == sorted("%s_cluster_by_gap_reads_%s.list" % (s, side) for s in got for side in ("left", "right"))
        for scf in got:
            for side in ("left", "right"):
                e = exp["%s_cluster_by_gap_reads_%s.list" % (scf, side)].splitlines()
                assert sorted(got[scf][side]) == sorted(e), (scf, side)
                # the reference iterates SAM order; only the order of tags inside one record is dict-defined
                assert [l.split()[0] for l in got[scf][side]] == [l.split()[0] for l in e]


def test_discordant_positions(case):
    for lib in case.libs:
        got = ["%d %d %d %d" % r for r in _collect(case, lib)["rows"]]
        assert got == case.exp_lines(lib["folder"] + "/discordant_reads_pos.txt.sorted.txt")


def test_low_mapq_lists(case):
    for lib in case.libs:
        got = _collect(case, lib)["dlists"]
        exp = case.exp_dir(lib["folder"] + "/discordant_reads_list/")
        assert sorted(exp) == sorted("%s_cluster_by_discordant_reads_%s.list" % (s, side) for s in got for side in ("left", "right"))
        for scf in got:
            for side in ("left", "right"):
                e = exp["%s_cluster_by_discordant_reads_%s.list" % (scf, side)].splitlines()
                assert got[scf][side] == e, (scf, side)


def test_gap_fastq_byte_exact(case):
    per_lib = {"gap_reads": [], "gap_reads_high_quality": []}
    for lib in case.libs:
        got = _collect(case, lib)
        for kind in per_lib:
            exp = case.exp_dir("%s/%s/" % (lib["folder"], kind))
            assert sorted(exp) == sorted(k + ".fastq" for k in got[kind])
            for k, txt in got[kind].items():
                assert txt == exp[k + ".fastq"], (lib["folder"], kind, k)
            per_lib[kind].append(got[kind])
    ids = O.gap_ids(case.fai_names, _gaps(case))
    for kind in per_lib:
        merged = O.merge_libraries(per_lib[kind], ids)
        exp = case.exp_dir("merged/%s/" % kind)
        assert sorted(exp) == sorted(k + ".fastq" for k in merged)
        for k, txt in merged.items():
            assert txt == exp[k + ".fastq"]


def test_kmerutils_known_answers():
    kat = json.load(open(os.path.join(GOLDEN, "kmerutils_kat.json")))
    for c in kat["pack"]:
        assert ["%016x" % v for v in O.all_kmers64(c["seq"], c["k"])] == c["kmers_hex"], c["seq"]
        assert O.pack_kmer64(c["seq"], 0, c["k"]) == int(c["kmers_hex"][0], 16)
    for c in kat["tostr"]:
        assert O.kmer_to_string(int(c["kmer_hex"], 16), c["k"]) == c["str"]
    for c in kat["predicate"]:
        src = O.all_kmers64(c["src"], c["k"])
        assert int(O.read_contains_freq_kmers(src, c["read"], c["k"], c["thr"])) == c["result"], c


def test_wide_kmer_layout_extends_kmerutils():
    s = "GATTACAGATTACAGATTACAGATTACAGATTACAGATTACA"
    for k in (1, 5, 31, 32):
        assert O.pack_kmer128(s, 3, k) >> 64 == O.pack_kmer64(s, 3, k)
    f = O.pack_kmer128(s, 0, 41)
    assert O.kmer128_to_string(O.revcomp_kmer128(f, 41), 41) == O._rc(s[:41])


def test_quick_check_equals_the_reference_prefilter():
    """oracle or_quick_check vs the feasible pairs the reference's own QuickCheckerContigsMatch printed (oracle/_ref/quickcheck_kat,
    ContigsCompactor.cpp:1982-2095): every committed contig set, k = 8, 10, 12."""
    import gzip
    from oracle import c_oracle as CO
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "quickcheck_kat.json.gz"), "rt").read())
    assert len(cases) >= 9
    total = 0
    for c in cases:
        got = CO.quick_check(c["contigs"], c["k"])
        assert got == [tuple(p) for p in c["pairs"]], (c["k"], len(c["contigs"]))
        total += len(got)
    assert total > 300


def test_overlap_evaluation_equals_the_reference():
    """oracle or_overlap_evaluate vs the reference's own ContigsCompactor::Evaluate (oracle/_ref/evaluate_kat, ContigsCompactor.cpp:
    1572-1976) on every ordered node pair of every committed contig set: class, and for classes 1/2 the end row, the clip, the
    overlap size, the merged length and the containment flag."""
    import gzip
    from oracle import c_oracle as CO
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "evaluate_kat.json.gz"), "rt").read())
    assert len(cases) >= 8
    n_pairs = n_ovl = 0
    for c in cases:
        nodes = CO.merger_nodes(c["contigs"])
        for r in c["results"]:
            got = CO.overlap_evaluate(nodes[r[0]], nodes[r[1]], c["params"])
            assert got["res"] == r[2], (r, got)
            if r[2]:
                assert [got["row_end"], got["nclip"], got["overlap"], got["merged_len"], got["containment"]] == r[3:], (r, got)
                n_ovl += 1
            n_pairs += 1
    assert n_pairs >= 900 and n_ovl >= 100


# ---- f-1: the contig picker (pick_contigs.py:64-358, 361-539), from flanks.sam on ----

def test_picker_restatement_equals_the_reference_on_prepared_sam_files():
    """tests/golden/pick_kat.json.gz: 504 prepared flanks.sam texts (hand-built: every clip-type pair, both strands, several hits of
    a type, equal spans, span 0 and -1, secondary lines, indel CIGARs, lower case / IUPAC; random hit sets and random pairs) and the
    picked_seqs.fa / picked_contigs.fa the reference's own pick_contigs.py wrote for each, full pick and extended pick."""
    import gzip
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "pick_kat.json.gz")).read())
    n_full = n_ext = 0
    for c in cases:
        contigs = [tuple(x) for x in c["contigs"]]
        assert list(O.pick_full_from_sam(c["id"], c["sam"], contigs)) == c["full"], c["id"]
        n_full += c["full"][0] is not None
        if c["ext"] != "TIE":                              # (the reference's int-vs-str tie test raises under Python 3)
            assert list(O.pick_extended_from_sam(c["id"], c["sam"], contigs)) == c["ext"], c["id"]
            n_ext += c["ext"][0] is not None
    assert len(cases) >= 500 and n_full > 150 and n_ext > 300


def test_picker_ledger_is_the_concatenation_of_the_per_gap_files():
    """pick_full_constructed_contigs appends every gap's picked_seqs.fa / picked_contigs.fa to the ledger in list order (:564-572)."""
    import gzip
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "pick_kat.json.gz")).read())
    for score in (30, 15):
        mine = [c for c in cases if c["score"] == score]
        led = [c["ledgers_of_score"] for c in mine if c["ledgers_of_score"]][0]
        assert led[0] == "".join(c["full"][0] or "" for c in mine) and led[1] == "".join(c["full"][1] or "" for c in mine)


# ---- f-3: ContigsMerger from the edges to the merged contigs (ContigsCompactor.cpp:773-983, GraphUtils.cpp:625-859) ----

def test_merger_restatement_equals_the_reference_binary():
    """tests/golden/merger_kat.json.gz: 41 contig sets (chains, reverse-complemented links, forks, alternative routes, containment,
    dirty ends, a 2-cycle, random tilings with errors) and the NEW_CONTIG_MERGE records + paths the reference's own ContigsMerger
    (built from its sources, -t 1) printed: same merged sequences, same paths, same order."""
    import gzip
    from oracle import c_oracle as CO
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "merger_kat.json.gz")).read())
    n_new = 0
    for c in cases:
        got = O.merger_new_contigs(c["contigs"], CO.GAPPADDER_OVL)
        mine = [{"seq": s, "path": ["c%d%s" % (v >> 1, "_R" if v & 1 else "") for v in p]} for p, s in got]
        assert mine == c["new"]
        n_new += len(mine)
    assert len(cases) >= 40 and n_new >= 45
