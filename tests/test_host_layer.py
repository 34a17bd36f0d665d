"""Host-side mirror of the reference's stage interfaces (gappadder_amd/*.py) on CPU: Preprocess end-to-end, and the
FASTQ join / library merge fed with the REFERENCE's own list files — all against the reference-generated golden trees."""
import os

import pytest

from golden_util import CASES, Case
import pipeline_util as PU


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return Case(request.param)


def test_preprocess_cli_matches_reference(case, tmp_path):
    from gappadder_amd import main as M
    cfgp, wf, _ = PU.materialise(case, str(tmp_path))
    M.main(["-c", "Preprocess", "-g", cfgp])
    got = PU.tree(wf)
    assert got["gap_positions.txt"] == case.expected["gap_positions.txt"]
    exp = case.exp_dir("flank_regions/")
    assert {k: v for k, v in got.items() if k.startswith("flank_regions/")} == {"flank_regions/" + k: v for k, v in exp.items()}


def test_join_and_merge_from_reference_lists(case, tmp_path):
    """collect_discordant_regions_v2, merge_dispatch_reads_for_gaps_v2, dispatch_high_quality_reads_for_gaps and
    merge_reads_v2 given the reference's scaffold/discordant lists reproduce its position file and per-gap FASTQ files."""
    from gappadder_amd import main as M
    from gappadder_amd.merge_reads import ReadsMerger
    from gappadder_amd.run_multi_threads_discordant import DiscordantReadsCollector
    cfgp, wf, st = PU.materialise(case, str(tmp_path))
    M.main(["-c", "Preprocess", "-g", cfgp])
    cfg = M.parse_configuration(cfgp)
    folders = M.prepare_folders(cfg["alignments"], wf)
    sf_fai = cfg["draft"] + ".fai"
    for lib, folder, (bam, _, _), (l, r) in zip(case.libs, folders, cfg["alignments"], cfg["raw_reads"]):
        for sub in ("scaffold_reads_list_all/", "discordant_reads_list/"):
            for name, txt in case.exp_dir(lib["folder"] + "/" + sub).items():
                open(folder + sub + name, "w").write(txt)
        drc = DiscordantReadsCollector(sf_fai, bam, folder, 2, gf=object(), samtools_path=st)
        drc.collect_discordant_regions_v2(folder + "discordant_reads_pos.txt")
        drc.merge_dispatch_reads_for_gaps_v2(l, r)
        drc.dispatch_high_quality_reads_for_gaps(l, r)
    for name in ("gap_reads", "gap_reads_alignment", "gap_reads_high_quality"):
        ReadsMerger().merge_reads_v2(sf_fai, wf + "gap_positions.txt", folders, name, wf + "merged/", 2)
    got = PU.tree(wf)
    for rel, txt in case.expected.items():
        top = rel.split("/")[0]
        if rel.endswith(".fastq") or rel.endswith("discordant_reads_pos.txt.sorted.txt") or "discordant_temp/" in rel:
            assert got.get(rel) == txt, rel
        elif rel.endswith("_reads.list") or rel.endswith("discordant_reads_pos.txt"):
            assert sorted(got[rel].splitlines()) == sorted(txt.splitlines()), rel   # dict-ordered in the reference
    assert not [k for k in got if k.endswith(".fastq") and k not in case.expected]


def test_velvet_style_fasta_format():
    from gappadder_amd.assemble_gaps import format_contigs, velvet_kv
    txt = format_contigs([("ACGT" * 20, 52, 150), ("TTTT", 1, 3)])
    lines = txt.splitlines()
    assert lines[0] == ">NODE_1_length_52_cov_2.884615" and len(lines[1]) == 60 and lines[2] == "ACGT" * 5
    assert lines[3] == ">NODE_2_length_1_cov_3.000000"
    assert "-" not in lines[0] and " " not in lines[0]      # names flow into '-'/'_' split protocols downstream
    assert velvet_kv(29) == 29 and velvet_kv(28) == 27


def test_gap_scan_quirks():
    from gappadder_amd.gnrt_pos_true_seqs import flanks, scan_gaps
    from oracle import gp_oracle as O
    for seq in ("ACGT" + "N" * 120 + "acgtACGT" + "N" * 99 + "A" + "NNNN", "N" * 200, "ACGTNNNNnnnnNNNNA" * 10, "NACGT" + "N" * 100 + "C"):
        assert scan_gaps(seq, 100) == O.scan_gaps(seq, 100)
        assert scan_gaps(seq, 3) == O.scan_gaps(seq, 3)
    s = "ACGTACGTAC" * 100
    for start in (3, 5, 299, 300, 301):
        assert flanks(s, start, start + 50, 300) == O.flank_seqs(s, start, start + 50, 300)


def test_both_unmapped_round_files_match_reference(tmp_path):
    """collect_both_unmapped_reads.py, file side: the per-BAM / merged / split both-unmapped FASTQ files and
    gap_contigs_all.fa equal what the reference wrote for the same SAM text and the same first-round contigs
    (tests/golden/twolib/round2.json.gz; record order compared as written: py3 dicts keep insertion order on both sides)."""
    import gzip
    import json
    from gappadder_amd.collect_both_unmapped_reads import BothUnmappedReadsCollector
    case = Case("twolib")
    r2 = json.loads(gzip.open(os.path.join(case.dir, "round2.json.gz")).read())
    cfgp, wf, st = PU.materialise(case, str(tmp_path))
    data = os.path.join(str(tmp_path), "data")
    mwf = wf + "merged/"
    for key, fa in r2["contigs"].items():
        os.makedirs(mwf + "velvet_temp/" + key)
        open(mwf + "velvet_temp/%s/contigs.fa" % key, "w").write(fa)
    os.makedirs(mwf + "gap_reads")

    class NoGpu:     # the SAM text of the fixture carries '*' sequences: nothing to screen, the GPU must not be needed
        def __getattr__(self, name):
            raise AssertionError("GPU touched")
    bams = [os.path.join(data, "lib%d.bam" % i) for i in range(len(case.libs))]
    b = BothUnmappedReadsCollector(mwf, samtools_path=st, gf=NoGpu(), k=31)
    b.collect_both_unmapped_reads(bams, r2["ids"])
    for i in range(len(case.libs)):
        assert open(bams[i] + ".both_unmapped.fq").read() == r2["files"]["lib%d.both_unmapped.fq" % i]
    for fn in ("both_unmapped.fq", "both_unmapped_1.fq", "both_unmapped_2.fq", "gap_contigs_all.fa"):
        assert open(mwf + fn).read() == r2["files"][fn], fn
    assert len(r2["files"]["both_unmapped_1.fq"]) > 500 and r2["files"]["gap_contigs_all.fa"].count(">") == 4


def test_write_back_matches_reference(tmp_path):
    """put_gap_seq_back_to_scaffold.py: same new scaffold FASTA as the reference for the same draft, gap table and picked
    sequences (two gaps of scf0 and one of scf2 filled, one gap each left open, scf1 without gaps)."""
    import gzip
    import json
    from gappadder_amd.put_gap_seq_back_to_scaffold import put_gap_seq_back_to_scaffold
    case = Case("twolib")
    wb = json.loads(gzip.open(os.path.join(case.dir, "writeback.json.gz")).read())
    d = str(tmp_path)
    open(d + "/draft.fa", "w").write(case.draft_fa)
    open(d + "/draft.fa.fai", "w").write(case.fai)
    open(d + "/gap_positions.txt", "w").write(case.expected["gap_positions.txt"])
    open(d + "/picked.fa", "w").write(wb["picked_fa"])
    put_gap_seq_back_to_scaffold(d + "/draft.fa", d + "/gap_positions.txt", d + "/picked.fa", d + "/new.fa")
    got = open(d + "/new.fa").read()
    assert got == wb["new_scaffolds_fa"]
    assert got.count(">") == 3 and got.count("N") == 600    # the two open gaps + the 50-N run below min_gap_size


# ---- flank anchoring (pick_contigs.py:64-358, 361-539 with exact anchors) ----

def _rnd(rng, n):
    return "".join("ACGT"[i] for i in rng.integers(0, 4, size=n))


def test_host_picker_equals_the_oracle_on_exact_anchor_hits():
    """gappadder_amd.pick_contigs (hit tuples) against oracle/gp_oracle.py::pick_gap / pick_gap_extended — the reference's selection,
    pinned on its own answers, applied to the SAM text of the exact-anchor stand-in — full and extended pick, anchors 30 and 15."""
    from gappadder_amd.pick_contigs import pick_extended_sequence, pick_gap_sequence
    from oracle import gp_oracle as O
    import pick_util as PK
    n_full = n_ext = 0
    for g, (l, r, seqs) in enumerate(PK.picker_cases(21, 400)):
        contigs = [("c%d" % i, s) for i, s in enumerate(seqs)]
        gid = "0_%d" % (g + 1)
        for a in (30, 15):
            res = pick_gap_sequence(contigs, l, r, a)
            mine = (None, None) if res is None else (">%s_%s\n%s\n" % (gid, res[0], res[1]), ">%s_%s\n%s\n" % (gid, res[0], res[2]))
            assert mine == O.pick_gap(gid, contigs, l, r, a), (g, a)
            n_full += res is not None
            res = pick_extended_sequence(contigs, l, r, a)
            if res is None:
                mine = (None, None)
            else:
                hdr = ">%s_%s_%s_extended\n" % (gid, res[0], res[1])
                mine = tuple(hdr + x + "\n" if x is not None else None for x in res[2:])
            assert mine == O.pick_gap_extended(gid, contigs, l, r, a), (g, a)
            n_ext += res is not None and res[2] is not None
    assert n_full > 200 and n_ext > 300


def test_pick_quirks_of_the_reference_slices():
    import numpy as np
    from gappadder_amd.pick_contigs import pick_extended_sequence, pick_gap_sequence, revcomp
    rng = np.random.default_rng(3)
    left, right = _rnd(rng, 100), _rnd(rng, 100)
    la, ra = left[-30:], right[:30]
    gap = _rnd(rng, 200)
    c1 = _rnd(rng, 10) + la + _rnd(rng, 20) + la + gap + ra + _rnd(rng, 15) + ra + _rnd(rng, 5)
    name, seq, oriented = pick_gap_sequence([("a", c1)], left, right, 30)
    i, j = c1.find(la) + 30, c1.rfind(ra)
    assert seq == c1[i:j + 1] and oriented == c1            # forward: the slice keeps the first base of the right anchor (:341-349)
    name, seq_rc, oriented = pick_gap_sequence([("a", revcomp(c1))], left, right, 30)
    assert seq_rc == c1[i - 1:j] and oriented == c1         # reverse: it keeps the last base of the LEFT anchor instead (:343-345)
    c2 = la + gap[:50] + ra
    assert pick_gap_sequence([("short", c2), ("long", c1)], left, right, 30)[0] == "long"      # longest span (:313-321)
    assert pick_gap_sequence([("x", c2), ("y", c2)], left, right, 30)[0] == "x"                # ties: the first
    assert pick_gap_sequence([("w", ra + gap + la)], left, right, 30) is None                  # wrong order
    assert pick_gap_sequence([("o", la[:-5] + ra)], left, right, 30) is None                   # overlapping
    assert pick_gap_sequence([("m", la + gap)], left, right, 30) is None                       # one anchor
    assert pick_gap_sequence([("z", la + ra)], left, right, 30)[1] == ra[0]                    # span 0: one base
    c3 = left[-15:] + gap + right[:15]
    assert pick_gap_sequence([("c", c3)], left, right, 30) is None and pick_gap_sequence([("c", c3)], left, right, 15)[1] == gap + right[0]
    # extended (:361-539): left part + 'NN' + right part; the FIRST contig with a hit per side; right-only picked_contigs = 'NN' + contig
    into_gap, out_of_gap = _rnd(rng, 120), _rnd(rng, 90)
    cl, cr = left[-40:] + into_gap, out_of_gap + right[:50]
    assert pick_extended_sequence([("L", cl), ("R", cr)], left, right, 15) == ("L", "R", into_gap + "NN" + out_of_gap, cl + "NN" + cr)
    assert pick_extended_sequence([("L", cl)], left, right, 15) == ("L", "", into_gap + "NN", cl)
    assert pick_extended_sequence([("R", cr)], left, right, 15) == ("", "R", "NN" + out_of_gap, "NN" + cr)
    assert pick_extended_sequence([("s", left[-15:] + "ACGTA"), ("L", cl)], left, right, 15)[0] == "s"
    assert pick_extended_sequence([("R", revcomp(cr))], left, right, 15)[2] == "NN" + out_of_gap
    assert pick_extended_sequence([("L", revcomp(cl))], left, right, 15)[2] == left[-1] + into_gap + "NN"   # reverse left: one anchor base rides along (:496)
    both = out_of_gap[:25] + right[:15] + _rnd(rng, 30) + left[-15:] + into_gap[:10]          # one contig, anchors out of order
    assert pick_extended_sequence([("b", both)], left, right, 15)[:3] == ("b", "b", "NN" + out_of_gap[:25] + right[0])
    assert pick_extended_sequence([("n", _rnd(rng, 300))], left, right, 15) is None


def test_extended_pick_writes_the_reference_files(tmp_path):
    from gappadder_amd.pick_contigs import ContigsSelection
    wf = str(tmp_path / "wf" / "merged") + "/"
    os.makedirs(wf + "velvet_temp/0_1")
    os.makedirs(str(tmp_path / "wf" / "flank_regions"))
    import numpy as np
    rng = np.random.default_rng(5)
    left, right, fill = _rnd(rng, 60), _rnd(rng, 60), _rnd(rng, 70)
    with open(str(tmp_path / "wf" / "flank_regions" / "0_1.fa"), "w") as f:
        f.write(">0_1_left\n%s\n>0_1_right\n%s\n" % (left, right))
    with open(wf + "velvet_temp/0_1/contigs.fa", "w") as f:
        f.write(">31_29_NODE_1_length_60_cov_3.000000\n%s\n" % (left[-20:] + fill))
    cs = ContigsSelection(wf)
    sf = wf + "../picked_seqs.fa"
    assert cs.pick_full_constructed_contigs(15, ["0_1"], sf) == 0
    assert cs.pick_extended_contigs(15, ["0_1"], sf) == 1
    txt = open(sf).read()
    assert txt == ">0_1_31_29_NODE_1_length_60_cov_3.000000__extended\n%sNN\n" % fill
    assert "0_1" in cs.get_already_picked(sf)


def test_truth_check_of_closed_gaps_catches_wrong_fills():
    """bench.py::truth_check (the `gaps_closed_correct` count): a picked sequence equal to the true bases behind the planted gap counts
    as correct on either strand; one substituted or one missing base inside the gap does not."""
    import numpy as np
    import bench
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd.pick_contigs import revcomp
    cfg = GapFill.synth_cfg(seed=77, scaffold_len=200_000, n_scaffolds=2, gaps_per_scaffold=3, gap_len=500)
    gaps, flanks = GapFill.synth_layout(cfg)
    contigs, words = [], []
    for g in range(len(gaps)):
        st, en, sc = int(gaps[g]["start"]), int(gaps[g]["end"]), int(gaps[g]["scaffold"])
        t = GapFill.synth_truth(cfg, sc, st - 100, en - st + 200)
        assert t[100 - 5 - 30:100 - 5] == flanks[g][0][-30:] and t[en - st + 105:en - st + 135] == flanks[g][1][:30]
        if g == 1:
            t = t[:300] + ("A" if t[300] != "A" else "C") + t[301:]           # a substitution inside the gap
        if g == 2:
            t = t[:350] + t[351:]                                             # a missing base
        rev = g in (3, 4)
        if g == 4:
            t = t[:222] + ("G" if t[222] != "G" else "T") + t[223:]
        contigs.append(revcomp(t) if rev else t)
        words.append((30 << 56) | ((len(t) - 200 + 11) << 32) | ((0x7FFFFFFF - g) << 1) | int(rev))
    words[5] = 0                                                              # an open gap is not looked at
    ctg = np.zeros(len(contigs), dtype=B.CONTIG)
    off = 0
    for i, s in enumerate(contigs):
        ctg[i] = (i, 31, 29, len(s) - 28, len(s), 0, 0, off)
        off += len(s)
    r = bench.truth_check(cfg, gaps, flanks, ctg, "".join(contigs).encode(), np.array(words, dtype=np.uint64), GapFill)
    assert r["closed"] == 5 and r["correct"] == 2 and sorted(w["gap"] for w in r["wrong"]) == [1, 2, 4]
    assert r["causes"] == {"substitutions": 2, "length -1": 1}


# ---- contig merging, host side (MergeContigs.py; GraphUtils.cpp:625-859, ContigsCompactor.cpp:1422-1520) ----

class _OracleEvaluator:
    """Stand-in for the two GPU calls of MergeContigs on a box without a GPU: the oracle's prefilter and overlap evaluation."""
    def quick_check(self, sets, k=10):
        from gappadder_amd import _lib as B
        from oracle import c_oracle as CO
        rows = [(s, i, j) for s, cs in enumerate(sets) for i, j in CO.quick_check(cs, k)]
        out = np.zeros(len(rows), dtype=B.QCPAIR)
        for x, r in enumerate(rows):
            out[x] = r
        return out

    def overlap_evaluate(self, sets, pairs, params=None, relax=False):
        from gappadder_amd import _lib as B
        from oracle import c_oracle as CO
        out = np.zeros(len(pairs), dtype=B.OVL_RESULT)
        for x, p in enumerate(pairs):
            nodes = CO.merger_nodes(sets[int(p["set"])])
            r = CO.overlap_evaluate(nodes[int(p["i"])], nodes[int(p["j"])], params or CO.GAPPADDER_OVL, relax=relax)
            out[x] = tuple(r[n] for n in B.OVL_RESULT.names)
        return out


import numpy as np  # noqa: E402


def test_path_search_and_merged_strings_equal_the_reference(tmp_path):
    """MergeContigs.find_paths / merged_strings / merge_contigs (host logic; the GPU calls replaced by the oracle's evaluation) on
    the contig sets the reference's own ContigsMerger answered (tests/golden/merger_kat.json.gz): same paths, same merged contigs,
    same merge.info, and the file protocol of MergeContigs.py:66-99."""
    import gzip
    import json
    from golden_util import GOLDEN
    from gappadder_amd import MergeContigs as MC
    from oracle import c_oracle as CO, gp_oracle as O
    cases = json.loads(gzip.open(os.path.join(GOLDEN, "merger_kat.json.gz")).read())
    wf = str(tmp_path) + "/"
    ids = []
    for ci, c in enumerate(cases):
        _, adj = O.merger_edges(c["contigs"], CO.GAPPADDER_OVL)
        assert MC.find_paths(adj) == O.merger_paths(adj), ci
        gid = "0_%d" % (ci + 1)
        ids.append(gid)
        os.makedirs(wf + "velvet_temp/" + gid)
        with open(wf + "velvet_temp/%s/contigs.fa" % gid, "w") as f:
            f.write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(c["contigs"])))
    done = MC.merge_contigs(_OracleEvaluator(), wf, ids + ["missing"])
    assert set(done) == set(ids)
    n_direct = n_shrunk = 0
    for ci, (gid, c) in enumerate(zip(ids, cases)):
        d = wf + "velvet_temp/%s/" % gid
        nodup = MC.drop_contained([("c%d" % i, s) for i, s in enumerate(c["contigs"])])
        exp = O.merger_new_contigs([s for _, s in nodup], CO.GAPPADDER_OVL)
        merged = MC.read_fasta(d + "contigs.fa_no_dup.fa.merged.fa")
        new = [s for n, s in merged if n.startswith("NEW_CONTIG_MERGE_")]
        assert new == [s for _, s in exp] and done[gid] == len(exp), ci
        assert [(n, s) for n, s in merged if not n.startswith("NEW_")] == nodup
        if len(nodup) == len(c["contigs"]):          # nothing contained: the reference binary saw the same set
            assert new == [x["seq"] for x in c["new"]], ci
            info = [l.split()[1:] for l in open(d + "contigs.fa_no_dup.fa.merge.info").read().splitlines()]
            assert info == [x["path"] for x in c["new"]], ci
            n_direct += 1
        assert MC.read_fasta(d + "original_contigs_before_merging.fa") == [("c%d" % i, s) for i, s in enumerate(c["contigs"])]
        final = MC.read_fasta(d + "contigs.fa")
        assert final == MC.drop_contained(merged)
        n_shrunk += len(final) < len(merged)         # the exact pieces of a merged contig lie inside it: they are gone
    assert n_direct >= 30 and n_shrunk >= 20


def test_exact_dedup_and_the_merge_size_guard(tmp_path):
    from gappadder_amd import MergeContigs as MC
    from gappadder_amd.pick_contigs import revcomp
    rng = np.random.default_rng(8)
    a, b = _rnd(rng, 300), _rnd(rng, 200)
    recs = [("a", a), ("inside", a[50:150]), ("b", b), ("a_again", a), ("b_rc", revcomp(b)), ("inside_rc", revcomp(a[100:260])), ("c", _rnd(rng, 90))]
    assert [n for n, _ in MC.drop_contained(recs)] == ["a", "b", "c"]          # of identical contigs the first stays
    # a de-duplicated file above 1 MB is not merged (MergeContigs.py:70-74): it becomes contigs.fa as it is
    wf = str(tmp_path) + "/"
    os.makedirs(wf + "velvet_temp/0_1")
    big = [("n%d" % i, _rnd(rng, 8000)) for i in range(130)]
    with open(wf + "velvet_temp/0_1/contigs.fa", "w") as f:
        f.write("".join(">%s\n%s\n" % r for r in big + [("dup", big[0][1][:500])]))

    class NoGpu:
        def __getattr__(self, name):
            raise AssertionError("GPU touched")
    assert MC.merge_contigs(NoGpu(), wf, ["0_1"]) == {"0_1": 0}
    assert MC.read_fasta(wf + "velvet_temp/0_1/contigs.fa") == big
    assert len(MC.read_fasta(wf + "velvet_temp/0_1/original_contigs_before_merging.fa")) == 131


def test_high_quality_reads_that_bridge_two_contigs_are_appended(tmp_path):
    """collect_high_quality_unmap_to_contigs_reads (assemble_gaps.py:166-217 with exact matching for bwa): the assembly's own
    contigs come back and the reads clipped at two contigs follow them as FASTA records."""
    from gappadder_amd import assemble_gaps as AG
    from gappadder_amd.pick_contigs import read_fasta, revcomp
    rng = np.random.default_rng(9)
    g = _rnd(rng, 1000)
    c1, c2, c3 = g[0:400], g[440:800], _rnd(rng, 300)
    wf = str(tmp_path) + "/"
    os.makedirs(wf + "velvet_temp/0_1")
    os.makedirs(wf + "gap_reads_high_quality")
    open(wf + "velvet_temp/0_1/contigs.fa", "w").write(">merged_1\n%s\n>merged_2\n%s\n>merged_3\n%s\n" % (c1, c2, g[60:330]))
    open(wf + "velvet_temp/0_1/original_contigs_before_merging.fa", "w").write(">o1\n%s\n>o2\n%s\n>o3\n%s\n" % (c1, c2, c3))
    # ("inside_err": a read that lies inside BOTH overlapping contigs of a second pair with one sequencing error — an end-to-end alignment
    #  for bwa, no bridge: ADVICE r3)
    reads = [("bridge", g[350:500]), ("inside", g[100:250]), ("one_side", g[380:430] + _rnd(rng, 100)), ("bridge_rc", revcomp(g[360:510])),
             ("bridge", g[0:150]), ("short_overlap", g[390:400] + _rnd(rng, 140)),
             ("inside_err", g[120:190] + ("A" if g[190] != "A" else "C") + g[191:270])]
    open(wf + "gap_reads_high_quality/0_1.fastq", "w").write("".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in reads))
    ga = AG.GapAssembler("x.fai", "x.pos", 1, wf, kmer_list=[(31, 29)], gf=object())
    assert ga.collect_high_quality_unmap_to_contigs_reads(["0_1", "0_2"]) == 2
    got = read_fasta(wf + "velvet_temp/0_1/contigs.fa")
    assert got == [("o1", c1), ("o2", c2), ("o3", c3), ("bridge", g[350:500]), ("bridge_rc", revcomp(g[360:510]))]
    assert not os.path.exists(wf + "velvet_temp/0_1/original_contigs_before_merging.fa")


def test_open_gap_census_tells_the_causes_apart():
    """bench.py::open_gap_census on hand-built contig sets: a spanning contig, a contig whose copy of an anchor differs, a hole in the
    coverage, two contigs that overlap without being joined, and a gap without contigs each land in their own class."""
    import numpy as np
    import bench
    from gappadder_amd import _lib as B
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd.pick_contigs import revcomp
    cfg = GapFill.synth_cfg(seed=78, scaffold_len=200_000, n_scaffolds=1, gaps_per_scaffold=6, gap_len=500)
    gaps, _ = GapFill.synth_layout(cfg)
    sets = []
    for g in range(6):
        st, en = int(gaps[g]["start"]), int(gaps[g]["end"])
        t = GapFill.synth_truth(cfg, 0, st - 100, en - st + 200)        # t[100] = first gap base; anchors t[65:95], t[605:635]
        if g == 0: sets.append([revcomp(t)])                            # everything on one contig (reverse strand)
        if g == 1: sets.append([t[:80] + ("A" if t[80] != "A" else "C") + t[81:]])        # a substitution inside the left anchor
        if g == 2: sets.append([t[:330], t[370:]])                      # 40 true bases in no contig
        if g == 3: sets.append([t[:400], t[340:]])                      # overlapping halves, never joined
        if g == 4: sets.append([])
        if g == 5: sets.append([t])                                     # closed: not part of the census
    flat = [(g, s) for g, cs in enumerate(sets) for s in cs]
    ctg = np.zeros(len(flat), dtype=B.CONTIG)
    off = 0
    for i, (g, s) in enumerate(flat):
        ctg[i] = (g, 31, 29, len(s) - 28, len(s), 0, 0, off)
        off += len(s)
    best = np.zeros(6, dtype=np.uint64)
    best[5] = 1
    r = bench.open_gap_census(cfg, gaps, ctg, "".join(s for _, s in flat).encode(), best, GapFill)
    assert r["open_gaps"] == 5 and r["classified"] == 5
    assert (r["spanning_contig_unpicked"], r["anchor_differs"], r["coverage_hole"], r["fragmented"], r["no_contigs"]) == (1, 1, 1, 1, 1), r


def test_roofline_traffic_is_quoted_only_from_a_profile_of_the_running_build(tmp_path, monkeypatch):
    """bench.py::pmc_traffic: the committed rocprofv3 summary must name exactly the kernels the library launched (gf_screen_kernels,
    template arguments included), describe the same workload, and its launch times must be within 15 % of the run's — otherwise the
    line carries `traffic: null` instead of a stale ratio."""
    import json
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    names = ["void gf::pf4_scatter_lines_kernel<3u, false>(gf::Part4Params, unsigned int)", "gf::pf4_probe_kernel(gf::Part4Params)",
             "gf::pf4_resolve_kernel(gf::Part4Params)", "gf::pf4_list_kernel(gf::Part4Params)"]
    t = {"reads_per_launch": 900_000_000, "read_len": 150, "k": 51, "kernels": {n: {} for n in names}, "traffic_bytes_per_launch": 7.4e10,
         "rocprof_avg_launch_ns_sum": 15.4e6}
    (prof / "r04_traffic_c4.json").write_text(json.dumps(t))
    launched = "pf4_scatter_lines_kernel<3u, false>,pf4_probe_kernel,pf4_resolve_kernel,pf4_list_kernel"
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 15.2, launched) == 7.4e10
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 12.0, launched) is None                       # this run is 22 % faster: another build
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 15.2, launched.replace("false", "true")) is None    # another instantiation of pass A
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 15.2, launched.replace("pf4_scatter_lines", "pf4_scatter")) is None
    assert bench.pmc_traffic("C4", 450_000_000, 150, 51, 15.2, launched) is None                      # another workload
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 15.2, "") is None
    older = dict(t, kernels={n.replace("false", "true"): {} for n in names})                              # an older round's file beside it: skipped, the newer one still counts
    (prof / "r03_traffic_c4.json").write_text(json.dumps(older))
    assert bench.pmc_traffic("C4", 900_000_000, 150, 51, 15.2, launched) == 7.4e10


def test_a_read_that_fits_at_a_second_occurrence_of_its_seed_is_no_bridge(tmp_path):
    """A contig with an internal repeat: a read that lies wholly inside the SECOND copy shares its seeds with the first copy
    too, where its extension runs off or mismatches — bwa reports the end-to-end alignment, so the read is not clipped at that contig."""
    from gappadder_amd import assemble_gaps as AG
    from gappadder_amd.pick_contigs import read_fasta
    rng = np.random.default_rng(21)
    rep = _rnd(rng, 60)
    a, b, c = _rnd(rng, 120), _rnd(rng, 150), _rnd(rng, 200)
    c1 = a + rep + b + rep + c                      # the repeat twice, different neighbourhoods
    c2, c3 = _rnd(rng, 300), _rnd(rng, 300)
    wf = str(tmp_path) + "/"
    os.makedirs(wf + "velvet_temp/0_1")
    os.makedirs(wf + "gap_reads_high_quality")
    fa = ">m1\n%s\n>m2\n%s\n>m3\n%s\n" % (c1, c2, c3)
    open(wf + "velvet_temp/0_1/contigs.fa", "w").write(fa)
    second = len(a) + len(rep) + len(b)             # start of the second copy
    inside_second = c1[second - 40:second + 100]    # fits end to end there; at the first copy its flanks mismatch
    real_bridge = c1[-80:] + c2[:70]                # clipped at m1 and at m2
    half = c2[-75:] + _rnd(rng, 75)                 # clipped at m2 only
    reads = [("inside_second", inside_second), ("real_bridge", real_bridge), ("half", half), ("rep_and_m3", inside_second[:90] + c3[:60])]
    open(wf + "gap_reads_high_quality/0_1.fastq", "w").write("".join("@%s\n%s\n+\n%s\n" % (n, s, "I" * len(s)) for n, s in reads))
    ga = AG.GapAssembler("x.fai", "x.pos", 1, wf, kmer_list=[(31, 29)], gf=object())
    assert ga.collect_high_quality_unmap_to_contigs_reads(["0_1"]) == 2
    got = [n for n, _ in read_fasta(wf + "velvet_temp/0_1/contigs.fa")]
    assert got == ["m1", "m2", "m3", "real_bridge", "rep_and_m3"]


def test_bridging_read_placements_by_sorting_equal_the_window_lookup():
    """assemble_gaps._placements_by_sort (2-bit window values joined by sorting: the form a run uses for ACGT-only texts) against
    _placements_by_lookup (every read window looked up among the contig windows: the definition) — repeats beyond the occurrence cap,
    reads shorter than a seed, both strands, several contigs."""
    import random
    from gappadder_amd import assemble_gaps as A
    from gappadder_amd.pick_contigs import revcomp
    rng = random.Random(31)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    n_placed = 0
    for it in range(120):
        seed_len = rng.choice([6, 30, 32])
        genome = rnd(rng.randint(100, 500))
        if it % 3 == 0:
            genome = genome[:40] + rnd(rng.randint(8, 40)) * rng.randint(2, 12) + genome[40:]
        strands = []
        for _ in range(rng.randint(0, 4)):
            a = rng.randint(0, len(genome) - 20)
            s = genome[a:a + rng.randint(10, 250)]
            strands += [s, revcomp(s)]
        reads = []
        for _ in range(rng.randint(0, 25)):
            a = rng.randint(0, len(genome) - 10)
            s = list(genome[a:a + rng.randint(3, 100)])
            for _ in range(rng.choice([0, 0, 1, 4])):
                s[rng.randrange(len(s))] = rng.choice("ACGT")
            reads.append("".join(s) if rng.random() < 0.5 else revcomp("".join(s)))
        want = A._placements_by_lookup(strands, reads, seed_len)
        assert A._placements_by_sort(strands, reads, seed_len) == want
        n_placed += sum(len(pl) for p in want.values() for pl in p.values())
    assert n_placed > 1000
    # a text with a letter beyond ACGT takes the lookup (N equals N there)
    contigs = [("a", "ACGTNACGTTGCA" * 3), ("b", "TTGACNCATGAC" * 3)]
    reads = {"r1": "GTNACGTTGCAACGTNACG", "r2": "ACNCATGACTTGACNC"}
    assert A.bridging_reads(contigs, reads, 8) == []


def test_both_unmapped_dictionary_and_mate_files_follow_the_line_loop(tmp_path, monkeypatch):
    """collect_both_unmapped_reads builds {head: 'seq\\n+\\nqual\\n'} and the _1 / _2 files column-wise; the reference's line loop
    (collect_both_unmapped_reads.py:205-236) on the same text — heads that repeat (later record, first position), trailing blanks, a
    file that ends inside a record — must give the same dictionary, in the same order, and the same two files."""
    import pytest
    from gappadder_amd import collect_both_unmapped_reads as CB
    text = ("@a_1 \nACGT\n+\nIIII\n@a_2\nTTTT \n+x\nJJJJ\n@b_2\nGG\n+\nII\n@b_1\nCC\t\n+\nKK\n"
            "@a_1\nAAAA\n+\nLLLL\n@c_1\nAC\n+\nII\n@c_2\nGT\n+\nII\n@tail_1\nACG\n+")
    bam = str(tmp_path / "x.bam")
    open(bam + ".both_unmapped.fq", "w").write(text)
    monkeypatch.setattr(CB, "run_collect_both_unmapped", lambda *a, **k: None)
    monkeypatch.setattr(CB.BothUnmappedReadsCollector, "align_unmapped_to_contigs", lambda self, ids: {})
    wf = str(tmp_path) + "/"
    c = CB.BothUnmappedReadsCollector(wf, "builtin", gf=object())
    c.collect_both_unmapped_reads([bam], [])
    want = {}
    lines = text.split("\n")
    for i in range(0, len(lines) - 3, 4):
        want[lines[i].rstrip()[1:]] = "".join(l.rstrip() + "\n" for l in lines[i + 1:i + 4])
    assert list(c.reads.items()) == list(want.items()) and c.reads["a_1"] == "AAAA\n+\nLLLL\n" and "tail_1" not in c.reads
    left = "".join("@" + k[:-2] + "\n" + v for k, v in want.items() if k[-1] == "1")
    right = "".join("@" + k[:-2] + "\n" + want[k[:-2] + "_2"] for k in want if k[-1] == "1")
    assert open(wf + "both_unmapped_1.fq").read() == left and open(wf + "both_unmapped_2.fq").read() == right
    assert open(wf + "both_unmapped.fq").read() == text
    # a `_1` record without its `_2`: KeyError, as in the reference
    open(bam + ".both_unmapped.fq", "w").write("@z_1\nAC\n+\nII\n")
    with pytest.raises(KeyError):
        c.collect_both_unmapped_reads([bam], [])


def test_gzip_and_bgzf_fastq_files_are_inflated_once(tmp_path):
    """fastq_io.plain_fastq (SURVEY.md §8f-4 FASTQ(.gz)): plain gzip, concatenated members and BGZF (a chain of small gzip members with an
    empty one at the end) come back as one plain copy; plain text passes through; the copy is reused while it is newer than its source."""
    import gzip
    import struct
    import zlib
    from gappadder_amd import fastq_io as F
    text = "".join("@r%d/1\n%s\n+\n%s\n" % (i, "ACGT" * 25, "I" * 100) for i in range(3000)).encode()
    plain = tmp_path / "a.fq"
    plain.write_bytes(text)
    assert F.plain_fastq(str(plain), str(tmp_path / "t")) == str(plain) and not (tmp_path / "t").exists()
    gz = tmp_path / "b.fq.gz"
    gz.write_bytes(gzip.compress(text[:100000]) + gzip.compress(text[100000:]))          # two members (`cat x.gz y.gz`)

    def bgzf(data, block=30000):
        out = b""
        for a in list(range(0, len(data), block)) + [len(data)]:      # the last, empty block = BGZF's end-of-file marker
            raw = data[a:a + block] if a < len(data) else b""
            co = zlib.compressobj(6, zlib.DEFLATED, -15)
            body = co.compress(raw) + co.flush()
            bsize = 12 + 6 + len(body) + 8
            out += b"\x1f\x8b\x08\x04" + b"\0" * 4 + b"\x00\xff" + struct.pack("<H", 6) + b"BC" + struct.pack("<HH", 2, bsize - 1) + body + \
                struct.pack("<II", zlib.crc32(raw), len(raw))
        return out
    bz = tmp_path / "c.fastq.gz"
    bz.write_bytes(bgzf(text))
    for src in (gz, bz):
        out = F.plain_fastq(str(src), str(tmp_path / "t"))
        assert out != str(src) and open(out, "rb").read() == text
        m = os.path.getmtime(out)
        assert F.plain_fastq(str(src), str(tmp_path / "t")) == out and os.path.getmtime(out) == m      # reused, not written again
