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


def test_pick_gap_sequence_anchors_and_quirks():
    from gappadder_amd.pick_contigs import pick_gap_sequence, revcomp
    import numpy as np
    rng = np.random.RandomState(3)
    g = "".join("ACGT"[i] for i in rng.randint(0, 4, 1200))
    left, gap, right = g[100:395], g[395:605], g[605:900]      # flanks exclude 5 bp next to the gap (gnrt_pos_true_seqs.py:94-99)
    contig = g[50:950]
    for c in (contig, revcomp(contig)):
        name, seq, oriented = pick_gap_sequence([("short", g[380:500]), ("NODE_1", c)], left, right, 30)
        assert name == "NODE_1" and oriented == contig
        assert seq == gap + right[0]                             # the reference's slice keeps one base of the right flank
    assert pick_gap_sequence([("a", g[50:600])], left, right, 30) is None          # right anchor missing
    assert pick_gap_sequence([("a", contig)], left[:20], right, 30) is None        # flank shorter than the anchor
    two = [("a", g[300:700]), ("b", g[300:395] + "ACGT" * 70 + g[605:700])]        # longest span wins (pick_contigs.py:300-321)
    assert pick_gap_sequence(two, left, right, 30)[0] == "b"


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


def test_pick_takes_the_longest_span_over_all_anchor_occurrences_and_both_orientations():
    import numpy as np
    from gappadder_amd.pick_contigs import pick_gap_sequence, revcomp
    rng = np.random.default_rng(3)
    left, right = _rnd(rng, 100), _rnd(rng, 100)
    la, ra = left[-30:], right[:30]
    gap = _rnd(rng, 200)
    # one contig with the left anchor twice and the right anchor twice: leftmost left, rightmost right (pick_contigs.py:300-321)
    c1 = _rnd(rng, 10) + la + _rnd(rng, 20) + la + gap + ra + _rnd(rng, 15) + ra + _rnd(rng, 5)
    name, seq, oriented = pick_gap_sequence([("a", c1)], left, right, 30)
    i, j = c1.find(la) + 30, c1.rfind(ra)
    assert seq == c1[i:j + 1] and oriented == c1            # the +1: the reference's 1-based / 0-based slice (:341-349)
    # the reverse-complemented contig gives the same gap sequence, reported in flank orientation
    assert pick_gap_sequence([("a", revcomp(c1))], left, right, 30)[1] == seq
    # among contigs the longest span wins, ties go to the first
    c2 = la + gap[:50] + ra
    assert pick_gap_sequence([("short", c2), ("long", c1)], left, right, 30)[0] == "long"
    assert pick_gap_sequence([("x", c2), ("y", c2)], left, right, 30)[0] == "x"
    # anchors in the wrong order, overlapping, or missing: not closed
    assert pick_gap_sequence([("w", ra + gap + la)], left, right, 30) is None
    assert pick_gap_sequence([("o", la[:-5] + ra)], left, right, 30) is None
    assert pick_gap_sequence([("m", la + gap)], left, right, 30) is None
    # a shorter anchor (the reference's second score, 15) finds what 30 misses when a base of the anchor differs
    c3 = left[-15:] + gap + right[:15]
    assert pick_gap_sequence([("c", c3)], left, right, 30) is None and pick_gap_sequence([("c", c3)], left, right, 15)[1] == gap + right[0]


def test_extended_pick_joins_partial_fills_with_NN():
    import numpy as np
    from gappadder_amd.pick_contigs import pick_extended_sequence, revcomp
    rng = np.random.default_rng(4)
    left, right = _rnd(rng, 80), _rnd(rng, 80)
    la, ra = left[-15:], right[:15]
    into_gap, out_of_gap = _rnd(rng, 120), _rnd(rng, 90)
    cl = left[-40:] + into_gap                      # reaches 120 bases into the gap from the left
    cr = revcomp(out_of_gap + right[:50])           # reaches 90 bases into the gap from the right, written reverse-complemented
    ln, rn, seq, txt = pick_extended_sequence([("L", cl), ("R", cr), ("junk", _rnd(rng, 200))], left, right, 15)
    assert (ln, rn) == ("L", "R") and seq == into_gap + "NN" + out_of_gap and txt == cl + "NN" + cr
    # only one side reachable
    assert pick_extended_sequence([("L", cl)], left, right, 15)[:3] == ("L", "", into_gap + "NN")
    assert pick_extended_sequence([("R", cr)], left, right, 15)[:3] == ("", "R", "NN" + out_of_gap)
    # the longer extension wins per side
    cl2 = la + into_gap[:30]
    assert pick_extended_sequence([("s", cl2), ("L", cl)], left, right, 15)[0] == "L"
    # one contig hit by both anchors (but not in order): the right side only (pick_contigs.py:468-486, equal match lengths)
    pre = _rnd(rng, 25)
    both = pre + ra + _rnd(rng, 30) + la + into_gap[:10]
    assert pick_extended_sequence([("b", both)], left, right, 15)[:3] == ("", "b", "NN" + pre)
    assert pick_extended_sequence([("b", ra + _rnd(rng, 30) + la)], left, right, 15) is None      # nothing but 'NN' to report
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
