"""The C restatement (oracle/gp_oracle.c, used for bigger parity cases and as cpu_baseline) against the Python
oracle, which is itself pinned on reference outputs (test_oracle_golden.py)."""
import numpy as np
import pytest

from golden_util import CASES, Case
from oracle import c_oracle as CO
from oracle import gp_oracle as O
import records_util as RU


@pytest.fixture(scope="module", params=CASES)
def case(request):
    return Case(request.param)


def test_tagger_matches_python_oracle_and_reference(case):
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    garr = RU.gaps_array(case.fai_names, gaps)
    for lib in case.libs:
        recs, fields = RU.sam_to_records(lib["sam"], case.fai_names)
        hits = CO.tag_alignments(recs, garr, lib["is"], lib["sd"], case.meta["clip_dist"], case.meta["anchor_mapq"])
        got = RU.hits_to_lines(hits, recs, fields, garr, case.fai_names)
        exp = case.exp_dir(lib["folder"] + "/scaffold_reads_list_all/")
        for scf in set(g[3] for g in gaps):
            for side in ("left", "right"):
                e = exp["%s_cluster_by_gap_reads_%s.list" % (scf, side)].splitlines()
                assert sorted(got.get(scf, {}).get(side, [])) == sorted(e), (lib["folder"], scf, side)


def test_low_mapq_matches_reference(case):
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    for lib in case.libs:
        rows = [tuple(int(x) for x in l.split()) for l in case.exp_lines(lib["folder"] + "/discordant_reads_pos.txt.sorted.txt")]
        table = RU.dpos_array(rows)
        recs, fields = RU.sam_to_records(lib["sam"], case.fai_names)
        hits = CO.tag_low_mapq(recs, table)
        got = RU.lowmapq_hits_to_lines(hits, fields, table, case.fai_names)
        exp = case.exp_dir(lib["folder"] + "/discordant_reads_list/")
        for name, txt in exp.items():
            scf, side = name.rsplit("_cluster_by_discordant_reads_", 1)
            side = side.split(".")[0]
            assert got.get(scf, {}).get(side, []) == txt.splitlines(), (lib["folder"], name)


def test_screen_matches_python_oracle(case):
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    seqs = dict(case.fasta_records())
    flanks = [O.flank_seqs(seqs[scf], s, e, case.meta["flank"]) for (s, e, _, scf) in gaps]
    lib = case.libs[0]
    reads = (RU.fastq_seqs(lib["fq1"]) + RU.fastq_seqs(lib["fq2"]))[:400]
    L = len(reads[0])
    for k, mh in ((31, 1), (41, 3)):
        exp = O.screen_reads(reads, flanks, k, mh)
        got = CO.screen_reads("".join(reads).encode(), L, flanks, k, mh)
        assert [(int(h["gap"]), int(h["read"])) for h in got] == exp
        assert len(exp) > 0


def test_pack_kmer64():
    assert CO.lib().or_pack_kmer64(b"ACGTACGTTTGACCA", 5) == O.pack_kmer64("ACGTACGTTTGACCA", 0, 5)
