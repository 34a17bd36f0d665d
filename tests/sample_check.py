"""The full-size checker: does what the GPU left in its output buffers equal the oracle's answers on the sample the oracle
computed?  bench.py's `cpu_baseline` leg calls these after the timed region (`parity_on_sample`), tests/test_gpu_scale.py
asserts on that at every BASELINE configuration's full size — and tests/test_sample_check.py plants a wrong hit, a wrong contig
base and a wrong pick in such buffers and requires every one of these functions to go red (the comparison
code must not be able to hide a bug of its own).  Test infrastructure: the product never imports this."""
import numpy as np

HIT = np.dtype([("gap", "<u4"), ("read", "<u4")])
TAGHIT = np.dtype([("rec", "<u4"), ("gap", "<u4"), ("kind", "<u2"), ("to_mate", "<u2")])


def hits_equal(gpu_hits, oracle_hits, n_sample):
    """Screen hits (gap, read) of the reads [0, n_sample): the GPU's list (any order, all reads) against the oracle's (sample only)."""
    sub = np.sort(np.ascontiguousarray(gpu_hits[gpu_hits["read"] < n_sample]).astype(HIT), order=["gap", "read"])
    want = np.sort(np.ascontiguousarray(oracle_hits).astype(HIT), order=["gap", "read"])
    return len(sub) == len(want) and sub.tobytes() == want.tobytes()


def taghits_equal(gpu_tags, oracle_tags, n_sample):
    """Tagger hits of the records [0, n_sample)."""
    order = ["rec", "gap", "kind", "to_mate"]
    sub = np.sort(np.ascontiguousarray(gpu_tags[gpu_tags["rec"] < n_sample]).astype(TAGHIT), order=order)
    want = np.sort(np.ascontiguousarray(oracle_tags).astype(TAGHIT), order=order)
    return len(sub) == len(want) and sub.tobytes() == want.tobytes()


def _contig_text(ctg, seq, i):
    return seq[int(ctg[i]["seq_off"]):int(ctg[i]["seq_off"]) + int(ctg[i]["length"])].decode()


def contigs_equal(ctg, seq, expected, kk, n_gaps):
    """The device's contig records of gaps [0, n_gaps) (any order) against expected[g][i] = the oracle's [(sequence, n_nodes,
    cov_sum)] of gap g at kk[i]."""
    for g in range(n_gaps):
        for (k, kv), e in zip(kk, expected[g]):
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
    idx = sorted(int(i) for (k, kv) in kk for i in np.nonzero((ctg["gap"] == g) & (ctg["k"] == k) & (ctg["kv"] == kv))[0])
    want = [("c%d" % i, _contig_text(ctg, seq, i)) for i in idx]
    for a_len in (30, 15):
        seqs, ctgs_txt = PO.pick_gap("0_1", want, flanks[g][0], flanks[g][1], a_len)
        if seqs:
            hdr, body = seqs.split("\n")[:2]
            ci = idx[[n for n, _ in want].index(hdr[len(">0_1_"):])]
            rev = int(ctgs_txt.split("\n")[1] != dict(want)["c%d" % ci])
            return (a_len << 56) | (len(body) << 32) | ((0x7FFFFFFF - ci) << 1) | rev
    return 0


def picks_equal(ctg, seq, best, flanks, kk, n_gaps):
    return all(expected_pick_word(ctg, seq, g, flanks, kk) == int(best[g]) for g in range(n_gaps))
