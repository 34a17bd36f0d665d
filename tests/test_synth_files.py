"""tools/synth_files writes the workload of include/gf_synth.h as files: the FASTQ reads and the BAM records must be the ones the
oracle's C generator (same header) produces, the draft the true sequence with the gaps as N-runs."""
import gzip
import struct

import numpy as np

from oracle import c_oracle as CO
import synth_files_util as SF


def test_files_hold_the_synthetic_workload(tmp_path):
    n_pairs, L = 3000, 150
    cfgp, _ = SF.write_case(str(tmp_path), 20260003, 200_000, 2, 5, 1000, [(300, 30, n_pairs)], [(31, 29)])
    data = str(tmp_path) + "/data/"
    cfg = CO.synth_cfg(seed=20260003, scaffold_len=200_000, n_scaffolds=2, gaps_per_scaffold=5, gap_len=1000, read_len=L, insert_mean=300, insert_sd=30)
    packed, recs = CO.synth_pairs(cfg, 0, n_pairs)
    reads = CO.unpack_reads(packed, L)
    for m in (0, 1):
        lines = open(data + "lib0_%d.fq" % (m + 1)).read().splitlines()
        assert len(lines) == 4 * n_pairs
        for p in (0, 1, 17, n_pairs - 1):
            assert lines[4 * p] == "@r%d/%d" % (p, m + 1) and lines[4 * p + 2] == "+" and lines[4 * p + 3] == "I" * L
            assert lines[4 * p + 1].encode() == reads[(2 * p + m) * L:(2 * p + m + 1) * L]
    # the draft: N exactly in the planted gaps
    gaps, flanks = CO.synth_layout(cfg)
    seqs = {}
    for blk in open(data + "draft.fa").read().split(">")[1:]:
        h, s = blk.split("\n", 1)
        seqs[h] = s.replace("\n", "")
    assert list(seqs) == ["scf0", "scf1"] and all(len(s) == 200_000 for s in seqs.values())
    for g, (l, r) in zip(gaps, flanks):
        s = seqs["scf%d" % g["scaffold"]]
        st, en = int(g["start"]), int(g["end"])
        assert s[st:en] == "N" * (en - st) and s[st - 1] != "N" and s[en] != "N"
        assert s[st - 300:st - 5] == l and s[en + 5:en + 300] == r
    # the BAM: coordinate-sorted, every record equal to the generator's (by read name and mate flag)
    d = gzip.open(data + "lib0.bam").read()
    assert d[:4] == b"BAM\x01"
    o = 8 + struct.unpack_from("<i", d, 4)[0]
    n_ref = struct.unpack_from("<i", d, o)[0]
    o += 4
    for _ in range(n_ref):
        o += 8 + struct.unpack_from("<i", d, o)[0]
    by_read = {int(r["read"]): r for r in recs}
    seen, last = 0, (-1, -1)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    while o < len(d):
        bs, ref, pos, l_name, mapq, _bin, n_cig, flag, l_seq, mref, mpos, tlen = struct.unpack_from("<iiiBBHHHiiii", d, o)
        name = d[o + 36:o + 36 + l_name - 1].decode()
        r = by_read[2 * int(name[1:]) + (0 if flag & 0x40 else 1)]
        u = lambda x: -1 if x == 0xFFFFFFFF else int(x)
        assert (ref, mref, flag, mapq, tlen) == (u(r["ref"]), u(r["mate_ref"]), int(r["flag"]), int(r["mapq"]), int(r["tlen"])), name
        assert pos + 1 == int(r["pos"]) or ref < 0
        assert mpos + 1 == int(r["mate_pos"]) or mref < 0
        ops = struct.unpack_from("<%dI" % n_cig, d, o + 36 + l_name)
        cf = (2 if ops and ops[-1] & 15 in (4, 5) else 0) + (1 if ops and ops[0] & 15 in (4, 5) else 0)
        assert cf == int(r["clipflag"]) and (n_cig == 0) == bool(flag & 4)
        assert sum(x >> 4 for x in ops) in (0, L) and l_seq == L
        nib = np.frombuffer(d[o + 36 + l_name + 4 * n_cig:o + 36 + l_name + 4 * n_cig + (L + 1) // 2], dtype=np.uint8)
        seq = "".join("=ACMGRSVTWYHKDBN"[c] for pair in zip(nib >> 4, nib & 15) for c in pair)[:L].encode()
        want = reads[int(r["read"]) * L:(int(r["read"]) + 1) * L]
        assert seq == (want.translate(comp)[::-1] if flag & 0x10 else want), name
        key = (ref if ref >= 0 else 1 << 30, pos)
        assert key >= last
        last = key
        seen += 1
        o += 4 + bs
    assert seen == 2 * n_pairs
