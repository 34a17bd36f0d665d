"""The CLI on FILES of the synthetic workload (tools/synth_files) against the device-resident run of the same workload: the path
bench.py times and the path a user invokes must recruit the same reads and close the same gaps."""
import os

import numpy as np
import pytest

import synth_files_util as SF

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("record_reads", ["pread", "mmap"])
def test_cli_on_files_equals_the_device_resident_run(tmp_path, record_reads, monkeypatch):
    import torch
    monkeypatch.setenv("GF_RECORD_READS", record_reads)      # how the pooled FASTQ records are fetched from the input files (device_collect._RecordFiles)
    from gappadder_amd import main as M
    from gappadder_amd.hip_api import GapFill
    from gappadder_amd.pipeline import DeviceLibrary, Pipeline
    seed, slen, nscf, gps, glen, L = 20260007, 400_000, 3, 4, 120, 150
    libs = [(300, 30, 60_000), (2000, 200, 20_000)]
    kk = [(31, 29), (41, 39)]
    cfgp, wf = SF.write_case(str(tmp_path), seed, slen, nscf, gps, glen, libs, kk, kmer_screen=31)
    M.main(["-c", "All", "-g", cfgp])
    # the same workload synthesised straight into HBM, through the same Pipeline
    gf = GapFill(0)
    cfg0 = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L)
    gaps, flanks = GapFill.synth_layout(cfg0)
    gf.set_gaps(gaps, nscf, flanks)
    pipe = Pipeline(gf, len(gaps), L, kk, keep_read_ids=True)
    rb = 38
    for no, (is_, sd, n_pairs) in enumerate(libs):
        cfg = GapFill.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=nscf, gaps_per_scaffold=gps, gap_len=glen, read_len=L, insert_mean=is_,
                                insert_sd=sd, library=no)
        d_reads = torch.empty(2 * n_pairs * rb + 64, dtype=torch.uint8, device="cuda")
        d_recs = torch.empty(2 * n_pairs * 32, dtype=torch.uint8, device="cuda")
        gf.synth_pairs_dev(cfg, 0, n_pairs, d_reads.data_ptr(), d_recs.data_ptr())
        pipe.add_library(DeviceLibrary("lib%d" % no, is_, sd, 2 * n_pairs, d_reads, d_recs))
    gf.sync()
    pipe.prepare()
    pipe.step()
    res = pipe.fetch()
    keys = ["%d_%d" % (int(g["scaffold"]), int(g["idx_in_scaffold"])) for g in gaps]
    n_pooled = 0
    for no, lb in enumerate(pipe.libs):
        off = lb.d_pool_off.cpu().numpy()
        ids = lb.d_ids[:int(off[-1])].cpu().numpy()
        folder = "%s%d_is%d/gap_reads/" % (wf, no + 1, libs[no][0])
        for g, key in enumerate(keys):
            want = ["@r%d_%d" % (i >> 1, (i & 1) + 1) for i in ids[int(off[g]):int(off[g + 1])]]
            got = open(folder + key + ".fastq").read().splitlines()[0::4] if os.path.exists(folder + key + ".fastq") else []
            assert got == want, (no, key)
            n_pooled += len(want)
    assert n_pooled > 1500
    # closed gaps: the first pick of the CLI (merge in between may close more) holds at least the device's closed set, with its sequences
    picked = {}
    for blk in open(wf + "picked_seqs.fa").read().split(">")[1:]:
        h, s = blk.split("\n", 1)
        picked["_".join(h.split("_")[:2])] = s.replace("\n", "")
    closed = [keys[g] for g in np.nonzero(res.best)[0]]
    assert len(closed) >= 6 and set(closed) <= set(picked)
    truth_ok = 0
    for g in np.nonzero(res.best)[0]:
        st, en, sc = int(gaps[g]["start"]), int(gaps[g]["end"]), int(gaps[g]["scaffold"])
        t = (GapFill.synth_truth(cfg0, sc, st - 5, en - st + 11), GapFill.synth_truth(cfg0, sc, st - 6, en - st + 11))
        truth_ok += picked[keys[g]] in t
    assert truth_ok >= len(closed) - 1
    # the contig-merge round for the gaps the step left open (Pipeline.merge_open_gaps): only open gaps, every pick confirmed by the host picker
    from gappadder_amd.pick_contigs import pick_gap_sequence
    mg = pipe.merge_open_gaps(res)
    open_gaps = set(int(g) for g in np.nonzero(res.best == 0)[0])
    assert set(mg["closed"]) <= open_gaps and mg["gaps_tried"] <= len(open_gaps)
    for g, (a_len, span1, ci, rev) in mg["closed"].items():
        gg, contig = mg["contigs"][ci]
        assert gg == g
        r = pick_gap_sequence([("c", contig)], flanks[g][0], flanks[g][1], a_len)
        assert r is not None and len(r[1]) == span1 and (r[2] != contig) == bool(rev), g


def test_bench_e2e_extra_at_full_c3_size():
    """bench.py's `e2e_files_C3` extra as the driver runs it: the CLI (`-c All`, builtin BAM, k-mer screen on) on the C3-sized files recruits
    exactly what the device-resident run of C3 recruits (5 M reads: screen, tagger, second hop, pool keys, pooled reads) and closes every gap."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--e2e-only", "C3"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-800:]
    d = json.loads(r.stdout.decode().strip().splitlines()[-1])
    assert "error" not in d, d
    assert d["same_recruits_as_the_device_resident_run"] is True
    assert d["reads"] == 5_000_000 and d["gaps"] == 200 and d["picked_seqs"] == 200
    assert d["libraries"][0]["screen_hits"] > 100_000 and d["libraries"][0]["pooled_reads"] > 200_000
    assert set(d["device_collect_s"]) >= {"ingest_fastq", "ingest_bam", "join", "recruit_and_sizing", "pools", "assemble_and_pick", "write_files"}


def test_both_unmapped_files_of_the_single_pass_equal_the_second_pass(tmp_path):
    """`-c All` on the device path writes {bam}.both_unmapped.sam / .fq while the BAM goes by (DeviceCollector._ingest_bam) and the second
    round takes them as they are; a run of collect_both_unmapped_reads.run_collect_both_unmapped alone — a second decode of the whole
    file — must write the same bytes, whatever the piece size of the first pass."""
    from gappadder_amd import collect_both_unmapped_reads as CB
    from gappadder_amd import main as M
    from gappadder_amd.hip_api import GapFill
    cfgp, wf = SF.write_case(str(tmp_path), 20260011, 300_000, 2, 3, 1000, [(300, 30, 40_000)], [(31, 29)], kmer_screen=31)   # gaps that swallow whole pairs
    bam = os.path.join(str(tmp_path), "data", "lib0.bam")
    os.environ["GF_INGEST_CHUNK_BYTES"] = str(300_000)        # several pieces
    try:
        M.main(["-c", "All", "-g", cfgp])
    finally:
        del os.environ["GF_INGEST_CHUNK_BYTES"]
    first = {ext: open(bam + ".both_unmapped." + ext, "rb").read() for ext in ("sam", "fq")}
    assert first["sam"].count(b"\n") > 200 and first["fq"].count(b"\n") == 4 * first["sam"].count(b"\n")
    assert os.path.abspath(bam) not in CB.PREPARED             # the round took the prepared files (and only once)
    for ext in ("sam", "fq"):
        os.remove(bam + ".both_unmapped." + ext)
    CB.run_collect_both_unmapped(bam, "builtin", GapFill(0))
    for ext in ("sam", "fq"):
        assert open(bam + ".both_unmapped." + ext, "rb").read() == first[ext]
    flags = [int(l.split(b"\t")[1]) for l in first["sam"].splitlines()]
    assert all(f & 12 == 12 for f in flags)
