"""End-to-end on the GPU through the reference's own CLI surface: `main -c Preprocess`, `-c Collect` on the golden inputs
must reproduce the files the reference wrote; `-c Assembly` and the kmc/kmc_dump/velveth/velvetg executables must agree
with the oracle's definition of the assembly."""
import os
import subprocess
import sys

import pytest

from golden_util import CASES, Case
from oracle import c_oracle as CO
import pipeline_util as PU

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", params=CASES)
def run(request, tmp_path_factory):
    from gappadder_amd import main as M
    case = Case(request.param)
    root = str(tmp_path_factory.mktemp(request.param))
    cfgp, wf, st = PU.materialise(case, root, kmers=((31, 29), (41, 39), (41, 38)))
    for stage in ("Preprocess", "Collect", "Assembly"):
        M.main(["-c", stage, "-g", cfgp])
    return case, wf, PU.tree(wf)


def test_collect_reproduces_the_reference_tree(run):
    case, wf, got = run
    checked = 0
    for rel, txt in case.expected.items():
        exact = (rel.endswith(".fastq") or rel.endswith(".sorted.txt") or "discordant_temp/" in rel or rel.endswith(".fa")
                 or rel == "gap_positions.txt" or "/discordant_reads_list/" in rel)
        if exact:
            assert got.get(rel) == txt, rel
        else:   # list files whose line order is dict-defined in the reference: same lines
            assert sorted(got[rel].splitlines()) == sorted(txt.splitlines()), rel
        checked += 1
    assert checked == len(case.expected) and checked > 15
    later = ("merged/velvet_temp/", "picked_seqs.fa",                                  # Assembly stage, first round
             "merged/both_unmapped", "merged/gap_contigs_all.fa", "merged/unmapped_reads/")   # ... second round
    extra = [k for k in got if k not in case.expected and not k.startswith(later)]
    assert not extra, extra


def test_scaffold_lists_keep_sam_record_order(run):
    case, wf, got = run
    for rel, txt in case.expected.items():
        if "/scaffold_reads_list_all/" in rel:
            assert [l.split()[0] for l in got[rel].splitlines()] == [l.split()[0] for l in txt.splitlines()], rel


def test_assembly_stage_files(run):
    from gappadder_amd.assemble_gaps import format_contigs
    case, wf, got = run
    ids = [k[len("merged/gap_reads/"):-len(".fastq")] for k in got if k.startswith("merged/gap_reads/")]
    assert ids
    n_ctg = n_merged_sets = 0
    for gid in ids:
        seqs = [l for i, l in enumerate(got["merged/gap_reads/%s.fastq" % gid].splitlines()) if i % 4 == 1]
        L = max(len(s) for s in seqs)
        blob = "".join(s.ljust(L, "N") for s in seqs).encode()
        merged = ""
        for (k, kv_cfg, kv) in ((31, 29, 29), (41, 39, 39), (41, 38, 37)):     # an even k_velvet runs at kv-1
            exp = format_contigs(CO.assemble_pool(blob, L, k, kv)) if L >= k else ""
            assert got["merged/velvet_temp/%s/contigs_%d_%d.fa" % (gid, k, kv_cfg)] == exp, (gid, k, kv_cfg)
            merged += "".join((">%d_%d_%s" % (k, kv_cfg, l[1:]) if l.startswith(">") else l) for l in exp.splitlines(True))
            n_ctg += exp.count(">")
        # the assembly's contigs.fa is what the merge step set aside (MergeContigs.py:96-99), followed by the bridging high-quality
        # reads of the rescue round, if any (assemble_gaps.py:211-216); contigs.fa itself is the merged set by now
        orig = got.get("merged/velvet_temp/%s/original_contigs_before_merging.fa" % gid, got["merged/velvet_temp/%s/contigs.fa" % gid])
        assert orig.startswith(merged), gid
        tail = orig[len(merged):].splitlines()
        assert len(tail) % 2 == 0 and all(l.startswith(">") for l in tail[0::2]), gid
        n_merged_sets += "merged/velvet_temp/%s/contigs.fa_no_dup.fa.merged.fa" % gid in got
    assert n_ctg > 0 and n_merged_sets > 0


def test_kmc_velvet_executables_follow_the_reference_command_lines(run, tmp_path):
    """The command lines of assemble_gaps.py:96-118, the reference's cvtFaToFq step restated in between."""
    case, wf, got = run
    gid = sorted(k for k in got if k.startswith("merged/gap_reads/"))[0][len("merged/gap_reads/"):-len(".fastq")]
    bind = os.path.join(ROOT, "gappadder_amd", "bin")
    t = str(tmp_path)
    fq = "%smerged/gap_reads/%s.fastq" % (wf, gid)
    env = dict(os.environ, PYTHONPATH=ROOT)
    sh = lambda *a: subprocess.check_call([sys.executable] + list(a), env=env)
    sh(bind + "/kmc", "-k31", "-cs10000000", "-m52", fq, t + "/x.res", t + "/kmc_tmp")
    sh(bind + "/kmc_dump", "-ci0", t + "/x.res", t + "/x.dump")
    dump = open(t + "/x.dump").read().splitlines()
    assert dump and all(len(l.split("\t")[0]) == 31 and int(l.split("\t")[1]) >= 2 for l in dump)
    assert [l.split("\t")[0] for l in dump] == sorted(l.split("\t")[0] for l in dump)
    with open(t + "/kmers.fq", "w") as f:        # cvtFaToFq (assemble_gaps.py:56-79): the whole dump line is the "sequence"
        for i, l in enumerate(dump):
            f.write("@%d\n%s\n+\n%s\n" % (i, l, "5" * len(l)))
    sh(bind + "/velveth", t + "/vdir", "29", "-fastq", "-short", t + "/kmers.fq")
    sh(bind + "/velvetg", t + "/vdir", "-min_contig_lgth", "40")
    assert open(t + "/vdir/contigs.fa").read() == got["merged/velvet_temp/%s/contigs_31_29.fa" % gid]


def test_short_gaps_are_closed_end_to_end(tmp_path):
    """Preprocess -> Collect -> Assembly -> flank anchoring on error-free reads over 110-140 bp gaps: every gap is closed
    and the picked sequence is the true sequence between the flanks (+ the one right-flank base the reference's slice keeps)."""
    import sys as _s
    _s.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    from synth_text import make_case
    from gappadder_amd import main as M

    class C:      # shaped like golden_util.Case for pipeline_util.materialise
        pass
    raw = make_case("closable", 20260021)
    c = C()
    c.draft_fa, c.fai = raw["draft_fa"], raw["fai"]
    c.libs = [{"sam": l["sam"], "fq1": l["fq1"], "fq2": l["fq2"], "is": l["is"], "sd": l["sd"]} for l in raw["libs"]]
    c.meta = {"min_gap": raw["min_gap"], "flank": raw["flank"]}
    cfgp, wf, _ = PU.materialise(c, str(tmp_path), kmers=((31, 29),))
    for stage in ("Preprocess", "Collect", "Assembly"):
        M.main(["-c", stage, "-g", cfgp])
    picked = {}
    for line in open(wf + "picked_seqs.fa").read().split(">")[1:]:
        h, s = line.split("\n", 1)
        picked["_".join(h.split("_")[:2])] = s.strip()
    names = [l.split()[0] for l in raw["fai"].splitlines()]
    gaps = [l.split() for l in open(wf + "gap_positions.txt")]
    assert len(gaps) == 4 and len(picked) == 4
    cnt = {}
    for st, en, _, scf in gaps:
        cnt[scf] = cnt.get(scf, 0) + 1
        gid = "%d_%d" % (names.index(scf), cnt[scf])
        truth = raw["true_seqs"][scf]
        # forward contig: the slice keeps the first base of the right anchor; reverse-complemented contig: the last base of the left
        # anchor instead (pick_contigs.py:341-349)
        assert picked[gid] in (truth[int(st) - 5:int(en) + 6], truth[int(st) - 6:int(en) + 5]), gid


def test_kmer_screen_mode_adds_the_flank_matching_pairs(tmp_path):
    """`"kmer_screen": 31` in the config: gap_reads/{id}.fastq = the reference's alignment-recruited reads UNION the reads (and
    mates) whose canonical 31-mers hit the gap's flanks — checked against the oracle's screen of the same FASTQ files."""
    from gappadder_amd import main as M
    from oracle import gp_oracle as O
    case = Case("twolib")
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmer_screen=31)
    for stage in ("Preprocess", "Collect"):
        M.main(["-c", stage, "-g", cfgp])
    got = PU.tree(wf)
    gaps = O.gap_positions(case.fasta_records(), case.meta["min_gap"])
    ids = O.gap_ids(case.fai_names, gaps)
    seqs = dict(case.fasta_records())
    flanks = [O.flank_seqs(seqs[scf], s, e, case.meta["flank"]) for (s, e, _, scf) in gaps]
    lib = case.libs[0]
    recs = {1: list(O.fastq_records(lib["fq1"])), 2: list(O.fastq_records(lib["fq2"]))}
    L = len(recs[1][0][1])
    want = {gid: set() for gid in ids}
    for m in (1, 2):
        blob = "".join(r[1] for r in recs[m]).encode()
        for h in CO.screen_reads(blob, L, flanks, 31):
            rid = O.fastq_read_id(recs[m][int(h["read"])][0])
            want[ids[int(h["gap"])]].update({rid + "_1", rid + "_2"})
    n_new = 0
    for gid in ids:
        ref = case.expected.get("1_is300/gap_reads/%s.fastq" % gid, "")
        ref_ids = {l[1:] for i, l in enumerate(ref.splitlines()) if i % 4 == 0}
        mine = got.get("1_is300/gap_reads/%s.fastq" % gid, "")
        mine_ids = [l[1:] for i, l in enumerate(mine.splitlines()) if i % 4 == 0]
        assert set(mine_ids) == ref_ids | want[gid], gid
        assert len(mine_ids) == len(set(mine_ids))
        n_new += len(set(mine_ids) - ref_ids)
    assert n_new > 50


def test_both_unmapped_round_recruits_by_contig_kmers(tmp_path):
    """collect_both_unmapped_reads.py with the aligner replaced by the exact k-mer screen: a both-unmapped record is recruited
    for a gap iff it shares a canonical 31-mer with one of the gap's first-round contigs (oracle predicate), its mate comes
    along; the records are appended to gap_reads/{key}.fastq and written to unmapped_reads/{key}.fastq."""
    import numpy as np
    import stat
    from gappadder_amd.collect_both_unmapped_reads import BothUnmappedReadsCollector
    from gappadder_amd.hip_api import GapFill
    rng = np.random.RandomState(5)
    lut = np.frombuffer(b"ACGT", np.uint8)
    comp = bytes.maketrans(b"ACGT", b"TGCA")
    L, k = 100, 31
    rnd = lambda n: lut[rng.randint(0, 4, n)].tobytes().decode()
    keys = ["0_1", "0_2", "3_1"]
    contigs = {key: [rnd(rng.randint(150, 400)) for _ in range(2)] for key in keys}
    wf = str(tmp_path) + "/merged/"
    os.makedirs(wf + "gap_reads")
    for key in keys:
        os.makedirs(wf + "velvet_temp/" + key)
        with open(wf + "velvet_temp/%s/contigs.fa" % key, "w") as f:
            for n, c in enumerate(contigs[key], 1):
                f.write(">31_29_NODE_%d_length_%d_cov_9.000000\n" % (n, len(c) - 28))
                f.write("".join(c[i:i + 60] + "\n" for i in range(0, len(c), 60)))
    open(wf + "gap_reads/0_1.fastq", "w").write("@old_1\nACGT\n+\nIIII\n")
    sam, recs = [], []
    for p in range(240):
        mates = []
        for m in range(2):
            if m == 0 and p % 5 < 3:     # mate 1 of 60 % of the pairs comes from a contig (one substitution, either strand)
                c = contigs[keys[p % 3]][p % 2]
                a = rng.randint(0, len(c) - L + 1)
                s = bytearray(c[a:a + L].encode())
                e = rng.randint(L)
                s[e] = lut[(list(b"ACGT").index(s[e]) + 1) % 4]
                s = bytes(s)
                if p % 2:
                    s = s.translate(comp)[::-1]
                mates.append(s.decode())
            else:
                mates.append(rnd(L))
        for m, flag in ((0, 77), (1, 141)):
            sam.append("q%d\t%d\t*\t0\t0\t*\t*\t0\t0\t%s\t%s\n" % (p, flag, mates[m], "I" * L))
            recs.append(("q%d_%d" % (p, m + 1), mates[m]))
    data = str(tmp_path) + "/data"
    os.makedirs(data)
    bam = data + "/lib0.bam"
    open(bam, "w").close()
    open(bam + ".sam", "w").write("".join(sam))
    st = data + "/samtools_standin.py"
    open(st, "w").write(PU.SAMTOOLS_STANDIN)
    os.chmod(st, os.stat(st).st_mode | stat.S_IEXEC)
    gf = GapFill(0)
    try:
        got = BothUnmappedReadsCollector(wf, samtools_path=st, gf=gf, k=k)
        got.collect_both_unmapped_reads([bam], keys + ["9_9"])
    finally:
        gf.close()
    # oracle: the k-mer predicate on the same records, contigs of a gap joined by N
    blob = "".join(s for _, s in recs).encode()
    hits = CO.screen_reads(blob, L, [("N".join(contigs[key]), "") for key in keys], k)
    total = 0
    for g, key in enumerate(keys):
        idx = set()
        for h in hits:
            if int(h["gap"]) == g:
                i = int(h["read"])
                idx.update((i, i ^ 1))
        exp = "".join("@%s\n%s\n+\n%s\n" % (recs[i][0], recs[i][1], "I" * L) for i in sorted(idx))
        assert open(wf + "unmapped_reads/%s.fastq" % key).read() == exp
        before = "@old_1\nACGT\n+\nIIII\n" if key == "0_1" else ""
        assert open(wf + "gap_reads/%s.fastq" % key).read() == before + exp
        total += len(idx)
    assert total >= 2 * 140 and not os.path.exists(wf + "unmapped_reads/9_9.fastq")


def test_builtin_bam_mode_writes_the_same_tree_as_the_samtools_pipes(run, tmp_path):
    """software_path.samtools = "builtin": the BAM is inflated and decoded on the GPU in one pass (gappadder_amd/bam_io.py)
    instead of one `samtools view` pipe per scaffold and stage; every file of the working folder must come out identical."""
    from gappadder_amd import main as M
    case, _, ref_tree = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
    os.remove(os.path.join(str(tmp_path), "data", "draft.fa.fai"))      # `samtools faidx` is part of the builtin mode too
    for stage in ("Preprocess", "Collect", "Assembly"):
        M.main(["-c", stage, "-g", cfgp])
    assert open(os.path.join(str(tmp_path), "data", "draft.fa.fai")).read() == case.fai
    got = PU.tree(wf)
    assert sorted(got) == sorted(ref_tree)
    for rel in ref_tree:
        if "/scaffold_reads_list_all/" in rel or "/discordant_reads_list/" in rel or rel.endswith(".fastq") or rel.endswith(".fa"):
            assert got[rel] == ref_tree[rel], rel
        else:
            assert sorted(got[rel].splitlines()) == sorted(ref_tree[rel].splitlines()), rel


def _run_stages(cfgp, stages, env=None):
    from gappadder_amd import main as M
    old = {k: os.environ.get(k) for k in (env or {})}
    os.environ.update(env or {})
    try:
        for stage in stages:
            M.main(["-c", stage, "-g", cfgp])
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _same_tree(got, ref_tree):
    assert sorted(got) == sorted(ref_tree)
    for rel in ref_tree:
        if "/scaffold_reads_list_all/" in rel or "/discordant_reads_list/" in rel or rel.endswith(".fastq") or rel.endswith(".fa"):
            assert got[rel] == ref_tree[rel], rel
        else:
            assert sorted(got[rel].splitlines()) == sorted(ref_tree[rel].splitlines()), rel


def test_all_in_one_on_resident_libraries_writes_the_same_tree(run, tmp_path):
    """`-c All` with software_path.samtools = "builtin": BAM and FASTQ are read once into HBM, gappadder_amd/pipeline.py recruits,
    pools, assembles and picks on the device (the chain bench.py times), and every file of the reference's working folder is WRITTEN
    FROM those results — the first assembly round included (no trip of the pools through gap_reads/*.fastq).  The tree must equal the
    staged run through the `samtools view` pipes."""
    case, _, ref_tree = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
    import json
    tfile = os.path.join(str(tmp_path), "timings.json")
    _run_stages(cfgp, ["All"], env={"GF_TIMINGS": tfile})
    _same_tree(PU.tree(wf), ref_tree)
    t = json.load(open(tfile))
    assert {"ingest_fastq", "ingest_bam", "join", "recruit_and_sizing", "pools", "assemble_and_pick", "write_files"} <= set(t["seconds"])
    assert len(t["libraries"]) == len(case.libs) and all(l["records"] > 0 for l in t["libraries"])


def test_resident_libraries_in_many_small_chunks(run, tmp_path):
    """The same with 3-KiB file chunks: FASTQ pieces cut at record boundaries, BGZF blocks and BAM records that straddle pieces."""
    case, _, ref_tree = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
    _run_stages(cfgp, ["Preprocess", "Collect", "Assembly"], env={"GF_INGEST_CHUNK_BYTES": "3000"})
    _same_tree(PU.tree(wf), ref_tree)


def test_resident_path_equals_the_per_scaffold_path_with_the_kmer_screen(tmp_path):
    """parameters.kmer_screen on resident libraries (the screen runs inside the pipeline's step) against the host join of
    kmer_recruit.py: same gap_reads, same left/right_reads.list."""
    case = Case("twolib")
    trees = []
    for sub, env in (("a", {}), ("b", {"GF_DEVICE_COLLECT": "0"})):
        root = os.path.join(str(tmp_path), sub)
        os.makedirs(root)
        cfgp, wf, _ = PU.materialise(case, root, kmer_screen=31, builtin_bam=True)
        _run_stages(cfgp, ["Preprocess", "Collect"], env=env)
        trees.append(PU.tree(wf))
    _same_tree(trees[0], trees[1])
    assert any(k.endswith("gap_reads/0_1.fastq") for k in trees[0])


def test_inputs_the_resident_path_does_not_take_fall_back_to_the_per_scaffold_path(run, tmp_path, capfd):
    """A FASTQ pair whose mates are not in the same order (here: the right file reversed) is joined by name on the host, as the
    reference does; the device path says so and steps aside."""
    case, _, _ = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), builtin_bam=True)
    fq2 = os.path.join(str(tmp_path), "data", "lib0_2.fq")
    lines = open(fq2).read().splitlines()
    recs = ["\n".join(lines[i:i + 4]) + "\n" for i in range(0, len(lines), 4)]
    open(fq2, "w").write("".join(recs[::-1]))
    _run_stages(cfgp, ["Preprocess", "Collect"])
    assert "do not carry the same ids in the same order" in capfd.readouterr().err
    got = PU.tree(wf)
    for rel, txt in case.expected.items():          # same recruits; the right-file records of a pool now come in the reversed file's order
        if "/gap_reads/" in rel and rel.startswith("1_"):
            split = lambda t: sorted("\n".join(t.splitlines()[i:i + 4]) for i in range(0, len(t.splitlines()), 4))
            assert split(got[rel]) == split(txt), rel


def test_libraries_beyond_the_device_memory_take_the_per_scaffold_path_and_its_time_is_measured(run, tmp_path, capfd):
    """The resident path keeps every library whole in HBM: DeviceCollector.check_footprint estimates the need from the file sizes BEFORE
    anything is read and steps aside when it does not fit (here: a limit of 1 MB through GF_DEVICE_COLLECT_MAX_BYTES); an allocation that
    fails later (torch's OutOfMemoryError, GF_E_NOMEM, a capacity of the one-shot step outgrown) ends in the same fallback (main.py).
    Same tree either way — and the wall time of both paths on the same files is measured: the fallback decodes the BAM once per stage
    and joins the FASTQ by name on the host; it must stay within an order of magnitude of the resident path on inputs this small."""
    import time
    case, _, ref_tree = run
    times = {}
    for sub, env in (("resident", {}), ("fallback", {"GF_DEVICE_COLLECT_MAX_BYTES": "1000000"})):
        root = os.path.join(str(tmp_path), sub)
        os.makedirs(root)
        cfgp, wf, _ = PU.materialise(case, root, kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
        _run_stages(cfgp, ["Preprocess"])
        t0 = time.perf_counter()
        _run_stages(cfgp, ["Collect"], env=env)
        times[sub] = time.perf_counter() - t0
        err = capfd.readouterr().err
        assert ("GB of device memory" in err) == (sub == "fallback"), err[-400:]
        _run_stages(cfgp, ["Assembly"])
        _same_tree(PU.tree(wf), ref_tree)
    print("Collect on %s: resident path %.2f s, per-scaffold fallback %.2f s" % (case.name if hasattr(case, "name") else "golden case", times["resident"], times["fallback"]))
    assert times["fallback"] < 10 * times["resident"] + 5.0, times


def test_an_allocation_failure_inside_the_resident_path_ends_in_the_fallback(run, tmp_path, capfd, monkeypatch):
    """torch.cuda.OutOfMemoryError raised in the middle of the resident Collect (planted in DeviceCollector._ingest_pair): the CLI frees the
    libraries and runs the per-scaffold path; any OTHER RuntimeError still propagates."""
    import torch
    from gappadder_amd import device_collect as DC
    case, _, ref_tree = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
    _run_stages(cfgp, ["Preprocess"])

    def boom(self, *a, **k):
        raise torch.cuda.OutOfMemoryError("HIP out of memory. Tried to allocate 1.00 TiB (planted by the test)")
    monkeypatch.setattr(DC.DeviceCollector, "_ingest_pair", boom)
    _run_stages(cfgp, ["Collect"])
    assert "device-resident Collect gave up" in capfd.readouterr().err
    monkeypatch.undo()
    _run_stages(cfgp, ["Assembly"])
    _same_tree(PU.tree(wf), ref_tree)

    def other(self, *a, **k):
        raise RuntimeError("something else entirely")
    monkeypatch.setattr(DC.DeviceCollector, "_ingest_pair", other)
    with pytest.raises(RuntimeError, match="something else"):
        _run_stages(cfgp, ["Collect"])


def test_gzip_fastq_inputs_write_the_same_tree(run, tmp_path):
    """FASTQ(.gz): the read files gzip-compressed — `-c All` on resident libraries inflates them once into
    {wf}tmp_fastq/ and writes the tree of the plain-text run (the per-gap FASTQ records are cut out of the plain copies)."""
    import gzip
    import json
    case, _, ref_tree = run
    cfgp, wf, _ = PU.materialise(case, str(tmp_path), kmers=((31, 29), (41, 39), (41, 38)), builtin_bam=True)
    cfg = json.load(open(cfgp))
    for rr in cfg["raw_reads"]:
        for side in ("left", "right"):
            raw = open(rr[side], "rb").read()
            open(rr[side] + ".gz", "wb").write(gzip.compress(raw[:len(raw) // 2]) + gzip.compress(raw[len(raw) // 2:]))
            os.remove(rr[side])
            rr[side] += ".gz"
    json.dump(cfg, open(cfgp, "w"))
    _run_stages(cfgp, ["All"])
    got = {k: v for k, v in PU.tree(wf).items() if not k.startswith("tmp_fastq/")}
    _same_tree(got, ref_tree)
    assert len([f for f in os.listdir(wf + "tmp_fastq")]) == 2 * len(case.libs)


def test_resident_path_with_n_bases_ragged_reads_and_a_late_long_read(tmp_path):
    """Reads with N, reads of different lengths (packed at the longest, the tail masked) and a read longer than the first ones promised
    (the ingest restarts at its length): the N masks travel with the pooled reads through the library merge into the first assembly
    round on the device.  `-c All` on resident libraries must write the tree of the per-scaffold path, contigs included."""
    import random
    case = Case("twolib")
    rng = random.Random(11)

    def mutate(fq, lib_no):
        lines = fq.splitlines()
        for r in range(len(lines) // 4):
            s, q = lines[4 * r + 1], lines[4 * r + 3]
            if r % 9 == 0:                                   # an N somewhere
                p = rng.randrange(len(s))
                s = s[:p] + "N" + s[p + 1:]
            if lib_no == 1 and r % 3 == 0:                   # ragged: 110-149 bases
                n = rng.randrange(110, len(s))
                s, q = s[:n], q[:n]
            if lib_no == 0 and r == len(lines) // 4 - 7:     # one longer read near the end of the file
                s, q = s + "ACGTTGCA", q + "IIIIIIII"
            lines[4 * r + 1], lines[4 * r + 3] = s, q
        return "\n".join(lines) + "\n"

    trees = []
    for sub, env in (("resident", {"GF_INGEST_GUESS_RECORDS": "50"}), ("per_scaffold", {"GF_DEVICE_COLLECT": "0"})):
        rng.seed(11)                                         # the same mutations for both runs
        root = os.path.join(str(tmp_path), sub)
        os.makedirs(root)
        cfgp, wf, _ = PU.materialise(case, root, kmers=((31, 29), (41, 39)), builtin_bam=True)
        for i in range(len(case.libs)):
            for m in (1, 2):
                p = os.path.join(root, "data", "lib%d_%d.fq" % (i, m))
                text = open(p).read()
                open(p, "w").write(mutate(text, i))
        tfile = os.path.join(root, "timings.json")
        _run_stages(cfgp, ["All"], env=dict(env, GF_TIMINGS=tfile))
        trees.append(PU.tree(wf))
        if sub == "resident":     # the first 50 records promised 150 bases; the ingest restarted at the long read's 158
            import json
            assert json.load(open(tfile))["read_len"] == 158
    _same_tree(trees[0], trees[1])
    assert any("N" in v for k, v in trees[0].items() if k.startswith("merged/gap_reads/"))
    assert sum(v.count(">") for k, v in trees[0].items() if k.endswith("/contigs.fa")) > 10


def test_resident_path_with_orphan_and_secondary_alignment_records(tmp_path):
    """Alignment records whose QNAME is in no FASTQ record (an orphan: the reference lists it and then finds nothing to pull), secondary
    copies of records (FLAG 0x100: the reference reads them like any other line) and hard clips: same tree from both paths."""
    case = Case("edge")

    class Mutated:
        pass
    c = Mutated()
    c.draft_fa, c.fai, c.meta = case.draft_fa, case.fai, case.meta
    c.libs = []
    for lib in case.libs:
        out = []
        for i, line in enumerate(lib["sam"].splitlines()):
            out.append(line)
            f = line.split("\t")
            if i % 40 == 0 and len(f) > 10:
                out.append("\t".join([f[0] + "_orphan"] + f[1:]))                                   # a name no FASTQ record has
            if i % 55 == 0 and len(f) > 10:
                out.append("\t".join([f[0], str(int(f[1]) | 0x100)] + f[2:5] + [f[5].replace("S", "H")] + f[6:]))   # secondary copy, hard clips
        c.libs.append(dict(lib, sam="\n".join(out) + "\n"))
    trees = []
    for sub, env in (("resident", {}), ("per_scaffold", {"GF_DEVICE_COLLECT": "0"})):
        root = os.path.join(str(tmp_path), sub)
        os.makedirs(root)
        cfgp, wf, _ = PU.materialise(c, root, kmers=((31, 29),), builtin_bam=True)
        _run_stages(cfgp, ["Preprocess", "Collect"], env=env)
        trees.append(PU.tree(wf))
    _same_tree(trees[0], trees[1])
    lists = "".join(v for k, v in trees[0].items() if "/scaffold_reads_list_all/" in k)
    assert "_orphan " in lists
    assert not any("_orphan" in v for k, v in trees[0].items() if k.endswith(".fastq"))
