#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  Nothing of the
reference is copied into the repo: its Python-2 sources are converted with lib2to3 into a
temp dir, three tiny shims are placed beside them (`user`, `Bio.SeqIO`, a `samtools`
stand-in that serves pre-made SAM text), `main.py -c Preprocess` and `-c Collect` are run on
seeded synthetic input, and only DATA is captured: the synthetic inputs and the files the
reference wrote (gap positions, flank FASTA, read lists, per-gap FASTQ).

Also builds the reference's own KmerUtils.cpp (via oracle/Makefile -> oracle/_ref/) and
dumps its known answers into tests/golden/kmerutils_kat.json.

usage: python tests/golden/make_golden.py            (re-creates every fixture)
"""
import gzip
import io
import json
import os
import shutil
import subprocess
import sys
import tarfile
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
from synth_text import make_case  # noqa: E402  (text-level synthetic generator, this repo's own)

REF_PY = ["put_gap_seq_back_to_scaffold.py", "main.py", "Utility.py", "gnrt_pos_true_seqs.py", "run_multi_threads_collect_reads.py",
          "collect_reads_for_gaps.py", "run_multi_threads_discordant.py",
          "collect_discordant_low_mapq_reads.py", "merge_reads.py", "assemble_gaps.py",
          "MergeContigs.py", "pick_contigs.py", "collect_both_unmapped_reads.py"]

SEQIO_SHIM = '''
class _Seq(str):
    pass
class _Rec(object):
    def __init__(self, id, seq, description=None):
        self.id = id
        self.seq = _Seq(seq)
        self.description = id if description is None else description
def write(record, handle, fmt):
    # Biopython's FastaWriter convention (Bio/SeqIO/FastaIO.py; Biopython itself is absent here): title = description when it
    # starts with the id, "id description" when there is another description, else the id; sequence wrapped at 60 columns
    assert fmt == "fasta"
    d = record.description
    if d and d.split(None, 1)[0] == record.id:
        title = d
    elif d:
        title = "%s %s" % (record.id, d)
    else:
        title = record.id
    s = str(record.seq)
    handle.write(">" + title + "\\n" + "".join(s[i:i + 60] + "\\n" for i in range(0, len(s), 60)))
def parse(path, fmt):
    if fmt == "fasta":
        name, chunks = None, []
        with open(path) as f:
            for line in f:
                line = line.rstrip("\\n")
                if line.startswith(">"):
                    if name is not None:
                        yield _Rec(name, "".join(chunks), desc)
                    name, chunks, desc = line[1:].split()[0], [], line[1:]
                else:
                    chunks.append(line)
        if name is not None:
            yield _Rec(name, "".join(chunks), desc)
    elif fmt == "fastq":
        with open(path) as f:
            while True:
                h = f.readline()
                if not h:
                    break
                s = f.readline().rstrip("\\n"); f.readline(); f.readline()
                yield _Rec(h[1:].split()[0], s)
    else:
        raise ValueError(fmt)
'''

SAMTOOLS_SHIM = '''#!/usr/bin/env python3
# stand-in for `samtools view <bam> "<scaffold>"` / `samtools faidx <fa>`: serves <bam>.sam text
import sys
if sys.argv[1] == "view" and sys.argv[2] == "-f":      # view -f <mask> <bam>
    mask, bam = int(sys.argv[3]), sys.argv[4]
    with open(bam + ".sam") as f:
        for line in f:
            if int(line.split("\\t")[1]) & mask == mask:
                sys.stdout.write(line)
elif sys.argv[1] == "view" and sys.argv[2].startswith("-"):
    pass  # -H / -h -S -b on files this stand-in does not have
elif sys.argv[1] == "view":
    bam, scf = sys.argv[2], sys.argv[3]
    with open(bam + ".sam") as f:
        for line in f:
            if line.split("\\t")[2] == scf:
                sys.stdout.write(line)
else:
    pass  # faidx (the generator writes the .fai itself), index, sort
'''


ROUND2_DRIVER = '''
import sys
from Utility import set_software_paths, set_parameters
import collect_both_unmapped_reads as M
samtools, draft, wf, ids, bams = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4].split(","), sys.argv[5].split(",")
set_software_paths("/nonexistent/bwa", samtools, "x", "x", "/nonexistent/kmc/", "/nonexistent/velvet/")
set_parameters(draft, 2, wf, 0, 100, 300)
M.BothUnmappedReadsCollector(wf).collect_both_unmapped_reads(bams, ids)
'''


def round2_contigs(case, coords, seed):
    """Made-up first-round contigs for two gaps (inputs of the both-unmapped round): pieces of the true gap interior, so that
    both-unmapped pairs (which come from gap interiors) share k-mers with them; 60-column FASTA, Velvet-style names.
    coords: {gapKey: (scaffold name, start, end)}."""
    import random
    rng = random.Random(seed)
    out = {}
    for key in sorted(coords):
        sname, st, en = coords[key]
        inner = case["true_seqs"][sname][st:en]
        recs = []
        for n in range(2):
            a = rng.randrange(0, max(1, len(inner) - 200))
            piece = inner[a:a + rng.randrange(120, 200)]
            recs.append(">31_29_NODE_%d_length_%d_cov_12.500000\n" % (n + 1, len(piece) - 28) +
                        "".join(piece[i:i + 60] + "\n" for i in range(0, len(piece), 60)))
        out[key] = "".join(recs)
    return out


def convert_reference(dst):
    for f in REF_PY:
        shutil.copy(os.path.join(REF, f), os.path.join(dst, f))
        os.chmod(os.path.join(dst, f), 0o644)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", "-j", "4", dst],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    open(os.path.join(dst, "user.py"), "w").close()
    os.mkdir(os.path.join(dst, "Bio"))
    open(os.path.join(dst, "Bio", "__init__.py"), "w").close()
    with open(os.path.join(dst, "Bio", "SeqIO.py"), "w") as f:
        f.write(SEQIO_SHIM)
    with open(os.path.join(dst, "Bio", "Seq.py"), "w") as f:
        f.write("class Seq(str):\n    pass\n")
    with open(os.path.join(dst, "Bio", "SeqRecord.py"), "w") as f:
        f.write("class SeqRecord(object):\n    def __init__(self, seq, id='', description=''):\n"
                "        self.seq, self.id, self.description = seq, id, description\n")
    st = os.path.join(dst, "samtools_shim.py")
    with open(st, "w") as f:
        f.write(SAMTOOLS_SHIM)
    os.chmod(st, 0o755)
    # `python` must resolve to this interpreter inside the reference's shell pipelines
    bindir = os.path.join(dst, "bin")
    os.mkdir(bindir)
    os.symlink(sys.executable, os.path.join(bindir, "python"))
    return st, bindir


def run_reference(case, out_tar, in_dir):
    """case: dict from synth_text.make_case.  Writes inputs to in_dir, expected tree to out_tar."""
    tmp = tempfile.mkdtemp(prefix="gp_golden_")
    try:
        code = os.path.join(tmp, "code")
        os.mkdir(code)
        samtools, bindir = convert_reference(code)
        data = os.path.join(tmp, "data")
        wf = os.path.join(tmp, "wf")
        os.mkdir(data)
        os.mkdir(wf)
        with open(os.path.join(data, "draft.fa"), "w") as f:
            f.write(case["draft_fa"])
        with open(os.path.join(data, "draft.fa.fai"), "w") as f:
            f.write(case["fai"])
        libs = []
        for i, lib in enumerate(case["libs"]):
            bam = os.path.join(data, "lib%d.bam" % i)
            open(bam, "w").close()
            with open(bam + ".sam", "w") as f:
                f.write(lib["sam"])
            lfq, rfq = os.path.join(data, "lib%d_1.fq" % i), os.path.join(data, "lib%d_2.fq" % i)
            with open(lfq, "w") as f:
                f.write(lib["fq1"])
            with open(rfq, "w") as f:
                f.write(lib["fq2"])
            libs.append((bam, lfq, rfq, lib["is"], lib["sd"]))
        cfg = {
            "draft_genome": {"fa": os.path.join(data, "draft.fa")},
            "raw_reads": [{"left": l, "right": r} for (_, l, r, _, _) in libs],
            "alignments": [{"bam": b, "is": str(i), "std": str(s)} for (b, _, _, i, s) in libs],
            "software_path": {"bwa": "bwa", "samtools": samtools, "velvet": "/nonexistent/velvet/",
                              "kmc": "/nonexistent/kmc/", "TERefiner": "./TERefiner_1",
                              "ContigsMerger": "./ContigsMerger"},
            "parameters": {"working_folder": wf, "min_gap_size": str(case["min_gap"]),
                           "flank_length": str(case["flank"]), "nthreads": "2", "verbose": "0"},
            "kmer_length": [{"k": 31, "k_velvet": [{"k": 29}]}],
        }
        cfgp = os.path.join(tmp, "cfg.json")
        with open(cfgp, "w") as f:
            json.dump(cfg, f)
        env = dict(os.environ)
        env["PATH"] = bindir + ":" + env["PATH"]
        env["LC_ALL"] = "C"
        for stage in ("Preprocess", "Collect"):
            subprocess.check_call([sys.executable, "main.py", "-c", stage, "-g", cfgp], cwd=code, env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        # ---- both-unmapped round (collect_both_unmapped_reads.py) on made-up first-round contigs, twolib only ----
        round2 = None
        if case["name"] == "twolib":
            coords = {"0_3": ("scf0", 15000, 17000), "2_2": ("scf2", 9000, 10000)}
            contigs = round2_contigs(case, coords, 77)
            mwf = os.path.join(wf, "merged") + "/"
            for key, fa in contigs.items():
                os.makedirs(os.path.join(mwf, "velvet_temp", key), exist_ok=True)
                with open(os.path.join(mwf, "velvet_temp", key, "contigs.fa"), "w") as f:
                    f.write(fa)
            for sub in ("unmapped_reads",):
                os.makedirs(os.path.join(mwf, sub), exist_ok=True)
            before = {fn: open(os.path.join(mwf, "gap_reads", fn)).read() for fn in os.listdir(os.path.join(mwf, "gap_reads"))}
            with open(os.path.join(code, "run_second_round.py"), "w") as f:
                f.write(ROUND2_DRIVER)
            subprocess.check_call([sys.executable, "run_second_round.py", samtools, os.path.join(data, "draft.fa"), mwf,
                                   ",".join(sorted(coords) + ["0_1"]), ",".join(b for (b, _, _, _, _) in libs)],
                                  cwd=code, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            after = {fn: open(os.path.join(mwf, "gap_reads", fn)).read() for fn in os.listdir(os.path.join(mwf, "gap_reads"))}
            assert after == before    # no bwa here: the reference's alignment step recruits nothing
            round2 = {"contigs": contigs, "ids": sorted(coords) + ["0_1"], "files": {}}
            for i, (bam, _, _, _, _) in enumerate(libs):
                round2["files"]["lib%d.both_unmapped.fq" % i] = open(bam + ".both_unmapped.fq").read()
            for fn in ("both_unmapped.fq", "both_unmapped_1.fq", "both_unmapped_2.fq", "gap_contigs_all.fa"):
                round2["files"][fn] = open(os.path.join(mwf, fn)).read()
                os.remove(os.path.join(mwf, fn))
            shutil.rmtree(os.path.join(mwf, "velvet_temp"))
            shutil.rmtree(os.path.join(mwf, "unmapped_reads"))
        # ---- write-back of closed sequences into the scaffolds (put_gap_seq_back_to_scaffold.py), twolib only ----
        writeback = None
        if case["name"] == "twolib":
            import random
            rng = random.Random(5)
            picked = ""
            for gid in ("0_1", "0_3", "2_2"):     # 0_2 and 2_1 stay open; scf1 has no gap
                seq = "".join(rng.choice("ACGT") for _ in range(rng.randrange(70, 200)))
                picked += ">%s_31_29_NODE_1_length_99_cov_7.000000\n%s\n" % (gid, seq)
            pf = os.path.join(tmp, "picked_seqs.fa")
            open(pf, "w").write(picked)
            newf = os.path.join(tmp, "new_scaffolds.fa")
            subprocess.check_call([sys.executable, "put_gap_seq_back_to_scaffold.py", os.path.join(data, "draft.fa"),
                                   os.path.join(wf, "gap_positions.txt"), pf, newf], cwd=code, env=env,
                                  stdout=subprocess.DEVNULL)
            writeback = {"picked_fa": picked, "new_scaffolds_fa": open(newf).read()}
        # ---- capture inputs (data only) ----
        os.makedirs(in_dir, exist_ok=True)
        def gz(name, text):
            with gzip.GzipFile(os.path.join(in_dir, name + ".gz"), "wb", mtime=0) as g:
                g.write(text.encode())
        with open(os.path.join(in_dir, "draft.fa.fai"), "w") as f:
            f.write(case["fai"])
        # C1's inputs (20 000 SAM lines, 2 x 10 000 FASTQ records) are not stored: tests/golden_util.py regenerates them from
        # the seed with synth_text.make_case (this repo's own integer-only generator); their digest is recorded instead
        regenerate = case["name"] == "c1"
        if not regenerate:
            gz("draft.fa", case["draft_fa"])
            for i, lib in enumerate(case["libs"]):
                gz("lib%d.sam" % i, lib["sam"])
                gz("lib%d_1.fq" % i, lib["fq1"])
                gz("lib%d_2.fq" % i, lib["fq2"])
        meta = {"min_gap": case["min_gap"], "flank": case["flank"], "anchor_mapq": 30, "clip_dist": 250,
                "libs": [{"is": l["is"], "sd": l["sd"]} for l in case["libs"]], "seed": case["seed"]}
        if regenerate:
            import hashlib
            meta["regenerate"] = True
            meta["inputs_sha256"] = hashlib.sha256("".join([case["draft_fa"]] + [l[x] for l in case["libs"] for x in ("sam", "fq1", "fq2")]).encode()).hexdigest()
        with open(os.path.join(in_dir, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
        # ---- capture what the reference wrote ----
        keep = []
        for root, _, files in os.walk(wf):
            for fn in files:
                p = os.path.join(root, fn)
                rel = os.path.relpath(p, wf)
                top = rel.split(os.sep)[0]
                if top in ("empty_dir",):
                    continue
                keep.append(rel)
        keep.sort()
        buf = io.BytesIO()
        with tarfile.open(fileobj=buf, mode="w") as tf:
            for rel in keep:
                ti = tf.gettarinfo(os.path.join(wf, rel), arcname=rel)
                ti.mtime = 0; ti.uid = ti.gid = 0; ti.uname = ti.gname = ""
                with open(os.path.join(wf, rel), "rb") as fh:
                    tf.addfile(ti, fh)
        with gzip.GzipFile(out_tar, "wb", mtime=0) as g:
            g.write(buf.getvalue())
        if writeback is not None:
            with gzip.GzipFile(os.path.join(os.path.dirname(out_tar), "writeback.json.gz"), "wb", mtime=0) as g:
                g.write(json.dumps(writeback, indent=0, sort_keys=True).encode())
        if round2 is not None:
            with gzip.GzipFile(os.path.join(os.path.dirname(out_tar), "round2.json.gz"), "wb", mtime=0) as g:
                g.write(json.dumps(round2, indent=0, sort_keys=True).encode())
        return keep
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def kmerutils_kat():
    """Known answers from the reference's own KmerUtils.cpp, built into oracle/_ref/ (never committed)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    exe = os.path.join(REPO, "oracle", "_ref", "kmerutils_kat")
    cases = [("ACGT", 4), ("CGTN", 4), ("ACGTACGTTTGACCA", 5), ("ACGT" * 8 + "TG", 32),
             ("acgtnNacgtTTGGCCAAxyzACGT", 7), ("TTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTTT", 31),
             ("GATTACAGATTACAGATTACAGATTACAGATTACAGATTACA", 31), ("A" * 40, 32), ("C", 1)]
    out = {"pack": [], "predicate": []}
    for seq, k in cases:
        r = subprocess.check_output([exe, "kmers", seq, str(k)]).decode().split()
        out["pack"].append({"seq": seq, "k": k, "kmers_hex": r})
    for kh in ("1b00000000000000", "fe00000000000000"):
        r = subprocess.check_output([exe, "tostr", kh, "5"]).decode().strip()
        out.setdefault("tostr", []).append({"kmer_hex": kh, "k": 5, "str": r})
    # IsReadContainingFreqKmers truth table: map = kmers of `src`; read; threshold
    src = "ACGTACGTTTGACCAGGATTACATTTGACCA"
    for read, k, thr in [("TTTGACCAGG", 5, 1), ("TTTGACCAGG", 5, 6), ("TTTGACCAGG", 5, 7), ("GGGGGGGGGG", 5, 1),
                         ("GGGGGGGGGG", 5, 0), ("ACGTACGTTTGACCAGGATTACATTTGACCA", 9, 23),
                         ("ACGTACGTTTGACCAGGATTACATTTGACCA", 9, 24), ("TGGTCAAATG", 5, 1)]:
        r = subprocess.check_output([exe, "pred", src, read, str(k), str(thr)]).decode().strip()
        out["predicate"].append({"src": src, "read": read, "k": k, "thr": thr, "result": int(r)})
    with open(os.path.join(HERE, "kmerutils_kat.json"), "w") as f:
        json.dump(out, f, indent=1)


def quickcheck_kat():
    """Known answers from the reference's own all-pairs 10-mer prefilter of the contig merger (QuickCheckerContigsMatch,
    ContigsCompactor.cpp:1982-2095), built into oracle/_ref/quickcheck_kat: contig sets -> feasible node pairs."""
    import random
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    exe = os.path.join(REPO, "oracle", "_ref", "quickcheck_kat")
    rng = random.Random(20260301)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    sets = []
    a = rnd(200)
    sets.append([a, a[150:] + rnd(150), rnd(120)])                                   # a suffix-prefix overlap, an unrelated contig
    g = rnd(1500)
    sets.append([g[i:i + rng.randrange(80, 400)] for i in range(0, 1300, 90)])       # a tiling: many overlaps
    sets.append([rc(x) if i % 2 else x for i, x in enumerate(sets[-1])])             # the same, every other contig reverse-complemented
    sets.append([rnd(30), rnd(31), rnd(40), "ACGT" * 10, "A" * 45, "AC" * 30])       # shortest legal contigs, low complexity
    n = rnd(300)
    sets.append([n[:100] + "NNNNN" + n[105:200], n[150:300], "N" * 12 + n[20:80]])   # N counts as A (KmerUtils.cpp:25-41)
    sets.append([rnd(rng.randrange(40, 300)) for _ in range(60)])                     # 60 unrelated contigs: hits by chance only
    rep = rnd(60)
    sets.append([rnd(100) + rep + rnd(100), rep + rnd(80), rnd(90) + rep, rc(rep) + rnd(70)])   # a shared repeat at contig ends
    out = []
    tmp = tempfile.mkdtemp(prefix="gp_qc_")
    try:
        for ci, contigs in enumerate(sets):
            fa = os.path.join(tmp, "c%d.fa" % ci)
            with open(fa, "w") as f:
                f.write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(contigs)))
            for k in ((10,) if ci else (10, 8, 12)):
                r = subprocess.check_output([exe, fa, str(k)]).decode().split()
                pairs = [[int(r[i]), int(r[i + 1])] for i in range(0, len(r), 2)]
                out.append({"contigs": contigs, "k": k, "pairs": pairs})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with gzip.GzipFile(os.path.join(HERE, "quickcheck_kat.json.gz"), "wb", mtime=0) as gzf:
        gzf.write(json.dumps(out, sort_keys=True).encode())
    print("quickcheck_kat:", len(out), "cases,", sum(len(o["pairs"]) for o in out), "feasible pairs")


def evaluate_kat():
    """Known answers from the reference's own pairwise overlap evaluation of the contig merger (ContigsCompactor::Evaluate,
    ContigsCompactor.cpp:1572-1976), built into oracle/_ref/evaluate_kat: contig sets + parameters -> the result of every ordered node
    pair (class; for classes 1/2: end row, clip, overlap size, merged length, containment)."""
    import random
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    exe = os.path.join(REPO, "oracle", "_ref", "evaluate_kat")
    rng = random.Random(20260302)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))
    def noisy(s, subs, indels):
        s = list(s)
        for _ in range(subs):
            i = rng.randrange(len(s)); s[i] = rng.choice([c for c in "ACGT" if c != s[i]])
        for _ in range(indels):
            i = rng.randrange(1, len(s) - 1)
            if rng.random() < 0.5: del s[i]
            else: s.insert(i, rng.choice("ACGT"))
        return "".join(s)
    gappadder = (-2.0, -2.0, 50, 0.005, 0.4, 12, 6)          # MergeContigs.py:75: -i1 -2.0 -i2 -2.0 -y 50 -s 0.4 -x 12; main.cpp:24-27 defaults
    sets = []
    g = rnd(900)
    sets.append(([g[0:300], g[220:560], g[500:900]], gappadder))                                   # clean suffix-prefix overlaps
    sets.append(([g[0:300], rc(g[220:560]), g[500:900], rnd(200)], gappadder))                     # one contig reverse-complemented, one unrelated
    sets.append(([g[0:400], g[100:250], noisy(g[300:700], 6, 2), g[650:900] + rnd(40)], gappadder))  # containment; mismatches + indels; a dirty end (clip)
    sets.append(([noisy(g[0:350], 10, 3), noisy(g[250:600], 10, 3), noisy(g[560:900], 4, 0)], gappadder))
    sets.append(([g[0:300], g[220:560], g[500:900]], (-1.0, -1.0, 0, 0.005, 0.01, 100000, 6)))     # the tool's own defaults: class 1 only
    sets.append(([g[0:120], g[100:230], g[225:300], rnd(30), g[0:120]], (-2.0, -2.0, 10, 0.3, 0.2, 25, 15)))   # short overlaps around the thresholds, a duplicate
    sets.append(([rnd(rng.randrange(40, 160)) for _ in range(12)], gappadder))                     # unrelated contigs: chance overlaps at the ends
    sets.append((["ACGT" * 20, "CGTA" * 15 + rnd(30), "A" * 50, "A" * 30 + rnd(20)], gappadder))   # repeats: many equal scores (tie-breaking)
    out = []
    tmp = tempfile.mkdtemp(prefix="gp_ev_")
    try:
        for ci, (contigs, pr) in enumerate(sets):
            fa = os.path.join(tmp, "c%d.fa" % ci)
            with open(fa, "w") as f:
                f.write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(contigs)))
            lines = subprocess.check_output([exe, fa] + [repr(float(x)) for x in pr]).decode().strip().split("\n")
            out.append({"contigs": contigs, "params": list(pr), "results": [[int(x) for x in l.split()] for l in lines]})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with gzip.GzipFile(os.path.join(HERE, "evaluate_kat.json.gz"), "wb", mtime=0) as gzf:
        gzf.write(json.dumps(out, sort_keys=True).encode())
    print("evaluate_kat:", len(out), "cases,", sum(len(o["results"]) for o in out), "node pairs,",
          sum(1 for o in out for r in o["results"] if r[2]), "with an overlap")

BWA_SHIM = '''#!/usr/bin/env python3
# stand-in for `bwa index <fa>` / `bwa mem -T <score> -a <contigs.fa> <flanks.fa>`: serves the prepared SAM text <contigs.fa>.sam
import os, sys
if sys.argv[1] == "mem":
    p = sys.argv[-2] + ".sam"
    if os.path.exists(p):
        sys.stdout.write(open(p).read())
'''

SAMVIEW_SHIM = '''#!/usr/bin/env python3
# stand-in for `samtools view -S -`: SAM text through, header lines dropped
import sys
for line in sys.stdin:
    if not line.startswith("@"):
        sys.stdout.write(line)
'''

PICK_DRIVER = '''
import os, sys, json
from Utility import set_software_paths, set_parameters
bwa, samtools, wf, score, mode = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5]
ids = sys.argv[6].split(",")
set_software_paths(bwa, samtools, "x", "x", "/nonexistent/kmc/", "/nonexistent/velvet/")
set_parameters("x", 1, wf, 0, 100, 300)
import pick_contigs as M
cs = M.ContigsSelection(wf)
if mode == "full":
    cs.pick_full_constructed_contigs(score, ids, wf + "../picked_seqs.fa")
else:
    for i in ids:                      # one gap at a time: an int-vs-str tie test (a TypeError under Python 3) must not take the others down
        try:
            M.run_pick_extended_contig(i)
        except TypeError:
            open(wf + "velvet_temp/%s/TIE" % i, "w").close()
'''


def pick_kat():
    """Known answers of the reference's own contig picker (pick_contigs.py:64-358 full pick, 361-539 extended pick) for prepared
    flanks.sam texts: hand-built cases (every clip-type pair, both strands, several hits of one type, equal spans, hits that must be
    ignored) and random hit sets.  bwa is absent, so a stand-in serves the prepared SAM text; everything downstream of flanks.sam is
    the reference's code.  Output: tests/golden/pick_kat.json.gz = [{id, contigs, sam, score, full: [seqs, contigs, ledger], ext: [seqs, contigs] | "TIE"}]."""
    import random
    rng = random.Random(20260310)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))

    def sam(gid, side, flag, rname, pos, cigar):
        return "%s_%s\t%d\t%s\t%d\t60\t%s\t*\t0\t0\t*\t*\n" % (gid, side, flag, rname, pos, cigar)
    cases = []

    def add(contigs, lines, score=30):
        gid = "%d_%d" % (len(cases) // 7, len(cases) % 7 + 1)
        cases.append({"id": gid, "contigs": contigs, "sam": "".join(sam(gid, *l) for l in lines), "score": score})
    c = [("NODE_1_length_372_cov_9.000000", rnd(400)), ("NODE_2_length_172_cov_4.500000", rnd(200)), ("NODE_3_length_72_cov_2.000000", rnd(100))]
    n1, n2, n3 = (x[0] for x in c)
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 301, "30M265S")])                       # the plain forward case
    add(c, [("left", 16, n1, 301, "30M265S"), ("right", 16, n1, 21, "265S30M")])                     # both on the reverse strand
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 16, n1, 301, "30M265S")])                      # strands differ: nothing
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 301, "265S30M")])                       # (left clip, left clip): no such pair
    add(c, [("left", 0, n1, 1, "295M"), ("right", 0, n1, 330, "60M235S")])                           # unclipped + right clip
    add(c, [("left", 0, n1, 1, "295M"), ("right", 0, n1, 330, "71M")])                               # unclipped + unclipped
    add(c, [("left", 0, n1, 300, "30M265S"), ("right", 0, n1, 5, "265S30M")])                        # (right clip, left clip) forward: negative span
    add(c, [("left", 0, n1, 21, "100S30M165S"), ("right", 0, n1, 301, "30M265S")])                   # both-clipped hit ignored
    add(c, [("left", 0, n1, 21, "265S30M"), ("left", 0, n1, 51, "255S40M"), ("right", 0, n1, 301, "30M265S")])      # longer match of one type replaces
    add(c, [("left", 0, n1, 21, "265S30M"), ("left", 0, n1, 51, "265S30M"), ("right", 0, n1, 301, "30M265S")])      # equal match: the first stays
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 301, "30M265S"), ("left", 0, n2, 11, "265S30M"), ("right", 0, n2, 151, "30M265S")])   # two contigs: longest span
    add(c, [("left", 0, n2, 11, "265S30M"), ("right", 0, n2, 151, "30M265S"), ("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 301, "30M265S")])
    add(c, [("left", 0, n2, 11, "265S30M"), ("right", 0, n2, 151, "30M265S"), ("left", 0, n3, 1, "265S30M"), ("right", 0, n3, 141, "30M265S")])    # hmm: equal spans (110): the first
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n2, 151, "30M265S")])                       # flanks on different contigs: extended only
    add(c, [("left", 16, n1, 301, "30M265S"), ("right", 0, n2, 151, "30M265S")])
    add(c, [("left", 0, n1, 21, "265S30M")])                                                         # one side only
    add(c, [("right", 16, n2, 21, "265S30M")])
    add(c, [("left", 0, n1, 21, "*"), ("right", 0, n1, 301, "30M265S")])                             # unaligned line
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 51, "30M265S")])                        # span 0: one base written (the slice's +1)
    add(c, [("left", 0, n1, 21, "265S30M"), ("right", 0, n1, 50, "30M265S")])                        # span -1: picked_seqs.fa written EMPTY
    add(c, [("left", 16, n1, 301, "30M265S"), ("right", 16, n1, 21, "265S30M"), ("left", 0, n1, 40, "265S25M"), ("right", 0, n1, 200, "45M250S")])  # a longer forward pair after a reverse one: b_rc sticks
    add(c, [("left", 256, n1, 21, "265S30M"), ("right", 256, n1, 301, "30M265S")])                   # secondary lines (-a): extended treats them as reverse
    add(c, [("left", 0, n1, 21, "265S20M3I7M"), ("right", 0, n1, 301, "12M2D18M265S")], 15)          # indels: only M counts
    add([("a", rnd(80).lower()), ("b", "ACGTNNRYacgt" * 10)], [("left", 16, "b", 60, "20M5S"), ("right", 16, "b", 11, "5S20M"),
                                                                 ("left", 0, "a", 5, "5S20M"), ("right", 0, "a", 50, "20M5S")], 15)   # lower case, IUPAC: revcomp's table
    types_l = ["%dS%dM", "%dM%dS", "%dM"]
    for _ in range(240):                                 # random hit sets over 1-4 contigs
        ctg = [("c%d" % i, rnd(rng.randrange(60, 400))) for i in range(rng.randrange(1, 5))]
        lines = []
        for _h in range(rng.randrange(1, 9)):
            name, s = rng.choice(ctg)
            m = rng.randrange(15, 60)
            t = rng.choice(types_l)
            cigar = t % ((295 - m, m) if t.startswith("%dS") else (m, 295 - m) if t.endswith("S") else (m,))
            lines.append((rng.choice(["left", "right"]), rng.choice([0, 0, 16, 16, 256]), name, rng.randrange(1, max(2, len(s) - m)), cigar))
        add(ctg, lines, rng.choice([30, 15]))
    for _ in range(240):                                 # random PAIRS (one left + one right hit per contig, mostly compatible) + noise lines
        ctg = [("c%d" % i, rnd(rng.randrange(120, 500))) for i in range(rng.randrange(1, 5))]
        lines = []
        for name, s in ctg:
            if rng.random() < 0.25:
                continue
            rcs = rng.random() < 0.4
            ml, mr = rng.randrange(15, 50), rng.randrange(15, 50)
            a, b = sorted((rng.randrange(1, len(s) - 60), rng.randrange(1, len(s) - 60)))
            if rng.random() < 0.15:
                a, b = b, a
            kind = rng.random()
            lc = "%dM" % ml if kind < 0.15 else ("%dM%dS" % (ml, 295 - ml) if rcs else "%dS%dM" % (295 - ml, ml))
            rcg = "%dM" % mr if 0.1 < kind < 0.25 else ("%dS%dM" % (295 - mr, mr) if rcs else "%dM%dS" % (mr, 295 - mr))
            fl = 16 if rcs else 0
            pair = [("left", fl, name, b if rcs else a, lc), ("right", fl if rng.random() < 0.9 else 16 - fl, name, a if rcs else b, rcg)]
            rng.shuffle(pair)
            lines += pair
        for _h in range(rng.randrange(0, 3)):
            name, s = rng.choice(ctg)
            m = rng.randrange(15, 60)
            lines.insert(rng.randrange(0, len(lines) + 1), (rng.choice(["left", "right"]), rng.choice([0, 16]), name, rng.randrange(1, len(s) - m),
                                                             rng.choice(["%dS%dM" % (295 - m, m), "%dM%dS" % (m, 295 - m), "*", "10S%dM10S" % m])))
        add(ctg, lines, rng.choice([30, 15]))
    tmp = tempfile.mkdtemp(prefix="gp_pick_")
    try:
        code = os.path.join(tmp, "code")
        os.mkdir(code)
        _, bindir = convert_reference(code)
        for name, text in (("bwa_shim.py", BWA_SHIM), ("samview_shim.py", SAMVIEW_SHIM), ("run_pick.py", PICK_DRIVER)):
            with open(os.path.join(code, name), "w") as f:
                f.write(text)
            os.chmod(os.path.join(code, name), 0o755)
        env = dict(os.environ)
        env["PATH"] = bindir + ":" + env["PATH"]
        wf = os.path.join(tmp, "wf", "merged") + "/"
        os.makedirs(os.path.join(tmp, "wf", "flank_regions"))
        for cs in cases:
            d = wf + "velvet_temp/%s/" % cs["id"]
            os.makedirs(d)
            with open(d + "contigs.fa", "w") as f:
                f.write("".join(">%s\n%s\n" % (n, "\n".join(s[i:i + 60] for i in range(0, len(s), 60))) for n, s in cs["contigs"]))
            with open(d + "contigs.fa.sam", "w") as f:
                f.write("@SQ\tSN:x\tLN:1\n" + cs["sam"])
            with open(os.path.join(tmp, "wf", "flank_regions", cs["id"] + ".fa"), "w") as f:
                f.write(">%s_left\nACGT\n>%s_right\nACGT\n" % (cs["id"], cs["id"]))
        rd = lambda p: open(p).read() if os.path.exists(p) else None
        for score in (30, 15):
            ids = [cs["id"] for cs in cases if cs["score"] == score]
            ledger = os.path.join(tmp, "wf", "picked_seqs.fa")
            for p in (ledger, ledger + "_ori.txt"):
                if os.path.exists(p):
                    os.remove(p)
            subprocess.check_call([sys.executable, "run_pick.py", os.path.join(code, "bwa_shim.py"), os.path.join(code, "samview_shim.py"), wf,
                                   str(score), "full", ",".join(ids)], cwd=code, env=env, stdout=subprocess.DEVNULL)
            for cs in cases:
                if cs["score"] == score:
                    d = wf + "velvet_temp/%s/" % cs["id"]
                    assert rd(d + "flanks.sam") == cs["sam"]
                    cs["full"] = [rd(d + "picked_seqs.fa"), rd(d + "picked_contigs.fa")]
            ledgers = [rd(ledger), rd(ledger + "_ori.txt")]
            for cs in cases:
                if cs["score"] == score:
                    cs["ledgers_of_score"] = ledgers if cs["id"] == ids[0] else None
            # the extended pick re-reads the flanks.sam of the full pick (it never runs bwa itself); start from a clean folder like the
            # reference's last round does for gaps the full pick left open
            for cs in cases:
                if cs["score"] == score:
                    d = wf + "velvet_temp/%s/" % cs["id"]
                    for fn in ("picked_seqs.fa", "picked_contigs.fa"):
                        if os.path.exists(d + fn):
                            os.remove(d + fn)
            subprocess.check_call([sys.executable, "run_pick.py", os.path.join(code, "bwa_shim.py"), os.path.join(code, "samview_shim.py"), wf,
                                   str(score), "ext", ",".join(ids)], cwd=code, env=env, stdout=subprocess.DEVNULL)
            for cs in cases:
                if cs["score"] == score:
                    d = wf + "velvet_temp/%s/" % cs["id"]
                    cs["ext"] = "TIE" if os.path.exists(d + "TIE") else [rd(d + "picked_seqs.fa"), rd(d + "picked_contigs.fa")]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with gzip.GzipFile(os.path.join(HERE, "pick_kat.json.gz"), "wb", mtime=0) as gzf:
        gzf.write(json.dumps(cases, sort_keys=True).encode())
    print("pick_kat:", len(cases), "cases,", sum(1 for c in cases if c["full"][0]), "full picks,",
          sum(1 for c in cases if c["ext"] != "TIE" and c["ext"][0]), "extended picks,", sum(1 for c in cases if c["ext"] == "TIE"), "int-vs-str ties")


def merger_kat():
    """Known answers of the reference's own ContigsMerger (its main.cpp + sources built into oracle/_ref/contigs_merger, options of
    MergeContigs.py:75 with -t 1): contig sets -> the NEW_CONTIG_MERGE_n records it prints (sequence + the path of its .info file)."""
    import random
    subprocess.check_call(["make", "-s", "-C", os.path.join(REPO, "oracle"), "ref"])
    exe = os.path.join(REPO, "oracle", "_ref", "contigs_merger")
    rng = random.Random(20260320)
    rnd = lambda n: "".join(rng.choice("ACGT") for _ in range(n))
    comp = {"A": "T", "C": "G", "G": "C", "T": "A"}
    rc = lambda s: "".join(comp[c] for c in reversed(s))

    def noisy(s, subs, indels):
        s = list(s)
        for _ in range(subs):
            i = rng.randrange(len(s)); s[i] = rng.choice([c for c in "ACGT" if c != s[i]])
        for _ in range(indels):
            i = rng.randrange(1, len(s) - 1)
            if rng.random() < 0.5: del s[i]
            else: s.insert(i, rng.choice("ACGT"))
        return "".join(s)
    sets = []
    g = rnd(1600)
    sets.append([g[0:300], g[220:560], g[500:900]])                                             # a chain of three
    sets.append([g[0:300], rc(g[220:560]), g[500:900], rnd(200)])                               # one link reverse-complemented, a stranger
    sets.append([g[500:900], g[0:300], g[220:560]])                                             # the same chain listed out of order
    sets.append([g[0:400], g[100:250], g[300:700], g[650:1000]])                                # a contained contig: no edge
    sets.append([g[0:300], g[250:600], g[250:520] + rnd(150), g[550:900]])                      # a fork: two ways on from c0
    sets.append([g[0:300], g[200:500], g[420:700], g[230:480], g[650:1000]])                    # alternative routes: the larger total overlap
    sets.append([noisy(g[0:350], 3, 1), noisy(g[280:640], 3, 1), noisy(g[580:900], 2, 0)])      # dirty overlaps: clips and the relaxed re-evaluation
    sets.append([g[0:300] + rnd(30), g[260:600]])                                               # a dirty end that must be clipped (-y 50)
    sets.append([rnd(250), rnd(260), rnd(270)])                                                 # nothing overlaps: no new contig
    a, b = rnd(200), rnd(200)
    sets.append([a + b, b + a])                                                                 # a 2-cycle: one strongly connected component
    sets.append([a + b[:100], b[:100] + rnd(80) + a[:90], a[:90] + rnd(50)])                    # a chain through a repeat-like overlap
    for _ in range(30):                                                                         # random tilings of a random genome, random strands
        gg = rnd(rng.randrange(600, 2500))
        cs, pos = [], 0
        while pos < len(gg) - 80:
            ln = rng.randrange(120, 500)
            piece = gg[pos:pos + ln]
            if rng.random() < 0.3:
                piece = noisy(piece, rng.randrange(0, 3), rng.randrange(0, 2))
            cs.append(rc(piece) if rng.random() < 0.5 else piece)
            pos += ln - rng.randrange(20, 110)
        if rng.random() < 0.5:
            cs.append(rnd(rng.randrange(60, 200)))
        rng.shuffle(cs)
        sets.append(cs)
    out = []
    tmp = tempfile.mkdtemp(prefix="gp_mg_")
    try:
        for ci, contigs in enumerate(sets):
            fa = os.path.join(tmp, "c%d.fa" % ci)
            with open(fa, "w") as f:
                f.write("".join(">c%d\n%s\n" % (i, s) for i, s in enumerate(contigs)))
            txt = subprocess.check_output([exe, "-s", "0.4", "-i1", "-2.0", "-i2", "-2.0", "-x", "12", "-y", "50", "-k", "10", "-t", "1", "-m", "1",
                                           "-o", fa + ".info", fa], cwd=tmp).decode()
            recs = []
            for r in txt.split(">")[1:]:
                h, *body = r.split("\n")
                recs.append((h.split()[0], "".join(body)))
            new = [(h, q) for h, q in recs if h.startswith("NEW_CONTIG_MERGE_")]
            assert [q for h, q in recs if not h.startswith("NEW_CONTIG_MERGE_")] == contigs       # the originals follow, unchanged
            info = {l.split()[0]: l.split()[1:] for l in open(fa + ".info").read().splitlines() if l.strip()}
            out.append({"contigs": contigs, "new": [{"seq": q, "path": info[h]} for h, q in new]})
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    with gzip.GzipFile(os.path.join(HERE, "merger_kat.json.gz"), "wb", mtime=0) as gzf:
        gzf.write(json.dumps(out, sort_keys=True).encode())
    print("merger_kat:", len(out), "contig sets,", sum(len(o["new"]) for o in out), "merged contigs,", sum(1 for o in out if not o["new"]), "sets without")


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    kmerutils_kat()
    quickcheck_kat()
    evaluate_kat()
    pick_kat()
    merger_kat()
    for name, seed in (("twolib", 20260001), ("edge", 20260011), ("bounds", 20260031), ("c1", 20260001)):
        case = make_case(name, seed)
        d = os.path.join(HERE, name)
        if os.path.isdir(d):
            shutil.rmtree(d)
        os.makedirs(d)
        kept = run_reference(case, os.path.join(d, "expected.tar.gz"), os.path.join(d, "inputs"))
        print(name, "captured", len(kept), "reference output files")


if __name__ == "__main__":
    main()
