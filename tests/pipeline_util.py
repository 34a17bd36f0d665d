"""Materialise a golden case on disk (draft, .fai, SAM text behind a samtools stand-in, FASTQ pairs, config JSON)."""
import json
import os
import stat

SAMTOOLS_STANDIN = '''#!/usr/bin/env python3
# test stand-in for `samtools view <bam> "<scaffold>"`: serves <bam>.sam text
import sys
if sys.argv[1] == "view" and sys.argv[2] == "-f":       # view -f <mask> <bam>
    with open(sys.argv[4] + ".sam") as f:
        for line in f:
            if int(line.split("\\t")[1]) & int(sys.argv[3]) == int(sys.argv[3]):
                sys.stdout.write(line)
elif sys.argv[1] == "view":
    with open(sys.argv[2] + ".sam") as f:
        for line in f:
            if line.split("\\t")[2] == sys.argv[3]:
                sys.stdout.write(line)
'''


def materialise(case, root, kmers=((31, 29),), kmer_screen=0, builtin_bam=False):
    data, wf = os.path.join(root, "data"), os.path.join(root, "wf")
    os.makedirs(data)
    os.makedirs(wf)
    draft = os.path.join(data, "draft.fa")
    open(draft, "w").write(case.draft_fa)
    open(draft + ".fai", "w").write(case.fai)
    st = os.path.join(data, "samtools_standin.py")
    open(st, "w").write(SAMTOOLS_STANDIN)
    os.chmod(st, os.stat(st).st_mode | stat.S_IEXEC)
    libs = []
    for i, lib in enumerate(case.libs):
        bam = os.path.join(data, "lib%d.bam" % i)
        if builtin_bam:      # a real BAM (test-side writer) for software_path.samtools = "builtin"; no stand-in text beside it
            import bam_util
            names = [l.split()[0] for l in case.fai.splitlines() if l.strip()]
            lens = [int(l.split()[1]) for l in case.fai.splitlines() if l.strip()]
            stream = bam_util.sam_to_bam_stream(lib["sam"].splitlines(), names[::-1], lens[::-1])   # header order != .fai order
            open(bam, "wb").write(bam_util.bgzf_compress(stream, block=20000, seed=i + 1, levels=(6, 1, "fixed")))
        else:
            open(bam, "w").close()
            open(bam + ".sam", "w").write(lib["sam"])
        l, r = os.path.join(data, "lib%d_1.fq" % i), os.path.join(data, "lib%d_2.fq" % i)
        open(l, "w").write(lib["fq1"])
        open(r, "w").write(lib["fq2"])
        libs.append((bam, l, r, lib["is"], lib["sd"]))
    by_k = {}
    for k, kv in kmers:
        by_k.setdefault(k, []).append(kv)
    cfg = {"draft_genome": {"fa": draft},
           "raw_reads": [{"left": l, "right": r} for (_, l, r, _, _) in libs],
           "alignments": [{"bam": b, "is": str(i), "std": str(s)} for (b, _, _, i, s) in libs],
           "software_path": {"bwa": "bwa", "samtools": "builtin" if builtin_bam else st, "velvet": "/x/", "kmc": "/x/", "TERefiner": "x", "ContigsMerger": "x"},
           "parameters": {"working_folder": wf, "min_gap_size": str(case.meta["min_gap"]), "flank_length": str(case.meta["flank"]),
                          "nthreads": "2", "verbose": "0", "kmer_screen": kmer_screen},
           "kmer_length": [{"k": k, "k_velvet": [{"k": kv} for kv in kvs]} for k, kvs in by_k.items()]}
    cfgp = os.path.join(root, "cfg.json")
    json.dump(cfg, open(cfgp, "w"))
    return cfgp, wf + "/", st


def tree(wf):
    out = {}
    for r, _, files in os.walk(wf):
        for fn in files:
            p = os.path.join(r, fn)
            out[os.path.relpath(p, wf)] = open(p).read()
    return out
