"""tools/synth_files (the synthetic workload of include/gf_synth.h as FASTA + BAM + FASTQ files) for tests and bench.py's `e2e_files` extra."""
import json
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tools", "synth_files")


def build():
    src = os.path.join(ROOT, "tools", "synth_files.c")
    hdr = os.path.join(ROOT, "include", "gf_synth.h")
    if not os.path.exists(BIN) or max(os.path.getmtime(src), os.path.getmtime(hdr)) > os.path.getmtime(BIN):
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-o", BIN, src, "-lz"])
    return BIN


def write_case(root, seed, scaffold_len, n_scaffolds, gaps_per_scaffold, gap_len, libs, kmers, read_len=150, kmer_screen=0, nthreads=4):
    """libs: [(insert_mean, insert_sd, n_pairs)] (library numbers 0, 1, ...).  Returns (config path, working folder)."""
    build()
    data, wf = os.path.join(root, "data"), os.path.join(root, "wf")
    os.makedirs(data, exist_ok=True)
    os.makedirs(wf, exist_ok=True)
    for no, (is_, sd, n_pairs) in enumerate(libs):
        subprocess.check_call([BIN, data, str(seed), str(scaffold_len), str(n_scaffolds), str(gaps_per_scaffold), str(gap_len), str(read_len),
                               str(is_), str(sd), str(no), str(n_pairs)], stderr=subprocess.DEVNULL)
    by_k = {}
    for k, kv in kmers:
        by_k.setdefault(k, []).append(kv)
    cfg = {"draft_genome": {"fa": os.path.join(data, "draft.fa")},
           "raw_reads": [{"left": os.path.join(data, "lib%d_1.fq" % no), "right": os.path.join(data, "lib%d_2.fq" % no)} for no in range(len(libs))],
           "alignments": [{"bam": os.path.join(data, "lib%d.bam" % no), "is": str(is_), "std": str(sd)} for no, (is_, sd, _) in enumerate(libs)],
           "software_path": {"bwa": "bwa", "samtools": "builtin", "velvet": "/x/", "kmc": "/x/", "TERefiner": "x", "ContigsMerger": "x"},
           "parameters": {"working_folder": wf, "min_gap_size": "100", "flank_length": "300", "nthreads": str(nthreads), "verbose": "0",
                          "kmer_screen": kmer_screen},
           "kmer_length": [{"k": k, "k_velvet": [{"k": kv} for kv in kvs]} for k, kvs in by_k.items()]}
    cfgp = os.path.join(root, "cfg.json")
    json.dump(cfg, open(cfgp, "w"))
    return cfgp, wf + "/"
