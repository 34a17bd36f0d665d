"""Shared code of the level-1 executable shims (kmc, kmc_dump, velveth, velvetg): same command lines as the reference
issues at assemble_gaps.py:96-118, arithmetic on the GPU through libgapfill_hip.so."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from gappadder_amd import fastq_io  # noqa: E402
from gappadder_amd.assemble_gaps import format_contigs, velvet_kv  # noqa: E402
from gappadder_amd.hip_api import GapFill  # noqa: E402


def _ctx():
    return GapFill(int(os.environ.get("GF_DEVICE", "0")))


def kmer_string(hi, lo, k):
    v = (int(hi) << 64) | int(lo)
    return "".join("ACGT"[(v >> (126 - 2 * i)) & 3] for i in range(k))


def kmc(argv):
    """kmc -k{K} [-ci{n}] [-cs..] [-m..] in.fastq out_prefix tmpdir  ->  out_prefix.gfkmc (JSON: k + counted k-mers)"""
    k, ci, pos = 25, 2, []
    for a in argv:
        if a.startswith("-k"):
            k = int(a[2:])
        elif a.startswith("-ci"):
            ci = int(a[3:])
        elif a.startswith("-"):
            continue
        else:
            pos.append(a)
    src, prefix = pos[0], pos[1]
    seqs = fastq_io.read_fastq_seqs(src) if os.path.exists(src) else []
    recs = []
    if seqs and 16 <= k <= 64:
        packed, nm, _, L = fastq_io.pack_pools([seqs])
        if L >= k:
            km, cn = _ctx().count_kmers(packed, L, k, ci, n_mask=nm)
            recs = [[kmer_string(a, b, k), int(c)] for (a, b), c in zip(km, cn)]
    with open(prefix + ".gfkmc", "w") as f:
        json.dump({"k": k, "kmers": recs}, f)


def kmc_dump(argv):
    """kmc_dump [-ci{n}] [-cx{n}] in_prefix out.dump  ->  'KMER<TAB>COUNT' lines, ascending"""
    ci, pos = 0, []
    for a in argv:
        if a.startswith("-ci"):
            ci = int(a[3:])
        elif a.startswith("-"):
            continue
        else:
            pos.append(a)
    with open(pos[0] + ".gfkmc") as f:
        d = json.load(f)
    with open(pos[1], "w") as f:
        f.write("".join("%s\t%d\n" % (s, c) for s, c in d["kmers"] if c >= ci))


def velveth(argv):
    """velveth dir KV -fastq -short reads.fq  ->  dir/gf_velveth.json"""
    d, kv = argv[0], int(argv[1])
    files = [a for a in argv[2:] if not a.startswith("-")]
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "gf_velveth.json"), "w") as f:
        json.dump({"kv": kv, "reads": files}, f)


def velvetg(argv):
    """velvetg dir [-min_contig_lgth N]  ->  dir/contigs.fa (always written)"""
    d, min_contig = argv[0], 0
    for i, a in enumerate(argv):
        if a == "-min_contig_lgth":
            min_contig = int(argv[i + 1])
    with open(os.path.join(d, "gf_velveth.json")) as f:
        meta = json.load(f)
    reads = []
    for p in meta["reads"]:
        if os.path.exists(p):
            # cvtFaToFq leaves '\t{count}' on the sequence line (assemble_gaps.py:64-77): the read is the first field
            reads += [s.split()[0] for s in fastq_io.read_fastq_seqs(p) if s.split()]
    txt = ""
    kv = velvet_kv(meta["kv"])
    if reads:
        k = len(reads[0])
        reads = [r for r in reads if len(r) == k]
        if 16 <= k <= 64 and 15 <= kv < k:
            packed, nm, off, L = fastq_io.pack_pools([reads])
            ctg, seq = _ctx().assemble(packed, off, L, [(k, kv)], 1, min_contig, n_mask=nm)
            txt = format_contigs([(seq[int(c["seq_off"]):int(c["seq_off"]) + int(c["length"])].decode(), int(c["n_nodes"]),
                                   int(c["cov_sum"])) for c in ctg])
    with open(os.path.join(d, "contigs.fa"), "w") as f:
        f.write(txt)
