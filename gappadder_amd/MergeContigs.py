"""Contig dedup + overlap merge (the reference's MergeContigs.py:15-99 runs TERefiner -U / bwa / TERefiner -P,
-K and ContigsMerger per gap).  Built so far: the FIRST stage of ContigsMerger — the all-pairs 10-mer prefilter that decides
which contig pairs are worth an overlap alignment (QuickCheckerContigsMatch, ContigsCompactor.cpp:1982-2095) — on the GPU for all
gaps of a run at once (gf_quick_check, csrc/merge.hip; pinned on answers of the reference's own code, tests/golden/
quickcheck_kat.json.gz).  NOT built: the overlap DP of the surviving pairs (ContigsCompactor.cpp:1572-1976), the overlap graph /
path search (GraphUtils.cpp:625-859) and the bwa/TERefiner dedup; `contigs.fa` is therefore left as the assembly wrote it, and
this module only records the candidate pairs next to it (velvet_temp/{id}/merge_candidates.txt: 'nameA strandA nameB strandB')."""
import os

from .pick_contigs import read_fasta


def merge_candidates(gf, working_folder, id_list, kmer_len_quick=10):
    """Feasible (contig, strand) pairs of every gap's contigs.fa, one GPU call for the batch.  Returns {gap id: [(i, j)]} with
    nodes numbered 2 * contig + strand as in CompactVer3 (ContigsCompactor.cpp:782-800), pairs (i, i) and the pair of a contig
    with its own reverse complement left out (they are trivially feasible)."""
    ids, sets, names = [], [], []
    for gid in id_list:
        p = "%svelvet_temp/%s/contigs.fa" % (working_folder, gid)
        if not os.path.exists(p):
            continue
        recs = [(n, s) for n, s in read_fasta(p) if len(s) >= 30]
        if not recs:
            continue
        ids.append(gid)
        names.append([n for n, _ in recs])
        sets.append([s for _, s in recs])
    out = {gid: [] for gid in ids}
    if not sets:
        return out
    for t in gf.quick_check(sets, kmer_len_quick):
        s, i, j = int(t["set"]), int(t["i"]), int(t["j"])
        if i // 2 != j // 2:
            out[ids[s]].append((i, j))
    for gid, nm in zip(ids, names):
        with open("%svelvet_temp/%s/merge_candidates.txt" % (working_folder, gid), "w") as f:
            for i, j in out[gid]:
                f.write("%s %s %s %s\n" % (nm[i // 2], "-" if i & 1 else "+", nm[j // 2], "-" if j & 1 else "+"))
    return out
