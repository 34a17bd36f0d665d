"""Contig dedup + overlap merge (the reference's MergeContigs.py:15-99 runs TERefiner -U / bwa / TERefiner -P,
-K and ContigsMerger per gap).  Built so far: the FIRST stage of ContigsMerger — the all-pairs 10-mer prefilter that decides
which contig pairs are worth an overlap alignment (QuickCheckerContigsMatch, ContigsCompactor.cpp:1982-2095) — on the GPU for all
gaps of a run at once (gf_quick_check, csrc/merge.hip; pinned on answers of the reference's own code, tests/golden/
quickcheck_kat.json.gz) — and the SECOND: the pairwise overlap evaluation of the surviving pairs (ContigsCompactor::Evaluate,
ContigsCompactor.cpp:1572-1976: overlap alignment, end clipping, IsScoreSignificant, containment; gf_overlap_evaluate, pinned on
the reference's own answers, tests/golden/evaluate_kat.json.gz), which yields the EDGES of the merger's overlap graph exactly as
threadMergeContigV2 builds them (:624-690).  NOT built: the path search over that graph (GraphUtils.cpp:625-859) and the bwa/
TERefiner dedup; `contigs.fa` is therefore left as the assembly wrote it, and this module records next to it the candidate pairs
(velvet_temp/{id}/merge_candidates.txt: 'nameA strandA nameB strandB') and the edges (merge_edges.txt: 'nameA strandA nameB
strandB mode overlap', mode 12 = A then B, 21 = B then A; ContigsMerger's -s 0.4 -i1 -2.0 -i2 -2.0 -x 12 -y 50 of
MergeContigs.py:75)."""
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


def merge_edges(gf, working_folder, id_list, kmer_len_quick=10, params=None):
    """The edges of ContigsMerger's overlap graph for every gap's contigs.fa: every feasible node pair (i <= j, the pairs
    MultiThreadQuickChecker::threadQuickCheck collects, ContigsCompactor.cpp:1068-1098) is evaluated in that order and forms an edge
    when its overlap is at least -x long and no containment (threadMergeContigV2, :652-688).  Two GPU calls for the whole batch.
    Returns {gap id: [(i, j, mode, overlap)]}, mode '12' (node i then node j) or '21'; writes merge_edges.txt next to contigs.fa."""
    ids, sets, names = [], [], []
    for gid in id_list:
        p = "%svelvet_temp/%s/contigs.fa" % (working_folder, gid)
        if not os.path.exists(p):
            continue
        recs = [(n, s) for n, s in read_fasta(p) if 30 <= len(s) <= 8190]
        if not recs:
            continue
        ids.append(gid)
        names.append([n for n, _ in recs])
        sets.append([s for _, s in recs])
    out = {gid: [] for gid in ids}
    if not sets:
        return out
    pairs = gf.quick_check(sets, kmer_len_quick)
    res = gf.overlap_evaluate(sets, pairs, params)
    for t, r in zip(pairs, res):
        if int(r["res"]) == 2 and not int(r["containment"]):
            out[ids[int(t["set"])]].append((int(t["i"]), int(t["j"]), "12" if int(r["first_goes_first"]) else "21", int(r["overlap"])))
    for gid, nm in zip(ids, names):
        with open("%svelvet_temp/%s/merge_edges.txt" % (working_folder, gid), "w") as f:
            for i, j, mode, ov in out[gid]:
                f.write("%s %s %s %s %s %d\n" % (nm[i // 2], "-" if i & 1 else "+", nm[j // 2], "-" if j & 1 else "+", mode, ov))
    return out
