"""Contig dedup + overlap merge (the reference's MergeContigs.py:15-99 runs, per gap, TERefiner -P -g behind a
bwa self-alignment, ContigsMerger, and TERefiner again).  Here, for all gaps of a round at once:

  * ContigsMerger itself, stage by stage: the all-pairs 10-mer prefilter (QuickCheckerContigsMatch, ContigsCompactor.cpp:1982-2095)
    and the pairwise overlap evaluation (ContigsCompactor::Evaluate, :1572-1976) on the GPU (gf_quick_check, gf_overlap_evaluate,
    csrc/merge.hip) give the EDGES of its overlap graph (threadMergeContigV2 :624-690, addEdges :724-770); the path search over that
    graph (strongly connected components in topological order, candidate roots and ends, a shortest-path DP per root with -overlap as
    edge length, GraphUtils.cpp:625-859, 1028-1178, 1258-1344), the removal of reverse-complement twin paths (:1422-1454) and the
    merged strings (FormMergedSeqFromPath :1456-1520: the running string against the next node with Evaluate in its relaxed mode —
    one batched GPU call per path step — joined by SetMergedStringConcat :108-153) run on the host: tens of nodes per gap.  Every
    stage is pinned on answers of the reference's own code (tests/golden/{quickcheck,evaluate,merger}_kat.json.gz through oracle/).
    The reference orders several containers by pointer value (and, with its -t 5, edge lists by thread timing); this module uses the
    allocation order — node index, insertion order — i.e. the single-threaded reference.
  * the dedup around it (`bwa mem -a` of the contigs against themselves + TERefiner -P -g / -P, refiner.cpp:660-801) with bwa's place
    taken by EXACT containment: a contig whose sequence occurs, on either strand, inside another contig of the set is "perfectly
    mapped" there (Alignment.cpp:428-437) and is dropped.  Deliberate deviation: of IDENTICAL contigs the reference drops every
    copy (each maps perfectly onto the other); here the first copy stays.  After that step no contig lies inside another one, so
    TERefiner's second mode (-P without -g: nearly equal lengths, refiner.cpp:700-760) finds nothing under exact matching.
  * the file protocol of merge_contigs (MergeContigs.py:66-99): contigs.fa_no_dup.fa, …merge.info, …merged.fa, the renamed
    original_contigs_before_merging.fa and the new contigs.fa; merging is skipped when the de-duplicated file exceeds 1 MB (:70-74).

Contigs shorter than 30 or longer than 8190 bases (the overlap kernel's LDS diagonals) take no part in the merging and pass
through unchanged.  merge_edges.txt next to contigs.fa lists the graph edges ('nameA strandA nameB strandB mode overlap')."""
import os

from .pick_contigs import read_fasta, revcomp

MIN_NODE, MAX_NODE = 30, 8190
MAX_SET = 4096                    # contigs per gap handled by one call (gf_quick_check's pair matrix grows with the square)
MAX_PATHS_PER_ROOT = 20           # ContigsCompactor.cpp:34 MAX_CONTIG_IN_PATH_COUNT


def _sets(working_folder, id_list):
    ids, recs = [], []
    for gid in id_list:
        p = "%svelvet_temp/%s/contigs.fa" % (working_folder, gid)
        if os.path.exists(p):
            r = read_fasta(p)
            if r:
                ids.append(gid)
                recs.append(r)
    return ids, recs


def drop_contained(recs):
    """[(name, seq)] without the contigs that occur exactly, on either strand, inside another one (see the module text)."""
    seqs = [s.upper() for _, s in recs]
    rcs = [revcomp(s) for s in seqs]
    keep, kept_text = [], ""      # the kept contigs and their reverse complements, newline-separated: one substring search per candidate
    for i in sorted(range(len(recs)), key=lambda i: (-len(seqs[i]), i)):     # longest first: a contig lies inside one at least as long
        q = seqs[i]
        if not keep or q not in kept_text:
            keep.append(i)
            kept_text += "\n" + q + "\n" + rcs[i]
    return [recs[i] for i in sorted(keep)]


def graph_edges(gf, sets, kmer_len_quick=10, params=None):
    """Per contig set the adjacency lists of the merger's graph: adj[v] = [(w, -overlap)] in the order the reference adds the
    edges (pairs i <= j in the prefilter's order; mode 12: i -> j, mode 21: j -> i), and the raw edge list for merge_edges.txt.
    Two GPU calls for the whole batch."""
    adjs = [[[] for _ in range(2 * len(s))] for s in sets]
    edges = [[] for _ in sets]
    if not any(sets):
        return adjs, edges
    pairs = gf.quick_check(sets, kmer_len_quick)
    res = gf.overlap_evaluate(sets, pairs, params)
    for t, r in zip(pairs, res):
        if int(r["res"]) == 2 and not int(r["containment"]):
            s, i, j, ov = int(t["set"]), int(t["i"]), int(t["j"]), int(r["overlap"])
            first = bool(int(r["first_goes_first"]))
            edges[s].append((i, j, "12" if first else "21", ov))
            if first:
                adjs[s][i].append((j, -float(ov)))
            else:
                adjs[s][j].append((i, -float(ov)))
    return adjs, edges


def _components(adj):
    """Strongly connected components in the order AbstractGraph::SCC returns them (Tarjan from node 0 up, neighbours in edge
    order, result reversed = topological), every component sorted.  Iterative: a chain of contigs is as deep as it is long."""
    n = len(adj)
    index, low, on = [-1] * n, [0] * n, [False] * n
    stack, out, counter = [], [], 1
    for root in range(n):
        if index[root] >= 0:
            continue
        work = [(root, 0)]
        index[root] = low[root] = counter
        counter += 1
        stack.append(root)
        on[root] = True
        while work:
            v, ei = work[-1]
            if ei < len(adj[v]):
                work[-1] = (v, ei + 1)
                w = adj[v][ei][0]
                if index[w] < 0:
                    index[w] = low[w] = counter
                    counter += 1
                    stack.append(w)
                    on[w] = True
                    work.append((w, 0))
                elif on[w]:
                    low[v] = min(low[v], index[w])
                continue
            work.pop()
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop()
                    on[w] = False
                    comp.append(w)
                    if w == v:
                        break
                out.append(sorted(comp))
    return out[::-1]


def _terminals(comps, comp_of, adj, start):
    """FindSimplePathsTopSortStart: nodes no edge from another component enters (start) / that have no edge into another
    component (ends); of a component with several nodes only its first (last) node, and only when all of them qualify."""
    cand = set(comp_of)
    for comp in comps:
        for v in comp:
            for w, _ in adj[v]:
                if comp_of[w] != comp_of[v]:
                    if start:
                        cand.discard(w)
                    else:
                        cand.discard(v)
                        break
    for comp in comps:
        if len(comp) > 1:
            whole = all(v in cand for v in comp)
            keep = comp[0] if start else comp[-1]
            cand.difference_update(v for v in comp if v != keep)
            if not whole:
                cand.discard(keep)
    return sorted(cand)


def find_paths(adj, max_per_root=MAX_PATHS_PER_ROOT):
    """FindSimplePathsTopSort: sorted list of node-index tuples."""
    comps = _components(adj)
    order = [v for comp in comps for v in comp]
    pos = {v: i for i, v in enumerate(order)}
    comp_of = {v: ci for ci, comp in enumerate(comps) for v in comp}
    roots, ends = _terminals(comps, comp_of, adj, True), _terminals(comps, comp_of, adj, False)
    found = set()
    for root in roots:
        best = {pos[root]: (0.0, (root,))}
        for i in range(pos[root], len(order)):
            if i not in best:
                continue
            d, path = best[i]
            first_len = {}
            for w, length in adj[order[i]]:
                first_len.setdefault(w, length)             # GetEdgeTo: the first edge to a node counts
            for w, _ in adj[order[i]]:
                j = pos[w]
                if j >= i and (j not in best or d + first_len[w] < best[j][0]):
                    best[j] = (d + first_len[w], path + (w,))
        got = []
        for e in ends:
            if pos[e] in best and best[pos[e]][1] not in got:
                got.append(best[pos[e]][1])
        ranked = sorted(range(len(got)), key=lambda q: (-len(got[q]), q))
        found.update(got[q] for q in ranked[:max_per_root + 1])
    return sorted(found)


def merged_strings(gf, jobs, params=None):
    """jobs = [(nodes of a set, path)] -> merged sequence per job (FormMergedSeqFromPath): all jobs advance one node per round,
    one gf_overlap_evaluate call (relaxed mode) per round."""
    cur = [nodes[path[0]] for nodes, path in jobs]
    step = 1
    while True:
        live = [q for q, (_, path) in enumerate(jobs) if step < len(path)]
        if not live:
            return cur
        sets, pairs = [], []
        for n_set, q in enumerate(live):
            nxt = jobs[q][0][jobs[q][1][step]]
            if len(cur[q]) > MAX_NODE or len(nxt) > MAX_NODE:       # grown beyond the kernel's reach: the path ends here
                jobs[q] = (jobs[q][0], jobs[q][1][:step])
                continue
            sets.append([cur[q], nxt])
            pairs.append((len(sets) - 1, 0, 2, q))
        if sets:
            import numpy as np
            from . import _lib as B
            pp = np.zeros(len(pairs), dtype=B.QCPAIR)
            for x, (s, i, j, _) in enumerate(pairs):
                pp[x] = (s, i, j)
            res = gf.overlap_evaluate(sets, pp, params, relax=True)
            for (s, _, _, q), r in zip(pairs, res):
                s1, s2 = sets[s]
                n1, n2, re_, ce, nc = len(s1), len(s2), int(r["row_end"]), int(r["col_end"]), int(r["nclip"])
                if int(r["contained"]) and re_ + nc == n1 and n1 < n2:
                    cur[q] = s2
                elif int(r["contained"]) and ce + nc == n2 and n2 < n1:
                    pass
                elif re_ + nc == n1:
                    cur[q] = s1[:n1 - nc] + s2[ce:]
                else:
                    cur[q] = s2[:n2 - nc] + s1[re_:]
        step += 1


def _node_name(names, v):
    return names[v >> 1] + ("_R" if v & 1 else "")


def _write_fasta(path, recs, width=0):
    with open(path, "w") as f:
        for n, s in recs:
            if width:
                f.write(">%s\n%s" % (n, "".join(s[i:i + width] + "\n" for i in range(0, len(s), width))))
            else:
                f.write(">%s\n%s\n" % (n, s))


def merge_edges(gf, working_folder, id_list, kmer_len_quick=10, params=None):
    """The edges of ContigsMerger's overlap graph for every gap's contigs.fa as it stands.  Returns {gap id: [(i, j, mode,
    overlap)]}, mode '12' (node i then node j) or '21'; writes merge_edges.txt next to contigs.fa."""
    ids, recs = _sets(working_folder, id_list)
    keep = [(gid, [(n, s) for n, s in r if MIN_NODE <= len(s) <= MAX_NODE]) for gid, r in zip(ids, recs)]
    keep = [(gid, r) for gid, r in keep if r and len(r) <= MAX_SET]
    _, edges = graph_edges(gf, [[s for _, s in r] for _, r in keep], kmer_len_quick, params)
    out = {}
    for (gid, r), ed in zip(keep, edges):
        out[gid] = ed
        with open("%svelvet_temp/%s/merge_edges.txt" % (working_folder, gid), "w") as f:
            for i, j, mode, ov in ed:
                f.write("%s %s %s %s %s %d\n" % (r[i // 2][0], "-" if i & 1 else "+", r[j // 2][0], "-" if j & 1 else "+", mode, ov))
    return out


def _swap_in(folder, new_name):
    """contigs.fa := folder/new_name, the assembly's own contigs kept as original_contigs_before_merging.fa.  The backup is a COPY
    and contigs.fa is replaced in one rename, so that a failure at either step leaves the gap WITH a contigs.fa (the picking and
    bridging steps that follow look for it)."""
    import shutil
    shutil.copyfile(folder + "contigs.fa", folder + "original_contigs_before_merging.fa")
    os.replace(folder + new_name, folder + "contigs.fa")


def merge_sets(gf, rec_sets, kmer_len_quick=10, params=None):
    """The merger on contig sets in memory: rec_sets = [[(name, seq)]] (one set per gap, already de-duplicated) -> per set
    {"nodes": the records that took part, "edges": [(i, j, mode, overlap)], "new": [(NEW_CONTIG_MERGE_n, seq, path)]}.
    Two batched GPU calls for the graph edges of all sets + one per path step for the merged strings."""
    nodes_of = [[(n, s.upper()) for n, s in r if MIN_NODE <= len(s) <= MAX_NODE] for r in rec_sets]
    sets = [[s for _, s in nodes] for nodes in nodes_of]
    adjs, edges = graph_edges(gf, sets, kmer_len_quick, params)
    jobs, owner = [], []
    for wi, (nodes, adj, ed) in enumerate(zip(nodes_of, adjs, edges)):
        node_seqs = []
        for _, s in nodes:
            node_seqs += [s, revcomp(s)]
        paths = find_paths(adj) if ed else []
        kept = [p for i, p in enumerate(paths) if tuple(v ^ 1 for v in reversed(p)) not in paths[:i]]      # RemoveDupRevCompPaths
        for p in kept:
            if len(p) > 1:
                jobs.append((node_seqs, p))
                owner.append(wi)
    merged = merged_strings(gf, jobs, params) if jobs else []
    per = {}
    for wi, (nodes_p, seq) in zip(owner, zip(jobs, merged)):
        per.setdefault(wi, []).append((nodes_p[1], seq))
    return [{"nodes": nodes, "edges": ed,
             "new": [("NEW_CONTIG_MERGE_%d" % (q + 1), seq, path) for q, (path, seq) in enumerate(per.get(wi, []))]}
            for wi, (nodes, ed) in enumerate(zip(nodes_of, edges))]


def merge_contigs(gf, working_folder, id_list, kmer_len_quick=10, params=None):
    """merge_contigs (MergeContigs.py:66-99) for every gap of id_list.  Returns {gap id: number of NEW_CONTIG_MERGE records}."""
    ids, recs = _sets(working_folder, id_list)
    work = []                                                  # (gid, folder, de-duplicated records)
    done = {}
    for gid, r in zip(ids, recs):
        folder = "%svelvet_temp/%s/" % (working_folder, gid)
        nodup = drop_contained(r) if len(r) <= MAX_SET else r
        _write_fasta(folder + "contigs.fa_no_dup.fa", nodup)
        if os.path.getsize(folder + "contigs.fa_no_dup.fa") > 1000000 or len(nodup) > MAX_SET:      # MergeContigs.py:70-74
            try:
                _swap_in(folder, "contigs.fa_no_dup.fa")
            except OSError as e:
                import sys
                sys.stderr.write("contig merging: gap %s keeps its contigs (%r)\n" % (gid, e))
            done[gid] = 0
            continue
        work.append((gid, folder, nodup))
    for (gid, folder, nodup), m in zip(work, merge_sets(gf, [w[2] for w in work], kmer_len_quick, params)):
        nodes, new = m["nodes"], m["new"]
        with open(folder + "merge_edges.txt", "w") as f:
            for i, j, mode, ov in m["edges"]:
                f.write("%s %s %s %s %s %d\n" % (nodes[i // 2][0], "-" if i & 1 else "+", nodes[j // 2][0], "-" if j & 1 else "+", mode, ov))
        names = [n for n, _ in nodes]
        with open(folder + "contigs.fa_no_dup.fa.merge.info", "w") as f:
            for name, _, path in new:
                f.write("%s   %s\n" % (name, " ".join(_node_name(names, v) for v in path)))
        merged_recs = [(n, s) for n, s, _ in new] + nodup
        _write_fasta(folder + "contigs.fa_no_dup.fa.merged.fa", merged_recs, 60)                  # ContigsMerger dumps 60 columns
        final = drop_contained(merged_recs)
        try:            # per gap: a failed write leaves THIS gap's contigs.fa as it was (the merged set goes to a temporary name first)
            _write_fasta(folder + "contigs.fa.merged.tmp", final)
            _swap_in(folder, "contigs.fa.merged.tmp")
            done[gid] = len(new)
        except OSError as e:
            import sys
            sys.stderr.write("contig merging: gap %s keeps its contigs (%r)\n" % (gid, e))
            done[gid] = 0
    return done
