#!/usr/bin/env python3
"""CPU experiment (oracle only): which mate-pair depth / simplification closes 2-kb gaps?  Small C5-shaped draft, the oracle's
recruitment (k-mer screen + tagger, both libraries), per-gap pools, or_assemble_pool2 at (31,29),(41,39),(51,49), host picker."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import c_oracle as CO
from gappadder_amd.pick_contigs import pick_gap_sequence

def run(mp_cov, simplify, short_cov=43.0, slen=400_000, gps=6, seed=20260004, kk=((31, 29), (41, 39), (51, 49)), err=0.005):
    L = 150
    libs = [(300, 30, 0, short_cov), (5000, 500, 1, mp_cov)]
    cfg0 = CO.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=1, gaps_per_scaffold=gps, gap_len=2000, err=err)
    gaps, flanks = CO.synth_layout(cfg0)
    pools = [[] for _ in gaps]
    for is_mean, is_sd, lib, cov in libs:
        n_pairs = int(cov * slen / L / 2)
        if n_pairs == 0:
            continue
        cfg = CO.synth_cfg(seed=seed, scaffold_len=slen, n_scaffolds=1, gaps_per_scaffold=gps, gap_len=2000, insert_mean=is_mean, insert_sd=is_sd, library=lib, err=err)
        packed, recs = CO.synth_pairs(cfg, 0, n_pairs)
        blob = CO.unpack_reads(packed, L)
        keys = set()
        for h in CO.screen_reads(blob, L, flanks, min(k for k, _ in kk), threads=8):
            keys.add((int(h["gap"]), int(h["read"]))); keys.add((int(h["gap"]), int(h["read"]) ^ 1))
        for h in CO.tag_alignments(recs, gaps, is_mean, is_sd):
            keys.add((int(h["gap"]), int(recs[h["rec"]]["read"]) ^ int(h["to_mate"])))
        for g in range(len(gaps)):
            ids = sorted(r for gg, r in keys if gg == g)
            ids.sort(key=lambda r: (r & 1, r >> 1))
            pools[g] += [blob[i * L:(i + 1) * L] for i in ids]
    closed, nctg = 0, 0
    for g, p in enumerate(pools):
        ctgs = []
        for k, kv in kk:
            ctgs += [("c", s) for s, _, _ in CO.assemble_pool(b"".join(p), L, k, kv, simplify=simplify)]
        nctg += len(ctgs)
        if any(pick_gap_sequence(ctgs, flanks[g][0], flanks[g][1], a) for a in (30, 15)):
            closed += 1
    return closed, len(gaps), nctg, sum(len(p) for p in pools) / len(pools)

if __name__ == "__main__":
    for mp in (4.8, 10, 20, 30):
        for simp in (0, 2, 4):
            t = time.time()
            print("mp_cov %5.1f simplify %d -> closed %d / %d, contigs %d, reads/gap %.0f  (%.1fs)" % ((mp, simp) + run(mp, simp) + (time.time() - t,)), flush=True)
