"""Test-side helpers: SAM text -> 32-byte records (independent of the product's own parser), hit -> list line."""
import numpy as np

from oracle import gp_oracle as O
from oracle.c_oracle import ALNREC, DPOS, GAP

KIND = {0: "clip", 1: "discordant", 2: "unmap"}


def gaps_array(fai_names, gaps):
    """gaps: [(start, end, len, scaffold)] in file order -> structured array + per-gap (scaffoldIdx, n)."""
    idx = {n: i for i, n in enumerate(fai_names)}
    arr = np.zeros(len(gaps), dtype=GAP)
    cnt, pre = 1, None
    for i, (s, e, _, scf) in enumerate(gaps):
        if scf != pre:
            cnt = 1
        arr[i] = (idx[scf], s, e, cnt)
        cnt += 1
        pre = scf
    return arr


def sam_to_records(sam_text, fai_names):
    idx = {n: i for i, n in enumerate(fai_names)}
    lines = [l for l in sam_text.splitlines() if l and not l.startswith("@")]
    recs = np.zeros(len(lines), dtype=ALNREC)
    fields = []
    for i, l in enumerate(lines):
        f = l.split()
        fields.append(f[:9])
        ref = idx.get(f[2], 0xFFFFFFFF)
        mref = ref if f[6] == "=" else idx.get(f[6], 0xFFFFFFFF)
        flag = int(f[1])
        # read id: pair index = order of first appearance of the qname is not needed here; use the line index
        recs[i] = (int(f[3]), int(f[7]), int(f[8]), ref, mref, flag, int(f[4]), O.is_clipped(f[5]), i)
    return recs, fields


def hits_to_lines(hits, recs, fields, gaps_arr, fai_names):
    """Tagger hits -> {scaffold: {'left': [...], 'right': [...]}} in the reference's list-line format."""
    out = {}
    for h in hits:
        f = fields[h["rec"]]
        g = gaps_arr[h["gap"]]
        scf = fai_names[g["scaffold"]]
        flag = int(f[1])
        own = "left" if flag & 0x40 else "right"
        mate = "right" if own == "left" else "left"
        side = mate if h["to_mate"] else own
        if h["kind"] == 1:
            line = "%s %d %s discordant %s %s %s %d" % (f[0], g["idx_in_scaffold"], f[4], f[3], f[6], f[7],
                                                      int(g["end"]) - int(g["start"]))
        else:
            line = "%s %d %s %s" % (f[0], g["idx_in_scaffold"], f[4], KIND[int(h["kind"])])
        out.setdefault(scf, {"left": [], "right": []})[side].append(line)
    return out


def dpos_array(rows):
    arr = np.zeros(len(rows), dtype=DPOS)
    for i, r in enumerate(rows):
        arr[i] = r
    return arr


def lowmapq_hits_to_lines(hits, fields, table, fai_names):
    out = {}
    for h in hits:
        f = fields[h["rec"]]
        row = table[h["gap"]]
        side = "left" if int(f[1]) & 0x40 else "right"
        out.setdefault(f[2], {"left": [], "right": []})[side].append(
            "%s %d_%d %d" % (f[0], row["src_scaffold"], row["src_gap"], int(f[4])))
    return out


def fastq_seqs(text):
    lines = text.split("\n")
    return [lines[i + 1] for i in range(0, len(lines) - 3, 4)]
