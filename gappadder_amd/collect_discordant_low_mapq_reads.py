"""Drop-in twin of collect_discordant_low_mapq_reads.py (second hop: MAPQ==0 reads next to a discordant mate position),
arithmetic on the GPU (gf_tag_low_mapq).

    samtools view BAM "scf" | python -m gappadder_amd.collect_discordant_low_mapq_reads wf -
"""
import os
import sys

import numpy as np

from . import _lib as B
from . import sam_io
from .hip_api import GapFill


def read_rows(path):
    """discordant_temp/{scaffold}.list rows 'mIdx mPos sIdx gIdx' (already sorted by the reference's sort(1))."""
    rows = []
    with open(path) as f:
        for line in f:
            fl = line.split()
            if len(fl) >= 4:
                rows.append((int(fl[0]), int(fl[1]), int(fl[2]), int(fl[3])))
    return np.array(rows, dtype=B.DPOS) if rows else np.zeros(0, dtype=B.DPOS)


def low_mapq_lines(gf, sam_lines, scaffold, rows, n_scaffolds):
    """-> {'left': [...], 'right': [...]} or None when no MAPQ-0 record was seen (the reference then writes no file)."""
    idx = int(rows["mate_scaffold"][0]) if len(rows) else 0
    if hasattr(gf, "sam_pack"):      # SAM text parsed on the GPU; only this scaffold's name has an index, as below
        names = ["\x00%d" % i for i in range(idx)] + [scaffold]
        recs, cols = sam_io.decode_on_device(gf, sam_lines, names)
        if not (recs["mapq"] == 0).any():
            return None
    else:
        recs, cols = sam_io.decode(sam_lines, {scaffold: idx})
        if not any(int(f[4]) == 0 for f in cols):
            return None
    out = {"left": [], "right": []}
    if len(rows):
        for h in gf.tag_low_mapq(recs, rows):
            f = cols[h["rec"]]
            r = rows[h["gap"]]
            out["left" if int(f[1]) & 0x40 else "right"].append("%s %d_%d %d" % (f[0], r["src_scaffold"], r["src_gap"], int(f[4])))
    return out


def parse_discordant_reads_one_scaffold(working_folder, gf=None, n_scaffolds=None):
    open(working_folder + "cluster_by_discordant_reads_left.list", "w").close()
    open(working_folder + "cluster_by_discordant_reads_right.list", "w").close()
    lines = sys.stdin.read().splitlines()
    scaffolds = []
    for l in lines:
        f = l.split(None, 3)
        if len(f) > 2 and f[2] not in scaffolds:
            scaffolds.append(f[2])
    gf = gf or GapFill(int(os.environ.get("GF_DEVICE", "0")))
    for scf in scaffolds:
        sf = working_folder + "discordant_temp/" + scf + ".list"
        if not os.path.exists(sf):
            continue
        rows = read_rows(sf)
        gf.set_gaps(np.zeros(0, dtype=B.GAP), int(rows["mate_scaffold"].max()) + 1 if len(rows) else 1)
        res = low_mapq_lines(gf, [l for l in lines if l.split(None, 3)[2] == scf], scf, rows, None)
        if res is None:
            continue
        for side in ("left", "right"):
            with open(working_folder + "discordant_reads_list/%s_cluster_by_discordant_reads_%s.list" % (scf, side), "w") as f:
                f.write("".join(l + "\n" for l in res[side]))


if __name__ == "__main__":
    parse_discordant_reads_one_scaffold(sys.argv[1])
