"""Drop-in twin of the reference's collect_reads_for_gaps.py (one scaffold's SAM text on stdin -> the two list files),
with the per-record arithmetic on the GPU (gf_tag_alignments).

    samtools view BAM "scf" | python -m gappadder_amd.collect_reads_for_gaps gap_pos anchor_mapq wf IS sd clip_dist -
"""
import os
import sys

from . import _lib as B
from . import sam_io
from .hip_api import GapFill


class GapReadsCollector:
    def __init__(self, insert_size, derivation, dist_clip, gf=None):
        self.insert_size = insert_size
        self.derivation = derivation
        self.dist1 = insert_size - 3 * derivation
        self.dist2 = insert_size + 3 * derivation
        self.dist_clip = dist_clip
        self._gf = gf

    is_clipped = staticmethod(sam_io.clip_flag)

    def tag_lines(self, sam_lines, sf_gap_pos, sf_fai_or_names, anchor_mapq):
        """-> {scaffold: {'left': [line], 'right': [line]}} for the scaffolds seen in sam_lines that have gaps."""
        names = sf_fai_or_names if isinstance(sf_fai_or_names, list) else sam_io.read_fai(sf_fai_or_names)
        sidx = {n: i for i, n in enumerate(names)}
        gaps, _ = sam_io.read_gap_positions(sf_gap_pos, sidx)
        gf = self._gf or GapFill(int(os.environ.get("GF_DEVICE", "0")))
        gf.set_gaps(gaps, len(names))
        on_device = hasattr(gf, "sam_pack")     # SAM text parsed on the GPU (gf_sam_pack); columns are cut lazily for the hits
        recs, cols = sam_io.decode_on_device(gf, sam_lines, names) if on_device else sam_io.decode(sam_lines, sidx)
        return self.tag_decoded(gf, recs, cols, gaps, names, anchor_mapq, {})

    def tag_decoded(self, gf, recs, cols, gaps, names, anchor_mapq, out):
        """Tag one batch of decoded records (gf.set_gaps done); lines are appended to `out` in record order, so successive
        batches of one coordinate-sorted file (bam_io.decode_file) give the lists of one pass over it."""
        import numpy as np
        if getattr(cols, "on_device", False):   # builtin BAM mode: the records are on the GPU already
            hits = gf.tag_alignments_bam(len(recs), self.insert_size, self.derivation, self.dist_clip, anchor_mapq)
        else:
            hits = gf.tag_alignments(recs, self.insert_size, self.derivation, self.dist_clip, anchor_mapq)
        with_gaps = set(int(g) for g in gaps["scaffold"])
        # the reference opens the pair of files at the first record of a scaffold (:93-102)
        _, first = np.unique(recs["ref"], return_index=True)
        for i in sorted(first):
            r = int(recs["ref"][i])
            if r < len(names) and r in with_gaps and names[r] not in out:
                out[names[r]] = {"left": [], "right": []}
        if hasattr(cols, "prefetch"):
            cols.prefetch(hits["rec"])
        for h in hits:
            f = cols[h["rec"]]
            g = gaps[h["gap"]]
            own = "left" if int(f[1]) & 0x40 else "right"
            side = ("right" if own == "left" else "left") if h["to_mate"] else own
            if h["kind"] == B.KIND_DISCORDANT:
                line = "%s %d %s discordant %s %s %s %d" % (f[0], g["idx_in_scaffold"], f[4], f[3], f[6], f[7],
                                                          int(g["end"]) - int(g["start"]))
            else:
                line = "%s %d %s %s" % (f[0], g["idx_in_scaffold"], f[4], B.KIND_NAMES[int(h["kind"])])
            out[f[2]][side].append(line)
        return out

    def _run(self, sf_gap_pos, anchor_mapq, working_folder, names):
        open(working_folder + "cluster_by_gap_reads_left.list", "w").close()    # the reference's dummies (:70-71)
        open(working_folder + "cluster_by_gap_reads_right.list", "w").close()
        res = self.tag_lines(sys.stdin, sf_gap_pos, names, anchor_mapq)
        for scf, d in res.items():
            for side in ("left", "right"):
                with open(working_folder + "scaffold_reads_list_all/%s_cluster_by_gap_reads_%s.list" % (scf, side), "w") as f:
                    f.write("".join(l + "\n" for l in d[side]))

    # the two entry points of the reference differ only in the |TLEN| <= dist1 test, selected by IS >= 750 (:275);
    # the GPU kernel takes that switch from insert_size
    def parse_reads_fall_in_gaps_one_scaffold(self, sf_gap_pos, anchor_mapq, working_folder, names=None):
        self._run(sf_gap_pos, anchor_mapq, working_folder, names or _names_from_gap_pos(sf_gap_pos))

    parse_reads_fall_in_gaps_one_scaffold_short_is = parse_reads_fall_in_gaps_one_scaffold


def _names_from_gap_pos(sf_gap_pos):
    """Stand-alone CLI use has no .fai: scaffold order = order of first appearance in gap_positions.txt (indices are
    only used to group gaps here, the list lines carry names)."""
    names = []
    with open(sf_gap_pos) as f:
        for line in f:
            fl = line.split()
            if len(fl) >= 4 and fl[3] not in names:
                names.append(fl[3])
    return names


if __name__ == "__main__":
    sf_gap_pos, anchor_mapq, working_folder = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    insert_size, derivation, dist_clip = int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    GapReadsCollector(insert_size, derivation, dist_clip).parse_reads_fall_in_gaps_one_scaffold(sf_gap_pos, anchor_mapq, working_folder)
