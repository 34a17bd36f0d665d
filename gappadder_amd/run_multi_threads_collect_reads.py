"""Mirrors MultiThrdReadsCollector (run_multi_threads_collect_reads.py:9-38): one `samtools view BAM "scf"` per scaffold
that has gaps; instead of piping each into a CPython filter, all their records go through the GPU tagger.  With
samtools_path = "builtin" the BAM is read directly, once, inflated and decoded on the GPU (bam_io)."""
import subprocess

from . import bam_io
from . import sam_io
from .collect_reads_for_gaps import GapReadsCollector
from .hip_api import GapFill


def sam_of_scaffold(samtools_path, sf_bam, scaffold):
    """Text of `samtools view BAM "scaffold"` (run_multi_threads_collect_reads.py:30)."""
    return subprocess.run([samtools_path, "view", sf_bam, scaffold], check=True, stdout=subprocess.PIPE).stdout.decode()


class MultiThrdReadsCollector:
    def __init__(self, sf_fai, sf_bam, sf_gap_pos, anchor_mapq, gf=None):
        self.sf_fai = sf_fai
        self.sf_bam = sf_bam
        self.sf_gap_pos = sf_gap_pos
        self.anchor_mapq = anchor_mapq
        self._gf = gf

    def dispath_collect_jobs(self, nthreads, samtools_path, insert_size, derivation, clip_dist, working_folder):
        names = sam_io.read_fai(self.sf_fai)
        has_gap = set()
        with open(self.sf_gap_pos) as f:
            for line in f:
                fl = line.split()
                if len(fl) >= 4:
                    has_gap.add(fl[3])
        gf = self._gf or GapFill(0)
        grc = GapReadsCollector(insert_size, derivation, clip_dist, gf)
        open(working_folder + "cluster_by_gap_reads_left.list", "w").close()
        open(working_folder + "cluster_by_gap_reads_right.list", "w").close()
        def write(res):
            for s, d in res.items():
                for side in ("left", "right"):
                    with open(working_folder + "scaffold_reads_list_all/%s_cluster_by_gap_reads_%s.list" % (s, side), "w") as f:
                        f.write("".join(l + "\n" for l in d[side]))

        if bam_io.is_builtin(samtools_path):
            # one pass over the file: a scaffold's lists only depend on its own records, in file order, so the result equals
            # the per-scaffold `samtools view` loop below (records of scaffolds without gaps produce no hit and no file)
            gaps, _ = sam_io.read_gap_positions(self.sf_gap_pos, {n: i for i, n in enumerate(names)})
            gf.set_gaps(gaps, len(names))
            res = {}
            for recs, cols in bam_io.decode_file(gf, self.sf_bam, names):
                grc.tag_decoded(gf, recs, cols, gaps, names, self.anchor_mapq, res)
            write(res)
            return
        for scf in names:
            if scf not in has_gap:
                continue
            lines = sam_of_scaffold(samtools_path, self.sf_bam, scf).splitlines()
            write(grc.tag_lines(lines, self.sf_gap_pos, names, self.anchor_mapq))
