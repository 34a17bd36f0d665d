"""Mirrors DiscordantReadsCollector (run_multi_threads_discordant.py:12-594): discordant-region inversion, second hop,
FASTQ join into per-gap FASTQ files, high-quality subset.  Same file contracts; the second hop runs on the GPU."""
import os

import numpy as np

from . import _lib as B
from . import bam_io
from . import sam_io
from .collect_discordant_low_mapq_reads import low_mapq_lines, read_rows
from .hip_api import GapFill
from .run_multi_threads_collect_reads import sam_of_scaffold


class DiscordantReadsCollector:
    def __init__(self, sf_fai, sf_bam, working_folder, nthreads, gf=None, samtools_path="samtools"):
        self.sf_fai = sf_fai
        self.sf_bam = sf_bam
        self.working_folder = working_folder
        self.nthreads = nthreads
        self.samtools_path = samtools_path
        self._gf = gf

    # ---- run_multi_threads_discordant.py:19-122
    def collect_discordant_regions_v2(self, sf_out):
        names = sam_io.read_fai(self.sf_fai)
        sidx = {n: i for i, n in enumerate(names)}
        rows = []
        for scf in names:
            base = "%sscaffold_reads_list_all/%s_cluster_by_gap_reads_" % (self.working_folder, scf)
            if not (os.path.exists(base + "left.list") and os.path.exists(base + "right.list")):
                continue
            for side in ("left", "right"):
                with open(base + side + ".list") as f:
                    for rec in f:
                        fl = rec.split()
                        if len(fl) > 6 and fl[3] == "discordant":
                            mate = scf if fl[5] == "=" else fl[5]
                            rows.append((sidx[mate], int(fl[6]), sidx[scf], int(fl[1])))
        with open(sf_out, "w") as f:
            f.write("".join("%d %d %d %d\n" % r for r in rows))
        rows.sort()       # == sort -k1n -k2n -k3n -k4n (:103)
        with open(sf_out + ".sorted.txt", "w") as f:
            f.write("".join("%d %d %d %d\n" % r for r in rows))
        tmp = "%s/discordant_temp/" % self.working_folder
        by = {}
        for r in rows:
            by.setdefault(names[r[0]], []).append(r)
        for scf, rs in by.items():
            with open(tmp + scf + ".list", "w") as f:
                f.write("".join("%d %d %d %d\n" % r for r in rs))

    # ---- run_multi_threads_discordant.py:125-138 + collect_discordant_low_mapq_reads.py
    def dispath_collect_jobs(self):
        names = sam_io.read_fai(self.sf_fai)
        gf = self._gf or GapFill(0)
        open(self.working_folder + "cluster_by_discordant_reads_left.list", "w").close()
        open(self.working_folder + "cluster_by_discordant_reads_right.list", "w").close()
        def write(scf, res):
            for side in ("left", "right"):
                with open(self.working_folder + "discordant_reads_list/%s_cluster_by_discordant_reads_%s.list" % (scf, side), "w") as f:
                    f.write("".join(l + "\n" for l in res[side]))

        if bam_io.is_builtin(self.samtools_path):
            self._collect_from_bam(gf, names, write)
            return
        for scf in names:
            sf = self.working_folder + "discordant_temp/" + scf + ".list"
            if not os.path.exists(sf):
                continue
            rows = read_rows(sf)
            gf.set_gaps(np.zeros(0, dtype=B.GAP), len(names))
            lines = sam_of_scaffold(self.samtools_path, self.sf_bam, scf).splitlines()
            res = low_mapq_lines(gf, lines, scf, rows, len(names))
            if res is not None:
                write(scf, res)

    def _collect_from_bam(self, gf, names, write):
        """The per-scaffold loop above in one pass over the BAM (bam_io): the rows of every discordant_temp list form one
        table (rows carry their mate scaffold), a MAPQ-0 record is looked up in the rows of its own scaffold."""
        have = [i for i, n in enumerate(names) if os.path.exists(self.working_folder + "discordant_temp/" + n + ".list")]
        tables = [read_rows(self.working_folder + "discordant_temp/" + names[i] + ".list") for i in have]
        rows = np.concatenate(tables) if tables else np.zeros(0, dtype=B.DPOS)
        listed = set(have)
        gf.set_gaps(np.zeros(0, dtype=B.GAP), len(names))
        res = {}
        for recs, cols in bam_io.decode_file(gf, self.sf_bam, names):
            low = recs["mapq"] == 0
            for r in np.unique(recs["ref"][low]):          # the reference writes a scaffold's files once it saw a MAPQ-0 record
                if int(r) in listed:
                    res.setdefault(int(r), {"left": [], "right": []})
            if not len(rows):
                continue
            hits = gf.tag_low_mapq_bam(len(recs), rows) if cols.on_device else gf.tag_low_mapq(recs, rows)
            cols.prefetch(hits["rec"])
            for h in hits:
                f = cols[h["rec"]]
                row = rows[h["gap"]]
                res[int(row["mate_scaffold"])]["left" if int(f[1]) & 0x40 else "right"].append(
                    "%s %d_%d %d" % (f[0], row["src_scaffold"], row["src_gap"], int(f[4])))
        for i in sorted(res):
            write(names[i], res[i])

    # ---- FASTQ join (run_multi_threads_discordant.py:141-317 / 452-594)
    def _read_gap_map(self, names, side, high_quality):
        sidx = {n: i for i, n in enumerate(names)}
        m = {}
        if not high_quality:
            for scf in names:
                p = self.working_folder + "discordant_reads_list/%s_cluster_by_discordant_reads_%s.list" % (scf, side)
                if os.path.exists(p):
                    with open(p) as f:
                        for line in f:
                            fl = line.split()
                            m.setdefault(fl[0], {})[fl[1]] = 1
        for scf in names:
            p = self.working_folder + "scaffold_reads_list_all/%s_cluster_by_gap_reads_%s.list" % (scf, side)
            if os.path.exists(p):
                with open(p) as f:
                    for line in f:
                        fl = line.split()
                        if high_quality and int(fl[2]) != 60:
                            continue
                        m.setdefault(fl[0], {})["%d_%s" % (sidx[scf], fl[1])] = 1
        return m

    def _dispatch(self, sf_raw, m, suffix, folder):
        out = {}
        with open(sf_raw) as f:
            while True:
                h = f.readline()
                if not h:
                    break
                s, _, q = f.readline(), f.readline(), f.readline()
                rid = h.split()[0].split("/")[0][1:].rstrip()   # :212-214
                gaps = m.get(rid)
                if gaps:
                    rec = "@%s%s\n%s\n+\n%s\n" % (rid, suffix, s.rstrip(), q.rstrip())
                    for key in gaps:
                        out.setdefault(key, []).append(rec)
        for key, recs in out.items():
            with open("%s%s/%s.fastq" % (self.working_folder, folder, key), "a") as f:
                f.write("".join(recs))

    def _join(self, sf_raw_left, sf_raw_right, folder, high_quality, extra=None):
        names = sam_io.read_fai(self.sf_fai)
        d = self.working_folder + folder
        os.makedirs(d, exist_ok=True)
        for fn in os.listdir(d):          # the reference wipes the folder with rsync --delete (:199)
            os.remove(os.path.join(d, fn))
        left = self._read_gap_map(names, "left", high_quality)
        if extra and not high_quality:
            for rid, ks in extra[0].items():
                left.setdefault(rid, {}).update((kk, 1) for kk in ks)
        if not high_quality:
            with open(self.working_folder + "left_reads.list", "w") as f:
                f.write("".join("%s %s\n" % (g, r) for r, gs in left.items() for g in gs))
        self._dispatch(sf_raw_left, left, "_1", folder)
        right = self._read_gap_map(names, "right", high_quality)
        if extra and not high_quality:
            for rid, ks in extra[1].items():
                right.setdefault(rid, {}).update((kk, 1) for kk in ks)
        if not high_quality:
            with open(self.working_folder + "right_reads.list", "w") as f:
                f.write("".join("%s %s\n" % (g, r) for r, gs in right.items() for g in gs))
        self._dispatch(sf_raw_right, right, "_2", folder)

    def merge_dispatch_reads_for_gaps_v2(self, sf_raw_left, sf_raw_right, extra=None):
        """extra: optional ({readId: set(gapKey)}, same for the right file) from the flank-k-mer screen (kmer_recruit.py)."""
        self._join(sf_raw_left, sf_raw_right, "gap_reads", False, extra)

    def dispatch_high_quality_reads_for_gaps(self, sf_raw_left, sf_raw_right):
        self._join(sf_raw_left, sf_raw_right, "gap_reads_high_quality", True)
