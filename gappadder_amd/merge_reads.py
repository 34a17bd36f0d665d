"""Mirrors ReadsMerger.merge_reads_v2 (merge_reads.py:12-56): per gap id, concatenate the per-library FASTQ files in
library order into merged/{reads_folder}/{id}.fastq."""
import os

from . import sam_io


class ReadsMerger:
    def merge_reads_v2(self, sf_fai, sf_gap_pos, merge_folder_list, reads_folder, working_folder, n):
        d = "%s%s" % (working_folder, reads_folder)
        os.makedirs(d, exist_ok=True)
        sidx = {nm: i for i, nm in enumerate(sam_io.read_fai(sf_fai))}
        _, keys = sam_io.read_gap_positions(sf_gap_pos, sidx)
        for key in keys:
            parts = [os.path.join(folder, reads_folder, key + ".fastq") for folder in merge_folder_list]
            parts = [p for p in parts if os.path.exists(p)]
            if not parts:
                continue
            with open("%s/%s.fastq" % (d, key), "wb") as out:
                for p in parts:
                    with open(p, "rb") as f:
                        out.write(f.read())
