"""CLI with the reference's interface (main.py:26-275): python -m gappadder_amd.main -c {Clean,All,Preprocess,Collect,Assembly}
-g config.json, same JSON keys, same working-folder layout.  Collect and the first assembly round run on the GPU."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import time

from . import assemble_gaps
from . import bam_io
from .gnrt_pos_true_seqs import DGProcessor
from .hip_api import GapFill
from .merge_reads import ReadsMerger
from .run_multi_threads_collect_reads import MultiThrdReadsCollector
from .run_multi_threads_discordant import DiscordantReadsCollector

MERGE_FOLDER = "merged/"
SUB = ("scaffold_reads_list_all", "gap_reads", "gap_reads_for_alignment", "gap_reads_high_quality", "discordant_reads_list",
       "discordant_temp")
SUB_MERGED = ("gap_reads", "gap_reads_for_alignment", "gap_reads_high_quality", "kmc_temp", "temp", "kmers", "velvet_temp",
              "both_unmapped", "unmapped_reads")


def parse_configuration(path):
    with open(path) as f:
        data = json.load(f)
    for key in ("draft_genome", "alignments", "raw_reads"):
        if key not in data:
            raise SystemExit("configuration lacks '%s'" % key)
    cfg = {"draft": data["draft_genome"]["fa"],
           "alignments": [(r["bam"], int(r["is"]), int(r["std"])) for r in data["alignments"]],
           "raw_reads": [(r["left"], r["right"]) for r in data["raw_reads"]],
           "kmers": [(int(r["k"]), int(s["k"])) for r in data.get("kmer_length", []) for s in r["k_velvet"]]}
    p = data.get("parameters", {})
    cfg["min_gap"] = int(p.get("min_gap_size", 100))
    cfg["flank"] = int(p.get("flank_length", 300))
    cfg["nthreads"] = int(p.get("nthreads", 15))
    cfg["wf"] = p.get("working_folder", "./GAPPadder_Output/")
    cfg["samtools"] = data.get("software_path", {}).get("samtools", "samtools")
    cfg["kmer_screen"] = int(p.get("kmer_screen", 0))   # extension: flank-k-mer recruitment from the FASTQ files (0 = off)
    for path_, what in [(cfg["draft"], "draft genome")] + [(a[0], "bam") for a in cfg["alignments"]] + \
                       [(x, "raw reads") for pair in cfg["raw_reads"] for x in pair] + [(cfg["wf"], "working folder")]:
        if not os.path.exists(path_):
            raise SystemExit("The provided %s %s does not exist, please check!!!" % (what, path_))
    if not cfg["wf"].endswith("/"):
        cfg["wf"] += "/"
    return cfg


def prepare_folders(alignments, wf):
    folders = []
    for n, (_, is_, _) in enumerate(alignments, 1):
        p = "%s%d_is%d/" % (wf, n, is_)
        folders.append(p)
        for s in SUB:
            os.makedirs(p + s, exist_ok=True)
    for s in SUB_MERGED:
        os.makedirs(wf + MERGE_FOLDER + s, exist_ok=True)
    return folders


def collect_per_scaffold(cfg, gf, sf_fai, sf_gap_pos, folders, anchor_mapq, clip_dist, wf):
    """The reference's Collect stage call for call (main.py:226-270): a `samtools view` pipe per scaffold (or one decode pass over the
    BAM in builtin mode), list files, then the FASTQ join on the host."""
    for (bam, is_, sd), folder, (left, right) in zip(cfg["alignments"], folders, cfg["raw_reads"]):
        MultiThrdReadsCollector(sf_fai, bam, sf_gap_pos, anchor_mapq, gf).dispath_collect_jobs(
            cfg["nthreads"], cfg["samtools"], is_, sd, clip_dist, folder)
        drc = DiscordantReadsCollector(sf_fai, bam, folder, cfg["nthreads"], gf, cfg["samtools"])
        drc.collect_discordant_regions_v2(folder + "discordant_reads_pos.txt")
        drc.dispath_collect_jobs()
        extra = None
        if cfg["kmer_screen"]:
            from .kmer_recruit import screen_fastq_pair
            extra = screen_fastq_pair(gf, sf_fai, sf_gap_pos, wf, left, right, cfg["kmer_screen"])
        drc.merge_dispatch_reads_for_gaps_v2(left, right, extra)
        drc.dispatch_high_quality_reads_for_gaps(left, right)
    rm = ReadsMerger()
    for name in ("gap_reads", "gap_reads_alignment", "gap_reads_high_quality"):
        rm.merge_reads_v2(sf_fai, sf_gap_pos, folders, name, wf + MERGE_FOLDER, cfg["nthreads"])


def main_func(command, sf_config):
    cfg = parse_configuration(sf_config)
    wf = cfg["wf"]
    sf_fai = cfg["draft"] + ".fai"
    if not os.path.exists(sf_fai):
        if bam_io.is_builtin(cfg["samtools"]):
            bam_io.write_fai(cfg["draft"])
        else:
            subprocess.call([cfg["samtools"], "faidx", cfg["draft"]])
    sf_gap_pos = wf + "gap_positions.txt"
    anchor_mapq, clip_dist = 30, 250      # main.py:215-216
    if command in ("Clean", "All"):
        for fn in os.listdir(wf):
            p = os.path.join(wf, fn)
            shutil.rmtree(p) if os.path.isdir(p) else os.remove(p)
    timings = {"stages_s": {}}      # GF_TIMINGS=<file>: wall time per stage (+ the device Collect's own split) as JSON
    if command in ("Preprocess", "All"):
        t0 = time.perf_counter()
        dgp = DGProcessor(cfg["draft"], sf_gap_pos)
        dgp.gnrt_gap_positions(cfg["min_gap"])
        dgp.get_gap_flank_seqs(cfg["draft"], sf_gap_pos, cfg["flank"], sf_fai, wf)
        timings["stages_s"]["preprocess"] = time.perf_counter() - t0
    gf = GapFill(int(os.environ.get("GF_DEVICE", "0"))) if command in ("Collect", "Assembly", "All") else None
    first_round = None
    if command in ("Collect", "All"):
        t0 = time.perf_counter()
        folders = prepare_folders(cfg["alignments"], wf)
        if len(cfg["alignments"]) != len(cfg["raw_reads"]):
            raise SystemExit("# of alignment files and # of raw reads do not match!!!!!")
        # FASTQ(.gz): gzip / BGZF read files are inflated once into {wf}tmp_fastq/; every path below reads plain text
        from .fastq_io import plain_fastq
        cfg["raw_reads"] = [(plain_fastq(l, wf + "tmp_fastq"), plain_fastq(r, wf + "tmp_fastq")) for l, r in cfg["raw_reads"]]
        done = False
        if bam_io.is_builtin(cfg["samtools"]) and os.environ.get("GF_DEVICE_COLLECT", "1") != "0":
            # libraries resident in HBM, one pass over every file, the pipeline bench.py times (device_collect.py); with `-c All` the
            # first assembly round runs on the pools where they are
            from .device_collect import DeviceCollector, DeviceCollectUnsupported
            try:
                dc = DeviceCollector(gf, cfg, sf_fai, sf_gap_pos, anchor_mapq, clip_dist, kmers=cfg["kmers"] if command == "All" else None)
                res = dc.run(folders, wf + MERGE_FOLDER)
                done = True
                if command == "All" and res.k_pairs:
                    first_round = res
                timings["seconds"] = dc.t
                timings["libraries"] = [dict(lb.counts, reads=lb.n_reads, records=lb.n_recs, records_without_a_read=lb.records_without_a_read) for lb in dc.libs]
                timings["gaps"] = len(res.keys)
                timings["read_len"] = res.read_len        # the packed row length: the longest read of all libraries
                timings["gaps_closed_on_device"] = int(res.n_closed)
            except DeviceCollectUnsupported as e:
                sys.stderr.write("device-resident Collect not used (%s): per-scaffold path\n" % e)
            except (RuntimeError, MemoryError) as e:
                # out of device memory (torch's allocator or the library's GF_E_NOMEM) or a capacity of the one-shot step outgrown
                # (Pipeline.fetch's "step overflow"): inputs the streaming per-scaffold path handles chunk by chunk must not crash the
                # CLI — free the libraries, put the context back to its defaults, take that path.  Anything else is a real error.
                import torch
                from ._lib import GF_E_NOMEM, GF_E_NOSPACE, GapFillError
                oom = isinstance(e, (torch.cuda.OutOfMemoryError, MemoryError)) or "out of memory" in str(e).lower() or "step overflow" in str(e) \
                    or "keep overflowing" in str(e) or (isinstance(e, GapFillError) and e.code in (GF_E_NOMEM, GF_E_NOSPACE))
                if not oom:
                    raise
                sys.stderr.write("device-resident Collect gave up (%s): per-scaffold path\n" % (str(e).splitlines()[0][:200],))
                dc = res = first_round = None
                import gc
                gc.collect()
                torch.cuda.empty_cache()
                gf.set_option("asm_max_pool_reads", 0)
                gf.set_option("asm_big_pool_reads", 131072)
                for key in ("seconds", "libraries", "gaps", "read_len", "gaps_closed_on_device"):
                    timings.pop(key, None)
        if not done:
            collect_per_scaffold(cfg, gf, sf_fai, sf_gap_pos, folders, anchor_mapq, clip_dist, wf)
        timings["stages_s"]["collect" + ("_and_first_assembly_round" if first_round is not None else "")] = time.perf_counter() - t0
    if command in ("Assembly", "All"):
        t0 = time.perf_counter()
        for s in SUB_MERGED:
            os.makedirs(wf + MERGE_FOLDER + s, exist_ok=True)
        ga = assemble_gaps.GapAssembler(sf_fai, sf_gap_pos, cfg["nthreads"], wf + MERGE_FOLDER, cfg["kmers"], gf,
                                        bam_list=[bam for bam, _, _ in cfg["alignments"]], samtools_path=cfg["samtools"])
        if first_round is not None:
            assemble_gaps.set_first_round(first_round)
        res = ga.assemble_pipeline()
        print("assembled %d gaps, %d closed (picked_seqs.fa), %d with an extended (partial) fill, %d gaps got both-unmapped pairs in the "
              "second round, contigs merged in %d gap rounds (%d bridging high-quality reads); contigs in %svelvet_temp/*/contigs.fa"
              % (res["gaps"], res["closed"], res.get("extended", 0), res["second_round_gaps"], res["gaps_with_merged_contigs"],
                 res["bridging_reads"], wf + MERGE_FOLDER))
        timings["stages_s"]["assembly_rounds"] = time.perf_counter() - t0
        timings["assembly"] = res
    if os.environ.get("GF_TIMINGS"):
        sys.stderr.write("stages: " + ", ".join("%s %.3f s" % kv for kv in timings["stages_s"].items()) +
                         ("; device collect: " + ", ".join("%s %.3f s" % kv for kv in timings["seconds"].items()) if "seconds" in timings else "") + "\n")
        if os.environ["GF_TIMINGS"] not in ("1", ""):
            with open(os.environ["GF_TIMINGS"], "w") as f:
                json.dump(timings, f)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Run the GAPPadder recruit + local-assembly hot path on MI355X")
    ap.add_argument("-g", "--config", type=str, required=True, help="Configuration file name")
    ap.add_argument("-c", "--command", type=str, required=True, help="Clean | All | Preprocess | Collect | Assembly")
    a = ap.parse_args(argv)
    main_func(a.command, a.config)


if __name__ == "__main__":
    main()
