"""cProfile of the CLI (`-c All`, builtin BAM) on synthetic files: usage (on the GPU box): python3 tools/prof_e2e.py [C3|C2G|C2] [out prefix]
C3: 200 gaps, 5 M reads;  C2G: C2's 1 000 gaps in 50 scaffolds with a tenth of its reads (what the host-side rounds scale with)."""
import cProfile, pstats, sys, os, io, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import synth_files_util as SF
case = sys.argv[1] if len(sys.argv) > 1 else "C3"
prefix = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "e2e_profile_" + case)
root = tempfile.mkdtemp(prefix="gf_prof_")
if case == "C2":
    cfgp, wf = SF.write_case(root, 20260002, 5_000_000, 50, 20, 2000, [(300, 30, 25_000_000)], [(31, 29)], kmer_screen=31, nthreads=8)
elif case == "C2G":
    cfgp, wf = SF.write_case(root, 20260002, 5_000_000, 50, 20, 2000, [(300, 30, 2_500_000)], [(31, 29)], kmer_screen=31, nthreads=8)
else:
    cfgp, wf = SF.write_case(root, 20260003, 4_600_000, 1, 200, 1000, [(300, 30, 2_500_000)], [(41, 39)], kmer_screen=41, nthreads=8)
from gappadder_amd import main as M
pr = cProfile.Profile()
pr.enable()
M.main(["-c", "All", "-g", cfgp])
pr.disable()
import shutil
shutil.rmtree(root, ignore_errors=True)
for order in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(order).print_stats(60)
    open("%s_%s.txt" % (prefix, order), "w").write(s.getvalue())
