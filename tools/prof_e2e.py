import cProfile, pstats, sys, os, io, tempfile
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import synth_files_util as SF
root = tempfile.mkdtemp(prefix="gf_prof_")
cfgp, wf = SF.write_case(root, 20260003, 4_600_000, 1, 200, 1000, [(300, 30, 2_500_000)], [(41, 39)], nthreads=8)
from gappadder_amd import main as M
pr = cProfile.Profile()
pr.enable()
M.main(["-c", "All", "-g", cfgp])
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(45)
open("/root/repo/gpurun_out/r05_e2e_profile.txt", "w").write(s.getvalue())
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(30)
open("/root/repo/gpurun_out/r05_e2e_profile_tottime.txt", "w").write(s.getvalue())
