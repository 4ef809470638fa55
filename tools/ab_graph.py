import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import problems, dedalus_amd.public as d3
g = sys.argv[1] == "1"
s, f = problems.rayleigh_benard_3d(d3, Nx=64, Ny=64, Nz=32)
if g: s.enable_step_graph(True)
for _ in range(10): s.step(1e-3)
for rep in range(3):
    s.ex.sync(); t0 = time.time()
    for _ in range(200): s.step(1e-3)
    s.ex.sync(); el = time.time() - t0
    print("graph" if g else "plain", "%.1f steps/s" % (200 / el))
