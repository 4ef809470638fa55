import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import problems
import dedalus_amd.public as d3
from oracle.np_executor import NumpyExecutor
s1,f1=problems.rayleigh_benard_2d(d3,Nx=32,Nz=16)
s2,f2=problems.rayleigh_benard_2d(d3,Nx=32,Nz=16,dist_kw=dict(executor=NumpyExecutor()))
def cmp(name,a,b):
    a=s1.ex.download(a); b=np.asarray(b)
    print("%-10s finite=%s  rel=%.3e  norm=%.4e"%(name, np.isfinite(a).all(), np.linalg.norm(a-b.reshape(a.shape))/max(np.linalg.norm(b),1e-300), np.linalg.norm(b)))
    return a
for s in (s1,s2): s.sync_state_to_device()
cmp("X0",s1.X,s2.X)
ts1,ts2=s1.timestepper,s2.timestepper
for s,ts in ((s1,ts1),(s2,ts2)):
    s.pack.matvec(s.M_id,s.X,ts.MX0); s.pack.matvec(s.L_id,s.X,ts.LX[0]); s.evaluate_F(ts.F[0])
cmp("MX0",ts1.MX0,ts2.MX0); cmp("LX0",ts1.LX[0],ts2.LX[0]); cmp("F0",ts1.F[0],ts2.F[0])
# inspect pieces of F evaluation
for (l1,r1,n1),(l2,r2,n2) in zip(s1.nl_leaves,s2.nl_leaves):
    ev1,ev2=s1.evaluator_core,s2.evaluator_core
    ev1.new_pass(); ev2.new_pass()
    a,b=l1.args; a2,b2=l2.args
    cmp(" ga",ev1.eval_grid(a),ev2.eval_grid(a2)); cmp(" cb",ev1.eval_coeff(b),ev2.eval_coeff(b2)); cmp(" gb",ev1.eval_grid(b),ev2.eval_grid(b2)); cmp(" g",ev1.eval_grid(l1),ev2.eval_grid(l2))
k=1e-3; g=ts1.H[1,1]
lu1=s1.factor(1.0,k*g); lu2=s2.factor(1.0,k*g)
print("flagged", s1.pack.lu_meta[lu1])
rhs=np.random.default_rng(0).standard_normal((s1.R,s1.nx,s1.ny)); rhs[:,1,:]=0
r1=s1.ex.from_host(rhs); x1=s1.ex.empty(rhs.shape); x2=np.zeros_like(rhs)
s1.pack.solve(lu1,r1,x1); s2.pack.solve(lu2,rhs,x2)
xa=cmp("solve",x1,x2)
bad=np.argwhere(~np.isfinite(xa)); print("nonfinite entries", len(bad), bad[:10])
