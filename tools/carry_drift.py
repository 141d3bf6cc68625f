"""Drift of the residual carried from step to step in fixed-dt runs against the refresh period (fv_tune key 7): final state
after 1 500 one-iteration steps at 216^3 against the run that recomputes b - A u every step.  usage: python tools/carry_drift.py"""
import sys, numpy as np, time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from __graft_entry__ import load_package
import bench
fv = load_package(); lib = fv.load()
ns=[216]*3
mins,maxs=bench.spacing_box(ns); dn,src=bench.box_setup(ns)
res={}
for refresh in (0, 32, 128, 512, 4096):
    lib.fv_tune(7, refresh)
    p=fv.Problem.regulargrid(mins,maxs,ns,dn); p.assemble(np.array([1e-5]),src,np.full(len(dn),1e3))
    st=p.transient_begin(0.1,None,np.full(p.N,1e3))
    t0=time.perf_counter(); it,info,_=p.run_fixed(st,60.0,1500,1e-10); p.ctx.synchronize(); sec=time.perf_counter()-t0
    u=st.free_values()
    # true residual of the last step's system is not available directly; compare states
    res[refresh]=(u, it.sum(), sec, info.relres)
    p.close()
lib.fv_tune(7, 128)
base=res[0][0]
for k,(u,its,sec,rr) in res.items():
    print("refresh %5d: iters %d, %.3f s, last relres %.2e, max |u - u_fresh| / |drawdown| = %.3e, rel to heads %.3e" % (k, its, sec, rr, np.abs(u-base).max()/np.abs(1e3-base).max(), np.abs(u-base).max()/1e3))
