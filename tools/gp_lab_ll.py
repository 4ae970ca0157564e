"""CPU: log-likelihoods of the robust route (serial chain / two Newton iterations; host-compiled lanes) and of the sequential
recurrences (C port) against the long-double dense definition, draws flagged kFlagRobust, by conditioning score."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, R + '/tests', R + '/tools']
import numpy as np
import gp_host_lab as L
from oracle import c_port as C
from oracle.make_golden_r02 import gp_dense_ld
import test_gp_host as H
import subprocess
out = R+"/tests/_build/lab_ll.so"
subprocess.run(["g++","-O2","-std=c++17","-shared","-fPIC","-o",out,R+"/tests/gp_host_harness.cpp"],check=True)
lib = ctypes.CDLL(out); lib.harness_gp_state_doubles.restype = ctypes.c_int64
rows=[]
os.environ["LAB_NMAX"]="700"
for ci,(t,y,diag,cr,cc,dtm) in enumerate(L.cases(5, 60, nmax=700)):
    D=y.shape[0]
    res={}
    for mode in (-1, 2):
        lib.harness_set_newton(mode, 0)
        res[mode]=H.run(lib,t,y,diag,cr,cc,gll=np.ones(D))
    flags=res[-1][1]
    for d in np.nonzero(flags==1)[0]:
        co=(cr[d,:,0],cr[d,:,1],cc[d,:,0],cc[d,:,1],cc[d,:,2],cc[d,:,3])
        kap=(1+((cc[d,:,1]/cc[d,:,0])**2).max())*(cr[d,:,0].sum()+cc[d,:,0].sum())/diag[d].min()
        if kap>1e8: continue
        wl=C.celerite(t,y[d],diag[d],co,grad=False)
        ll,_=gp_dense_ld(t,y[d],diag[d],co)
        ll=float(ll)
        rows.append((kap, abs(res[-1][0][d]-ll)/abs(ll), abs(res[2][0][d]-ll)/abs(ll), abs(wl-ll)/abs(ll)))
rows=np.array(rows)
print("robust draws", len(rows))
for lo in (4,5,6,7):
    m=(rows[:,0]>=10.0**lo)&(rows[:,0]<10.0**(lo+1))
    if m.any(): print("1e%d (%d): ll rel err vs long double: chain %.1e  newton2 %.1e  C port (sequential) %.1e"%(lo,m.sum(),rows[m,1].max(),rows[m,2].max(),rows[m,3].max()))
