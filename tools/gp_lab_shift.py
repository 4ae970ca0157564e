"""CPU (tools/gp_host_lab.py): is d loglike / d(oscillation rate) of the lane pipeline independent of the ORIGIN of the time axis?\nThe series of ten random batches shifted by 0, 10 and 100 spans, against the C port at the unshifted stamps (and the C port's own\nsensitivity to the shift).  Before the phase-flux form (exo_celerite_core.hpp) the error grew with the shift: 7e-5 -> 7e-3."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
os.environ.setdefault("LAB_NMIN","4000"); os.environ.setdefault("LAB_NMAX","8000")
import gp_host_lab as L
from oracle import c_port as C
import test_gp_host as H
lib = L.build("worst", [])
k=0
for t, y, diag, cr, cc, dtm in L.cases(2, 40):
    D=y.shape[0]; Cn=max(2,t.size//128)
    res=[]
    for T0 in (0.0, 10*(t[-1]-t[0]), 100*(t[-1]-t[0])):
        ll, flags, Cu, g = H.run(lib, t+T0, y, diag, cr, cc, gll=np.ones(D), n_chunks=Cn)
        res.append(g["cplx"][:, :, 3].copy())
    d=0
    co=(cr[d,:,0],cr[d,:,1],cc[d,:,0],cc[d,:,1],cc[d,:,2],cc[d,:,3])
    wl,wg=C.celerite(t,y[d],diag[d],co,grad=True)
    wl2,wg2=C.celerite(t+100*(t[-1]-t[0]),y[d],diag[d],co,grad=True)
    sc=np.abs(wg["dc"]).max()
    print("case",k,"J",cr.shape[1]+2*cc.shape[1],"chunked gd err T0=0: %.1e  T0=10 span: %.1e  T0=100 span: %.1e | sequential T0=100 span: %.1e"%(
        np.abs(res[0][d]-wg["dc"]).max()/sc, np.abs(res[1][d]-wg["dc"]).max()/sc, np.abs(res[2][d]-wg["dc"]).max()/sc, np.abs(wg2["dc"]-wg["dc"]).max()/sc))
    k+=1
    if k>=10: break
