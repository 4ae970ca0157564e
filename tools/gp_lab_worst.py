"""CPU: which gradient component carries the error of the worst draws of tools/gp_host_lab.py (it was d/d(dc) alone\nbefore the phase-flux form)."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
pass
import gp_host_lab as L
from oracle import c_port as C
import test_gp_host as H
lib = L.build("worst", [])
np.set_printoptions(linewidth=220, precision=3)
out=[]
for t, y, diag, cr, cc, dtm in L.cases(1, 60):
    D=y.shape[0]
    Cn = 0
    ll, flags, Cu, g = H.run(lib, t, y, diag, cr, cc, gll=np.ones(D), n_chunks=Cn)
    for d in range(D):
        co=(cr[d,:,0],cr[d,:,1],cc[d,:,0],cc[d,:,1],cc[d,:,2],cc[d,:,3])
        wl,wg=C.celerite(t,y[d],diag[d],co,grad=True)
        if not np.isfinite(wl): continue
        errs={}
        for nm,got,want in (("y",g["y"][d],wg["y"]),("diag",g["diag"][d],wg["diag"]),("ar",g["real"][d,:,0],wg["ar"]),("cr",g["real"][d,:,1],wg["cr"]),("ac",g["cplx"][d,:,0],wg["ac"]),("bc",g["cplx"][d,:,1],wg["bc"]),("cc",g["cplx"][d,:,2],wg["cc"]),("dc",g["cplx"][d,:,3],wg["dc"])):
            if want.size: errs[nm]=np.abs(got-want).max()/(np.abs(want).max()+1e-300)
        e=max(errs.values())
        ba2=((cc[d,:,1]/cc[d,:,0])**2).max(); snr=(cr[d,:,0].sum()+cc[d,:,0].sum())/diag[d].min()
        out.append((e,(1+ba2)*snr,errs,cr[d]*[1,dtm],cc[d]*[1,1,dtm,dtm],np.diff(t).max()/dtm, abs(ll[d]-wl)/abs(wl)))
out.sort(key=lambda r:-r[0])
for r in [q for q in out if q[1] < 3e6][:7]:
    if True:
        print("err %.1e kappa %.1e llerr %.1e gap %.0f"%(r[0],r[1],r[6],r[5])); print("  errs",{k:"%.0e"%v for k,v in r[2].items()}); print("  real (a, c dt):",r[3].tolist()); print("  cplx (a,b,c dt,d dt):",r[4].tolist())
