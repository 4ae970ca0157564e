"""CPU: who is right in the conditioning tail -- the sequential recurrences (C port) or the time-parallel lane pipeline?\nThe worst draws of tools/gp_host_lab.py at kappa 1e6 .. 1e8 against the long-double dense definition: the C port holds\n1e-10, the lane pipeline does not (the boundary states of the scans), polished or not."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
import gp_host_lab as L
from oracle import c_port as C
from oracle.make_golden_r02 import gp_dense_ld
import test_gp_host as H
lib = L.build("truth", [])
def errs(g, wg, d):
    e={}
    for nm,got,want in (("y",g["y"][d],wg["y"]),("diag",g["diag"][d],wg["diag"]),("ar",g["real"][d,:,0],wg["ar"]),("cr",g["real"][d,:,1],wg["cr"]),("ac",g["cplx"][d,:,0],wg["ac"]),("bc",g["cplx"][d,:,1],wg["bc"]),("cc",g["cplx"][d,:,2],wg["cc"]),("dc",g["cplx"][d,:,3],wg["dc"])):
        if want.size: e[nm]=np.abs(got-want).max()/(np.abs(want).max()+1e-300)
    return e
cands=[]
for ci,(t, y, diag, cr, cc, dtm) in enumerate(L.cases(1, 60)):
    D=y.shape[0]
    res={}
    for pol in (0,4):
        lib.harness_set_polish(pol)
        res[pol]=H.run(lib, t, y, diag, cr, cc, gll=np.ones(D), n_chunks=0)
    for d in range(D):
        co=(cr[d,:,0],cr[d,:,1],cc[d,:,0],cc[d,:,1],cc[d,:,2],cc[d,:,3])
        wl,wg=C.celerite(t,y[d],diag[d],co,grad=True)
        if not np.isfinite(wl): continue
        ba2=((cc[d,:,1]/cc[d,:,0])**2).max(); snr=(cr[d,:,0].sum()+cc[d,:,0].sum())/diag[d].min(); kap=(1+ba2)*snr
        e0=max(errs(res[0][3],wg,d).values()); e4=max(errs(res[4][3],wg,d).values())
        if 1e6<=kap<1e8 and t.size<=900: cands.append((e4,e0,kap,ci,d,t,y[d],diag[d],co,res[0][3],res[4][3],wg))
cands.sort(key=lambda r:-r[0])
for e4,e0,kap,ci,d,t,y,diag,co,g0,g4,wg in cands[:4]:
    ll,gt=gp_dense_ld(t,y,diag,co)
    def vs(g,isdict=False):
        e={}
        for nm in ("y","diag","ar","cr","ac","bc","cc","dc"):
            w=gt["g"+nm] if False else gt[nm] if nm in gt else None
        return e
    truth={"y":gt["y"],"diag":gt["diag"],"ar":gt["ar"],"cr":gt["cr"],"ac":gt["ac"],"bc":gt["bc"],"cc":gt["cc"],"dc":gt["dc"]}
    def e_vs_truth(get):
        m=0
        for nm,w in truth.items():
            if w.size: m=max(m, np.abs(get(nm)-w).max()/(np.abs(w).max()+1e-300))
        return m
    seq=e_vs_truth(lambda nm: wg[nm])
    def lane(g):
        mp={"y":g["y"][d],"diag":g["diag"][d],"ar":g["real"][d,:,0],"cr":g["real"][d,:,1],"ac":g["cplx"][d,:,0],"bc":g["cplx"][d,:,1],"cc":g["cplx"][d,:,2],"dc":g["cplx"][d,:,3]}
        return e_vs_truth(lambda nm: mp[nm])
    print("case %d draw %d N %d kappa %.1e | vs C port: lanes %.1e polished %.1e | vs long double: C port %.1e lanes %.1e polished %.1e"%(ci,d,t.size,kap,e0,e4,seq,lane(g0),lane(g4)))
