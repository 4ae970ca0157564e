"""CPU: the states (F, S) entering the chunks as the scans deliver them against the exact ones (the chunks run one after the\nother, harness_set_polish(-1)) for one ill-conditioned draw: relative errors of 1e-5 .. 1e-2 in O(1) components --\nthe information-form elements (1 / diag ~ 1e8 next to unobserved directions) are where the accuracy goes."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
import gp_host_lab as L
import test_gp_host as H
lib = L.build("states", [])
lib.harness_gp_ckpt_layout.restype = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
def run_state(t, y, diag, real, cplx, mode):
    lib.harness_set_polish(mode)
    D, n = y.shape
    n_real, n_complex = real.shape[1], cplx.shape[1]
    ns = lib.harness_gp_state_doubles(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0)
    state = np.full(ns + 8, np.nan); ll = np.empty(D); flags = np.empty(D)
    p = lambda a: a.ctypes.data_as(_dp)
    lib.harness_gp_set_cadence_major(0)
    lib.harness_gp_fwd(p(t), p(y), None, p(diag), ctypes.c_int64(diag.shape[0]), ctypes.c_int64(n), p(real), n_real, p(cplx), n_complex, None, ctypes.c_int64(D), 0, p(ll), p(state), p(flags))
    K = ctypes.c_int64(); span = ctypes.c_int64(); Lc = ctypes.c_int64()
    off = lib.harness_gp_ckpt_layout(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, ctypes.byref(K), ctypes.byref(span), ctypes.byref(Lc))
    return state, off, K.value, span.value, Lc.value, ll
target=(47,6)
for ci,(t, y, diag, cr, cc, dtm) in enumerate(L.cases(1, 60)):
    if ci!=target[0]: continue
    d=target[1]
    t=np.ascontiguousarray(t); y=np.ascontiguousarray(y); diag=np.ascontiguousarray(diag); cr=np.ascontiguousarray(cr); cc=np.ascontiguousarray(cc)
    D,n=y.shape; J=cr.shape[1]+2*cc.shape[1]
    s0,off,K,span,Lc,ll0=run_state(t,y,diag,cr,cc,0)
    s1,_,_,_,_,ll1=run_state(t,y,diag,cr,cc,-1)
    print("J",J,"N",n,"L",Lc,"real (a,c dt)",(cr[d]*[1,dtm]).tolist(),"cplx",(cc[d]*[1,1,dtm,dtm]).tolist(), "diag min", diag[d].min())
    print("ll scan %.15g exact %.15g"%(ll0[d],ll1[d]))
    np.set_printoptions(linewidth=250, precision=3)
    for c in (1,2,5,10,20):
        g=(c*Lc)//span
        a=np.array([s0[off+(g*K+k)*D+d] for k in range(K)]); b=np.array([s1[off+(g*K+k)*D+d] for k in range(K)])
        print("chunk",c,"F scan",a[:J]); print("       F exact",b[:J]); print("       S exact",b[J:]); print("       S relerr",np.abs(a[J:]-b[J:])/(np.abs(b[J:])+1e-300)); print("       F relerr",np.abs(a[:J]-b[:J])/(np.abs(b[:J])+1e-300))
