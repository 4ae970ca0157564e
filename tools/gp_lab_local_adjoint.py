"""CPU: is the local adjoint term l_P = 1/2 gL (w w^T - J Y) of badj_prep_lane evaluated stably?  Double against 60 digits on the\nsame inputs: yes (1e-9) -- its inaccuracy against the exact recurrences is sensitivity to its inputs, not arithmetic."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np, mpmath as mp
import gp_host_lab as L
mp.mp.dps=60
lib = L.build("lp", [])
lib.harness_gp_elem_offset.restype = ctypes.c_int64
_dp = ctypes.POINTER(ctypes.c_double)
def fwd(t, y, diag, real, cplx, serial):
    lib.harness_set_polish(0); lib.harness_set_serial_scan(serial)
    D, n = y.shape; n_real, n_complex = real.shape[1], cplx.shape[1]
    ns = lib.harness_gp_state_doubles(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0)
    state = np.full(ns + 8, np.nan); ll = np.empty(D); flags = np.empty(D)
    p = lambda a: a.ctypes.data_as(_dp)
    lib.harness_gp_set_cadence_major(0)
    lib.harness_gp_fwd(p(t), p(y), None, p(diag), ctypes.c_int64(diag.shape[0]), ctypes.c_int64(n), p(real), n_real, p(cplx), n_complex, None, ctypes.c_int64(D), 0, p(ll), p(state), p(flags))
    out=(ctypes.c_int64*6)(); lib.harness_gp_offsets(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, out)
    C=ctypes.c_int64(); eoff=lib.harness_gp_elem_offset(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, ctypes.byref(C))
    return state, list(out), eoff
for ci,(t, y, diag, cr, cc, dtm) in enumerate(L.cases(1, 60)):
    if ci!=47: continue
    d=6
    t,y,diag,cr,cc=(np.ascontiguousarray(a) for a in (t,y,diag,cr,cc))
    D,n=y.shape; J=cr.shape[1]+2*cc.shape[1]; E=3*J*J+2*J
    s,(ob,B,op,K,C,Lc),eoff=fwd(t,y,diag,cr,cc,1)
    for c in (5, 9, 12):
        v=np.array([s[eoff+(c*E+e)*D+d] for e in range(E)])
        A=v[:J*J].reshape(J,J); b=v[J*J:J*J+J]; Cm=v[J*J+J:2*J*J+J].reshape(J,J); eta=v[2*J*J+J:2*J*J+2*J]; Jm=v[2*J*J+2*J:].reshape(J,J)
        m=np.array([s[ob+((1*C+c)*B+k)*D+d] for k in range(J)]); P=np.array([[s[ob+((1*C+c)*B+J+j*J+l)*D+d] for l in range(J)] for j in range(J)])
        def lp(xp, inv, mat):
            X=mat(np.eye(J).tolist()) + mat(P.tolist())*mat(Jm.tolist()) if xp=='mp' else np.eye(J)+P@Jm
            Y=inv(X)
            if xp=='mp':
                u=mp.matrix(eta.tolist())-mp.matrix(Jm.tolist())*mp.matrix(m.tolist()); w=Y.T*u; JY=mp.matrix(Jm.tolist())*Y
                return np.array([[float(0.5*(w[j]*w[l]-0.5*(JY[j,l]+JY[l,j]))) for l in range(J)] for j in range(J)]), np.array([float(x) for x in w])
            u=eta-Jm@m; w=Y.T@u; JY=Jm@Y
            return 0.5*(np.outer(w,w)-0.5*(JY+JY.T)), w
        a,wa=lp('mp', mp.inverse, mp.matrix); bb,wb=lp('np', np.linalg.inv, None)
        print("chunk",c,"l_P double vs 60 digits: max abs diff %.2e, max |l_P| %.2e; w rel diff %.1e; |w w^T| %.2e |JY| %.2e"%(np.abs(a-bb).max(), np.abs(a).max(), np.abs(wa-wb).max()/np.abs(wa).max(), np.abs(np.outer(wa,wa)).max(), np.abs(Jm@np.linalg.inv(np.eye(J)+P@Jm)).max()))
