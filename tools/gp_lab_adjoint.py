"""CPU: the adjoint boundary states (adjoint of the state entering every chunk) from the tree scan and from the serial chain\nagainst the exact ones (chunks run one after the other, harness_set_polish(-3)), and ONE adjoint application from the exact\ndownstream adjoint: Fbar is right to 1e-9, Pbar is off by 1e-3 -- the local term 1/2 (w w^T - J Y) is evaluated stably but\nis ill-conditioned in the well-measured direction of P (1e-8 next to entries of 100)."""
import os, sys, ctypes
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tools')]
import numpy as np
import gp_host_lab as L
lib = L.build("adj", [])
_dp = ctypes.POINTER(ctypes.c_double); _ip=ctypes.POINTER(ctypes.c_int32)
def run_all(t, y, diag, real, cplx, mode, serial):
    lib.harness_set_polish(mode); lib.harness_set_serial_scan(serial)
    D, n = y.shape
    n_real, n_complex = real.shape[1], cplx.shape[1]
    ns = lib.harness_gp_state_doubles(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0)
    state = np.full(ns + 8, np.nan); ll = np.empty(D); flags = np.empty(D)
    p = lambda a: a.ctypes.data_as(_dp)
    lib.harness_gp_set_cadence_major(0)
    args=(p(t), p(y), None, p(diag), ctypes.c_int64(diag.shape[0]), ctypes.c_int64(n), p(real), n_real, p(cplx), n_complex, None, ctypes.c_int64(D), 0)
    lib.harness_gp_fwd(*args, p(ll), p(state), p(flags))
    gll=np.ones(D); g={"y": np.empty((D, n)), "diag": np.empty((D, n)), "diag_sum": np.empty(D), "real": np.empty_like(real), "cplx": np.empty_like(cplx)}
    lib.harness_gp_vjp(*args, p(gll), p(state), p(g["y"]), p(g["diag"]), p(g["diag_sum"]), p(g["real"]), p(g["cplx"]))
    out=(ctypes.c_int64*6)(); lib.harness_gp_offsets(ctypes.c_int64(n), ctypes.c_int64(D), n_real, n_complex, 0, out)
    return state, list(out)
for ci,(t, y, diag, cr, cc, dtm) in enumerate(L.cases(1, 60)):
    if ci!=47: continue
    d=6
    t,y,diag,cr,cc=(np.ascontiguousarray(a) for a in (t,y,diag,cr,cc))
    D,n=y.shape; J=cr.shape[1]+2*cc.shape[1]
    ex,(ob,B,op,K,C,Lc)=run_all(t,y,diag,cr,cc,-3,0)
    def sym_idx(j,l):
        if j>l: j,l=l,j
        return j*J - j*(j-1)//2 + (l-j)
    def exact_adj(c1):   # adjoint of the state entering chunk c1: polish slot (1 or 3): after the sequential run slot 1 holds it (copied), last computed in 3
        v=np.array([ex[op + ((1*C + c1)*K + k)*D + d] for k in range(K)])
        return v[:J], v[J:]
    for name,serial in (("tree",0),("serial",1)):
        s0,_=run_all(t,y,diag,cr,cc,0,serial)
        wF=wS=0; worstc=None
        for c in range(0, C-2):
            # bnd(2, c): adjoint of (F, S) entering chunk c + 1
            Fb=np.array([s0[ob + ((2*C + c)*B + k)*D + d] for k in range(J)])
            Sb=np.array([[s0[ob + ((2*C + c)*B + J + j*J + l)*D + d] for l in range(J)] for j in range(J)])
            eF,eSv=exact_adj(c+1)
            eS=np.array([[eSv[sym_idx(j,l)] for l in range(J)] for j in range(J)])
            Sbs=0.5*(Sb+Sb.T)
            a=np.abs(Fb-eF).max()/np.abs(eF).max(); b=np.abs(Sbs-eS).max()/np.abs(eS).max()
            if a>wF: wF=a; worstc=c
            wS=max(wS,b)
        print(name, "adjoint boundary: worst Fbar err %.1e (chunk %s)  worst Sbar err %.1e"%(wF,worstc,wS))
    np.set_printoptions(linewidth=220, precision=4)
    s0,_=run_all(t,y,diag,cr,cc,0,1)
    for c in (5, 12):
        Fb=np.array([s0[ob + ((2*C + c)*B + k)*D + d] for k in range(J)])
        Sb=np.array([[s0[ob + ((2*C + c)*B + J + j*J + l)*D + d] for l in range(J)] for j in range(J)])
        eF,eSv=exact_adj(c+1)
        eS=np.array([[eSv[sym_idx(j,l)] for l in range(J)] for j in range(J)])
        print("chunk",c,"Sbar serial (sym)\n",0.5*(Sb+Sb.T)); print("Sbar exact\n",eS); print("diff\n",0.5*(Sb+Sb.T)-eS)
    # one adjoint application in double from the EXACT downstream adjoint: Pbar_c = lP_c + Abar^T Pbar_{c+1} Abar + sym(Abar^T Fbar_{c+1} g^T)
    import ctypes
    lib.harness_gp_elem_offset.restype = ctypes.c_int64
    Cc=ctypes.c_int64(); eoff=lib.harness_gp_elem_offset(ctypes.c_int64(n), ctypes.c_int64(D), cr.shape[1], cc.shape[1], 0, ctypes.byref(Cc))
    E=3*J*J+2*J
    for c in (5, 12):
        v=np.array([s0[eoff+(c*E+e)*D+d] for e in range(E)])      # chunk c's ADJOINT element (badj_prep wrote it over the element)
        Ab=v[:J*J].reshape(J,J); g=v[J*J:J*J+J]; lP=v[J*J+J:2*J*J+J].reshape(J,J); lF=v[2*J*J+J:2*J*J+2*J]
        # adjoint of the state entering chunk c + 1 (exact), as adjoint of (F, P): Pbar = -Sbar
        eF1,eS1v=exact_adj(c+1); eS1=np.array([[eS1v[sym_idx(j,l)] for l in range(J)] for j in range(J)])
        eF0,eS0v=exact_adj(c); eS0=np.array([[eS0v[sym_idx(j,l)] for l in range(J)] for j in range(J)])
        x=Ab.T@eF1
        Fb=lF+x
        Pb=lP+Ab.T@(-eS1)@Ab+0.5*(np.outer(x,g)+np.outer(g,x))
        print("chunk",c,": one adjoint application from the exact downstream adjoint: Fbar err %.1e, Pbar err %.1e (vs exact adjoint entering chunk %d)"%(np.abs(Fb-eF0).max()/np.abs(eF0).max(), np.abs(-Pb-eS0).max()/np.abs(eS0).max(), c))
