import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import exoplanet_amd as xo
from exoplanet_amd.gp import terms
from oracle import numpy_port as P
dev=torch.device('cuda:0')
T=lambda a: torch.as_tensor(np.asarray(a,dtype=np.float64),device=dev)
rng=np.random.default_rng(23)
N,D=400,3
t=np.sort(rng.uniform(0,30,N))
Qs,rhos,sig=np.array([0.3,0.9,4.0]),np.array([3.0,5.0,2.0]),0.7
kernel=terms.SHOTerm(sigma=T(np.full(D,sig)),rho=T(rhos),Q=T(Qs))
gp=xo.gp.GaussianProcess(kernel,t=T(t),yerr=0.3)
x=rng.normal(size=(D,N)); y=rng.normal(size=N)
for ch in ("0","1"):
    os.environ["EXO_GP_CHUNKS"]=ch
    al=gp.apply_inverse(T(y)).cpu().numpy()
    ll=gp.log_likelihood(T(y)).cpu().numpy()
    for d in range(D):
        co=P.sho_coefficients(*P.sho_from_sigma_rho(sig,rhos[d],Qs[d]),Qs[d])
        K=P.celerite_kernel(t[:,None]-t[None,:],*co)+0.09*np.eye(N)
        want=np.linalg.solve(K,y)
        wl,_=P.gp_loglike_dense(t,y,np.full(N,0.09),co)
        print('chunks',ch,'d',d,'alpha err',np.abs(al[d]-want).max()/np.abs(want).max(),'ratio',np.median(al[d]/want),'ll err',abs(ll[d]-wl)/abs(wl))
ts=np.sort(rng.uniform(-2,32,211))
mu=gp.predict(T(y),t=T(ts)).cpu().numpy()
for d in range(D):
    co=P.sho_coefficients(*P.sho_from_sigma_rho(sig,rhos[d],Qs[d]),Qs[d])
    K=P.celerite_kernel(t[:,None]-t[None,:],*co)+0.09*np.eye(N)
    want=P.celerite_kernel(ts[:,None]-t[None,:],*co)@np.linalg.solve(K,y)
    print('predict d',d,np.abs(mu[d]-want).max()/np.abs(want).max())
