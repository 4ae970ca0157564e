"""time-parallel vs sequential celerite path: agreement and time (run on the GPU box)"""
import os, sys, time, subprocess, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def run(D, N, kind):
    import exoplanet_amd as xo
    from exoplanet_amd.gp import terms
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(3)
    t = torch.tensor(np.sort(rng.uniform(0, N * 2.0 / 1440, N)), device=dev)
    y = torch.tensor(rng.normal(size=(D, N)) * 1e-3, device=dev, requires_grad=True)
    def leaf(v): return torch.tensor(v * (1 + 0.05 * rng.normal(size=(D,))), device=dev, requires_grad=True)
    if kind == "sho":
        ps = [leaf(2e-3), leaf(1.5), leaf(2.0)]
        mk = lambda: terms.SHOTerm(sigma=ps[0], rho=ps[1], Q=ps[2])
    elif kind == "wide":      # J = 8: four SHO terms
        ps = [leaf(2e-3), leaf(1.5), leaf(1e-3), leaf(4.0), leaf(5e-4), leaf(9.0), leaf(1.5e-3), leaf(0.7)]
        mk = lambda: (terms.SHOTerm(sigma=ps[0], rho=ps[1], Q=2.0) + terms.SHOTerm(sigma=ps[2], rho=ps[3], Q=1.0)
                      + terms.SHOTerm(sigma=ps[4], rho=ps[5], Q=0.8) + terms.SHOTerm(sigma=ps[6], rho=ps[7], Q=3.0))
    else:
        ps = [leaf(2e-3), leaf(3.0), leaf(5.0), leaf(1.5), leaf(0.6), leaf(1e-3), leaf(0.3)]
        mk = lambda: terms.RotationTerm(sigma=ps[0], period=ps[1], Q0=ps[2], dQ=ps[3], f=ps[4]) + terms.RealTerm(a=ps[5] ** 2, c=ps[6])
    yerr = torch.tensor(5e-4, device=dev)
    def step():
        gp = xo.gp.GaussianProcess(mk(), t=t, yerr=yerr)
        ll = gp.log_likelihood(y)
        g = torch.autograd.grad(ll.sum(), [y] + ps)
        return ll, g
    ll, g = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ll, g = step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 3 * 1e3
    return dict(ms=ms, ll=ll.detach().cpu().numpy().tolist(), g=[x.detach().double().cpu().numpy().ravel()[:50000].tolist() for x in g])


if __name__ == "__main__":
    if len(sys.argv) > 1:
        D, N, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
        print(json.dumps(run(D, N, kind)))
        sys.exit(0)
    for D, N, kind in ([(int(a), int(b), c) for a, b, c in [x.split(",") for x in os.environ["CASES"].split()]] if os.environ.get("CASES") else [(64, 20000, "sho"), (1024, 150000, "sho"), (128, 65000, "rot"), (5, 3000, "rot")]):
        out = {}
        for mode, env in (("chunked", {}), ("sequential", {"EXO_GP_CHUNKS": "1"})):
            e = dict(os.environ); e.update(env)
            r = subprocess.run([sys.executable, __file__, str(D), str(N), kind], env=e, capture_output=True, text=True)
            if r.returncode != 0:
                print(mode, "FAILED", r.stderr[-2000:]); continue
            out[mode] = json.loads(r.stdout.strip().splitlines()[-1])
        if len(out) == 2:
            a, b = out["chunked"], out["sequential"]
            ll_err = np.max(np.abs(np.array(a["ll"]) - np.array(b["ll"])) / np.abs(np.array(b["ll"])))
            gerr = []
            for x, y_ in zip(a["g"], b["g"]):
                x, y_ = np.array(x), np.array(y_)
                gerr.append(float(np.max(np.abs(x - y_)) / (np.max(np.abs(y_)) + 1e-300)))
            print(f"D={D} N={N} {kind}: chunked {a['ms']:.2f} ms, sequential {b['ms']:.2f} ms; loglike rel err {ll_err:.2e}; grad rel errs {['%.1e' % v for v in gerr]}")
