import mpmath as mp
mp.mp.dps = 60
def q(a):
    a = mp.mpf(a)
    if a == 0: return mp.mpf(-1)/3
    t = mp.sqrt(a)
    return (mp.atan(t)/t - 1)/a
def fit(n):
    # interpolate q at n Chebyshev nodes on [0,1] -> monomial coefficients (solve Vandermonde in mp)
    nodes = [(mp.cos(mp.pi*(2*k+1)/(2*n))+1)/2 for k in range(n)]
    A = mp.matrix(n, n); b = mp.matrix(n, 1)
    for i, x in enumerate(nodes):
        for j in range(n): A[i, j] = x**j
        b[i] = q(x)
    c = mp.lu_solve(A, b)
    return [c[i] for i in range(n)]
def err(c, rounded=True):
    cc = [mp.mpf(float(x)) for x in c] if rounded else c
    worst = 0
    for k in range(2001):
        t = mp.mpf(k)/2000
        a = t*t
        p = mp.mpf(0)
        for x in reversed(cc): p = p*a + x
        approx = t + t*a*p
        ex = mp.atan(t)
        if t > 0:
            worst = max(worst, abs(approx-ex)/ex)
    return worst
for n in (17, 18, 19, 20, 21, 22):
    c = fit(n)
    print(n, mp.nstr(err(c, False), 5), mp.nstr(err(c, True), 5))
c = fit(20)
for i, x in enumerate(c):
    print(i, repr(float(x)))
