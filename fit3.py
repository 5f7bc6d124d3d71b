import numpy as np
from scipy.special import erfc
from scipy.optimize import least_squares
v = np.linspace(0, 9, 45001)
tgt = erfc(v/np.sqrt(2))
def model(c, v):      # E = exp2(-(c0 + c1 v + ... ))
    q = np.zeros_like(v)
    for ck in c[::-1]:
        q = q*v + ck
    return np.exp2(-q)
def fit(n):
    m = (v>0.05)&(v<6)
    y = -np.log2(tgt[m])
    c = np.polyfit(v[m], y, n-1, w=np.sqrt(tgt[m])*v[m])[::-1]
    res = lambda c: 0.5*v*(model(c, v)-tgt)*1e7
    for p in (2,4,8,16,32,64):
        f = lambda c: np.sign(res(c))*np.abs(res(c))**(p/2)
        c = least_squares(f, c, method="lm", xtol=1e-15, ftol=1e-15, max_nfev=6000).x
    return c, np.abs(0.5*v*(model(c,v)-tgt)).max(), np.abs(model(c,v)-tgt).max()
for n in (5,6,7):
    c,e,e2 = fit(n)
    vv = np.linspace(0,3e4,300001); q=np.polyval(c[::-1],vv)
    print("%d coef (deg %d with constant): gelu err %.3g erfc err %.3g  min P beyond 9: %.3g lead %g"%(n,n-1,e,e2,q[vv>9].min(), c[-1]))
    print("   ", repr(c))
