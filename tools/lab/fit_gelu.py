"""Minimax fit of E(|v|) = exp2(-|v| Q(|v|)) to erfc(|v| / sqrt 2), weighted as the GELU uses it (0.5 |v| (E - erfc)):
  python tools/lab/fit_gelu.py            degree 4 (split_math.h, gelu3_*: 5.4e-7)
  python tools/lab/fit_gelu.py 2          degree 2 (split_math.h, gelu2h_micro -- the fused bf16 kernels: 8.6e-5)
"""
import sys
import numpy as np
from scipy.special import erfc, erf
DEG = int(sys.argv[1]) if len(sys.argv) > 1 else 4
from scipy.optimize import least_squares
v = np.linspace(0, 9, 90001)
tgt = erfc(v/np.sqrt(2))
def model(c, v):
    q = np.zeros_like(v)
    for ck in c[::-1]:
        q = q*v + ck
    return np.exp2(-v*q)
m = (v>0)&(v<6)
y = -np.log2(tgt[m])/v[m]
c = np.polyfit(v[m], y, DEG, w=np.sqrt(tgt[m]))[::-1]
res = lambda c: 0.5*v*(model(c, v)-tgt)*1e7
for p in (2,4,8,16,32,64,128,256):
    f = lambda c: np.sign(res(c))*np.abs(res(c)/6)**(p/2)
    c = least_squares(f, c, method="lm", xtol=1e-15, ftol=1e-15, max_nfev=8000).x
    print(p, np.abs(res(c)).max()/1e7)
print(repr(c))
c32 = c.astype(np.float32)
print([float(x) for x in c32])
# fp32 emulation: z = v (h=1), fma via float64
def f32(x): return np.asarray(x, np.float64).astype(np.float32).astype(np.float64)
def gelu32(vv):
    z = f32(vv); az = np.abs(z)
    k = [-float(x) for x in c32]
    q = np.full_like(az, k[DEG])
    for j in range(DEG - 1, -1, -1):
        q = f32(q*az + k[j])
    q = f32(q*az)
    e = f32(np.exp2(q))
    r = f32(1.0 - e)
    return f32(az*(0.5*r) + 0.5*z)      # h = 0.5 folded exactly
vv = np.linspace(-12, 12, 480001)
ref = 0.5*vv*(1+erf(vv/np.sqrt(2)))
g = gelu32(vv)
err = np.abs(g-ref)
print("fp32-emulated: max abs err %.3g at v=%.3f ; max rel err (|v|>1e-3) %.3g" % (err.max(), vv[err.argmax()], (err/np.maximum(np.abs(ref),1e-30))[np.abs(vv)>1e-3].max()))
for lo,hi in ((-12,-6),(-6,-3),(-3,-1),(-1,0),(0,1),(1,3),(3,6),(6,12)):
    s=(vv>=lo)&(vv<hi); print(lo,hi,"abs %.3g rel %.3g"%(err[s].max(), (err[s]/np.maximum(np.abs(ref[s]),1e-300)).max()))
