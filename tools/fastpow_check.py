import numpy as np, math
LN2 = math.log(2.0)
CL = [2.0/(LN2*(2*k+1)) for k in range(11)]
CE = [LN2**k/math.factorial(k) for k in range(14)]
def fast_pow(x, p):
    x = np.asarray(x, np.float64)
    m, e = np.frexp(x)          # m in [0.5,1)
    small = m < math.sqrt(0.5)
    m = np.where(small, m*2, m); e = np.where(small, e-1, e)
    num, den = m-1.0, m+1.0
    r = 1.0/den
    t = num*r
    t = t + (num - den*t)*r
    t2 = t*t
    P = np.full_like(t, CL[10])
    for k in range(9,-1,-1): P = P*t2 + CL[k]
    L = e + t*P
    z = p*L
    n = np.rint(z); rr = z-n
    Q = np.full_like(rr, CE[13])
    for k in range(12,-1,-1): Q = Q*rr + CE[k]
    return np.ldexp(Q, n.astype(np.int64))
rng = np.random.default_rng(0)
xs = np.concatenate([rng.uniform(0,1,200000), 10**rng.uniform(-38,4,200000), rng.uniform(0.83,1.01,200000)]).astype(np.float32).astype(np.float64)
xs = xs[xs>0]
for p in (0.159423828125, 78.84375, 0.4166666666666667):
    ref = np.power(xs.astype(np.longdouble), np.longdouble(p))
    got = fast_pow(xs, p)
    ok = np.isfinite(ref.astype(np.float64)) & (ref.astype(np.float64) > 1e-300)
    rel = np.abs((got[ok].astype(np.longdouble)-ref[ok])/ref[ok])
    libm = np.power(xs, p)
    rel2 = np.abs((libm[ok].astype(np.longdouble)-ref[ok])/ref[ok])
    print(p, 'fast max rel err %.3g  libm %.3g' % (rel.max(), rel2.max()))
# end-to-end PQ float results vs libm version
f = np.concatenate([rng.uniform(0,1,2000000), rng.uniform(0,12,200000)]).astype(np.float32)
def pq(f, powf):
    d = powf(f.astype(np.float64), 0.159423828125)
    return powf((0.8359375 + 18.8515625*d)/(1.0+18.6875*d), 78.84375).astype(np.float32)
a = pq(f, np.power); b = pq(f, fast_pow)
print('PQ float mismatches', int((a.view(np.uint32)!=b.view(np.uint32)).sum()), 'of', f.size, 'max ulp', int(np.abs(a.view(np.int32).astype(np.int64)-b.view(np.int32)).max()))
s1 = (1.055*np.power(f.astype(np.float64),0.4166666666666667).astype(np.float32)-0.055).astype(np.float32); s2=(1.055*fast_pow(f.astype(np.float64)+1e-30,0.4166666666666667).astype(np.float32)-0.055).astype(np.float32)
print('sRGB mismatches', int((s1.view(np.uint32)!=s2.view(np.uint32)).sum()))
print(['%.17g'%c for c in CL]); print(['%.17g'%c for c in CE])
