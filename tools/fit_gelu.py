import numpy as np
from scipy.special import erfc, erf
np.set_printoptions(precision=17)
TMAX=4.3
def Q(t):
    t=np.asarray(t,dtype=np.float64)
    out=np.empty_like(t)
    small=t<1e-8
    out[small]=-(2/np.sqrt(np.pi))*np.log2(np.e)
    ts=t[~small]
    out[~small]=np.log2(erfc(ts))/ts
    return out
# weighted least squares on dense Chebyshev-ish grid; weight = t^2*erfc(t) (sensitivity of gelu to dQ) + floor
n=20001
t=0.5*TMAX*(1-np.cos(np.pi*(np.arange(n)+0.5)/n))
for deg in (6,7,8,9):
    w=t*t*erfc(t)+1e-3
    # iterate reweighting (Lawson) for minimax of weighted error
    lw=np.ones_like(t)
    for it in range(60):
        V=np.vander(2*t/TMAX-1,deg+1,increasing=True)
        c,*_=np.linalg.lstsq(V*(w*lw)[:,None],Q(t)*w*lw,rcond=None)
        err=np.abs((V@c-Q(t))*w)
        lw=lw*(err/err.max()+1e-3)**0.5
        lw/=lw.max()
    # convert to monomials in t
    from numpy.polynomial import polynomial as P
    # p(u), u=2t/TMAX-1
    mono=np.zeros(1)
    base=np.array([-1.0,2/TMAX])
    pw=np.array([1.0])
    for k in range(deg+1):
        mono=P.polyadd(mono,c[k]*pw)
        pw=P.polymul(pw,base)
    # evaluate gelu in float32 emulation
    x=np.linspace(-8,8,400001).astype(np.float32)
    ax=np.abs(x)
    tt=np.minimum(ax*np.float32(0.70710678118654752),np.float32(TMAX)).astype(np.float32)
    q=np.float32(mono[-1])*np.ones_like(tt)
    for k in range(deg-1,-1,-1):
        q=(q*tt+np.float32(mono[k])).astype(np.float32)
    E=np.exp2((q*tt).astype(np.float32)).astype(np.float32)
    g=(np.maximum(x,0)-np.float32(0.5)*ax*E).astype(np.float32)
    ref=0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
    e=np.abs(g-ref)
    print(deg,"max abs err",e.max(),"at x=",x[e.argmax()],"max rel-to-max(|x|,1e-3)",(e/np.maximum(np.abs(ref),1e-3)).max())
    print("  coeffs (c0..):",[float(np.float32(m)) for m in mono])
# current A&S for comparison
x=np.linspace(-8,8,400001).astype(np.float32)
z=np.abs(x)*np.float32(0.70710678)
tt=(1/(1+np.float32(0.3275911)*z)).astype(np.float32)
poly=((((np.float32(1.061405429)*tt+np.float32(-1.453152027))*tt+np.float32(1.421413741))*tt+np.float32(-0.284496736))*tt+np.float32(0.254829592))*tt
er=1-poly*np.exp(-z*z)
g=0.5*x*(1+np.sign(x)*er)
ref=0.5*x.astype(np.float64)*(1+erf(x.astype(np.float64)/np.sqrt(2)))
print("A&S max abs err",np.abs(g-ref).max())
