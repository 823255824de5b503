import sys, ctypes as C, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import pyoracle as po
from cogaps_amd import _capi
lib = sys.argv[1]; G=int(sys.argv[2]); Sm=int(sys.argv[3]); K=int(sys.argv[4]); niter=int(sys.argv[5])
L = _capi.load() if lib=='hip' else _capi.bind(C.CDLL('/root/repo/tests/emul/libcogaps_emul_TESTONLY_w%s.so'%lib))
rng = np.random.default_rng(7)
a0 = rng.gamma(2.0,0.5,(G,3))*(rng.random((G,3))>0.5); p0 = rng.gamma(2.0,0.5,(Sm,3))*(rng.random((Sm,3))>0.3)
d = ((a0@p0.T)*(0.9+0.2*rng.random((G,Sm)))+0.01).astype(np.float32)
kw = dict(nPatterns=K, nIterations=max(niter,2), seed=123)
S = _capi.Session(d, lib=L, **kw)
wA = L.cogaps_reduction_width(S.dims('A')[1]); wP = L.cogaps_reduction_width(S.dims('P')[1])
O = po.Session(d, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4, **kw)
print('W', wA, wP)
t0=time.time(); ok=True
for it in range(niter):
    temp=min(1.0,2*it/max(niter,2)); S.set_annealing(temp); O.set_annealing(temp)
    nA,nP=S.draw_steps(); assert (nA,nP)==O.draw_steps()
    S.iterate(nA,nP); O.iterate(nA,nP)
    for w in 'AP':
        a,b=S.atoms(w),O.atoms(w)
        for f in a:
            if not np.array_equal(a[f],b[f]): print(it,w,'atoms',f,'MISMATCH'); ok=False
        if not np.array_equal(S.matrix(w),O.matrix(w)): print(it,w,'matrix MISMATCH'); ok=False
        if not np.array_equal(S.ap(w),O.ap(w)): print(it,w,'AP MISMATCH'); ok=False
        if S.chisq(w)!=O.chisq(w): print(it,w,'chisq MISMATCH',S.chisq(w),O.chisq(w)); ok=False
    if not ok: break
print('OK' if ok else 'FAILED', G,Sm,K,niter,'atoms',S.natoms('A'),S.natoms('P'),'%.1fs'%(time.time()-t0))
