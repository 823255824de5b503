import sys, ctypes as C, numpy as np, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import pyoracle as po
from cogaps_amd import _capi
win = sys.argv[1] if len(sys.argv)>1 else '256'
which = sys.argv[2] if len(sys.argv)>2 else 'gist'
niter = int(sys.argv[3]) if len(sys.argv)>3 else 40
L = _capi.load() if win=='hip' else _capi.bind(C.CDLL('/root/repo/tests/emul/libcogaps_emul_TESTONLY_w%s.so'%win))
if which=='gist':
    d = po.read_mtx('/root/repo/tests/golden/GIST.mtx'); K=7
else:
    d = np.loadtxt('/root/repo/tests/golden/modsimdata.csv', delimiter=',').astype(np.float32); K=3
NI=max(niter,2)
kw = dict(nPatterns=K, nIterations=NI, seed=42)
S = _capi.Session(d, lib=L, **kw)
wA = L.cogaps_reduction_width(S.dims('A')[1]); wP = L.cogaps_reduction_width(S.dims('P')[1])
O = po.Session(d, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4, **kw)
def cmp_trace(a, b, tag):
    ok = True
    if len(a['nproc'])!=len(b['nproc']) or not np.array_equal(a['nproc'], b['nproc']) or not np.array_equal(a['qlen'], b['qlen']):
        print(tag, 'BATCH MISMATCH', len(a['nproc']), len(b['nproc'])); 
        n=min(len(a['nproc']),len(b['nproc'])); bad=np.nonzero((a['nproc'][:n]!=b['nproc'][:n])|(a['qlen'][:n]!=b['qlen'][:n]))[0]
        if len(bad): i=bad[0]; print('  first bad batch', i, a['nproc'][max(0,i-2):i+3], b['nproc'][max(0,i-2):i+3], a['qlen'][max(0,i-2):i+3], b['qlen'][max(0,i-2):i+3])
        ok=False
    n = min(len(a['rec']), len(b['rec']))
    for f in ['type','r1','c1','r2','c2','pos','rng_state','atom1','atom2','batch']:
        x, y = a['rec'][f][:n], b['rec'][f][:n]
        if f in ('pos',):
            m = a['rec']['type'][:n]==ord('M'); x, y = x[m], y[m]
        if f=='atom2':
            m = a['rec']['type'][:n]==ord('E'); x, y = x[m], y[m]
        if not np.array_equal(x, y):
            i = np.nonzero(x!=y)[0][0]; print(tag, 'field', f, 'first mismatch at', i, x[i], y[i], 'batch', a['rec']['batch'][i] if f not in('pos','atom2') else '?'); ok=False
    if len(a['rec'])!=len(b['rec']): print(tag,'len', len(a['rec']), len(b['rec'])); ok=False
    return ok
t0=time.time(); tot=0
for it in range(niter):
    temp = min(1.0, 2*it/NI); S.set_annealing(temp); O.set_annealing(temp)
    nA, nP = S.draw_steps(); oA, oP = O.draw_steps()
    assert (nA,nP)==(oA,oP), (it, nA,nP,oA,oP)
    tot += nA+nP
    ta = S.update('A', nA, 1<<16); tb = O.update('A', nA, 1<<16)
    ok = cmp_trace(ta, tb, 'it%d A'%it)
    S.sync('P'); O.sync('P')
    ta = S.update('P', nP, 1<<16); tb = O.update('P', nP, 1<<16)
    ok &= cmp_trace(ta, tb, 'it%d P'%it)
    S.sync('A'); O.sync('A')
    for w in 'AP':
        a, b = S.atoms(w), O.atoms(w)
        for f in a:
            if not np.array_equal(a[f], b[f]): print('it',it,w,'atoms',f,'mismatch'); ok=False
        if not np.array_equal(S.matrix(w), O.matrix(w)): print('it',it,w,'matrix mismatch'); ok=False
        if not np.array_equal(S.ap(w), O.ap(w)): print('it',it,w,'AP mismatch', np.abs(S.ap(w)-O.ap(w)).max()); ok=False
        if S.avg_queue(w)!=O.avg_queue(w): print('avgq mismatch', w, S.avg_queue(w), O.avg_queue(w)); ok=False
        if S.chisq(w)!=O.chisq(w): print('chisq mismatch', w, S.chisq(w), O.chisq(w)); ok=False
    if not ok: print('FAILED at iteration', it); break
else:
    print('win',win,which,'all',niter,'iterations bit-identical; atoms', S.natoms('A'), S.natoms('P'), 'proposals', tot, 'time %.1f'%(time.time()-t0), 'avgq', S.avg_queue('A'), S.avg_queue('P'))
