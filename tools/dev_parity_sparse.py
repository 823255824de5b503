"""dev: sparse-model parity, library (emulator build `win` or `hip`) vs oracle on count data."""
import sys, os, ctypes, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'oracle')); sys.path.insert(0, os.path.join(R, 'tests'))
from cogaps_amd import _capi
import pyoracle as po
which, G, S_, K, iters = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
zeros = float(sys.argv[6]) if len(sys.argv) > 6 else 0.8
L = _capi.load() if which == 'hip' else _capi.bind(ctypes.CDLL(os.path.join(R, 'tests', 'emul', 'libcogaps_emul_TESTONLY_w%s.so' % which)))
rng = np.random.default_rng(G * 31 + S_)
A0 = rng.gamma(2, .5, (G, 4)) * (rng.random((G, 4)) > .5); P0 = rng.gamma(2, .5, (S_, 4)) * (rng.random((S_, 4)) > .4)
D = (np.ceil((A0 @ P0.T) * (0.9 + 0.2 * rng.random((G, S_)))) * (rng.random((G, S_)) > zeros)).astype(np.float32)
kw = dict(nPatterns=K, nIterations=max(iters, 2), seed=11, sparseOptimization=True)
S = _capi.Session(D, lib=L, **kw)
O = po.Session(D, math_mode=po.MATH_PORTABLE, redW_A=L.cogaps_reduction_width(S.dims('A')[1]), redW_P=L.cogaps_reduction_width(S.dims('P')[1]), redG=4, **kw)
t0 = time.time()
for it in range(iters):
    t = min(1.0, 2.0 * it / max(iters, 2)); S.set_annealing(t); O.set_annealing(t)
    nA, nP = S.draw_steps(); assert (nA, nP) == O.draw_steps()
    S.iterate(nA, nP); O.iterate(nA, nP)
    for w in 'AP':
        a, b = S.atoms(w), O.atoms(w)
        for f in ('pos', 'mass', 'left', 'right'):
            assert np.array_equal(a[f], b[f]), 'it %d %s atom %s differs' % (it, w, f)
        assert np.array_equal(S.matrix(w), O.matrix(w)), 'it %d %s column copy differs' % (it, w)
        assert np.array_equal(S.rows(w), O.rows(w)), 'it %d %s row copy differs' % (it, w)
        assert S.chisq(w) == O.chisq(w), 'it %d %s chisq %r %r' % (it, w, S.chisq(w), O.chisq(w))
print('OK sparse', which, G, S_, K, iters, 'atoms', S.natoms('A'), S.natoms('P'), 'zeros', float((D == 0).mean()), '%.1fs' % (time.time() - t0))
