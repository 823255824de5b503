"""dev: run the same short chain several times and compare (the chain must not depend on timing)."""
import sys, os, hashlib, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'oracle'))
from cogaps_amd import _capi
import parity_util as pu
sparse = sys.argv[1] == 'sparse'
g, s, k, iters, reps = (int(x) for x in sys.argv[2:7])
data = pu.synthetic_counts(g, s, zeros=0.95 if sparse else 0.3, rank=10, seed=1)
seen = {}
for r in range(reps):
    S = _capi.Session(data, nPatterns=k, nIterations=60, seed=42, sparseOptimization=sparse)
    upd = S.run_iterations(1, 0, iters)
    h = hashlib.sha1(S.matrix('A').tobytes() + S.matrix('P').tobytes() + S.atoms('A')['pos'].tobytes()).hexdigest()[:12]
    seen.setdefault((upd, S.natoms('A'), S.natoms('P'), h), []).append(r)
    S.close()
for key, runs in seen.items(): print(key, 'runs', runs)
print('DETERMINISTIC' if len(seen) == 1 else 'NONDETERMINISTIC')
