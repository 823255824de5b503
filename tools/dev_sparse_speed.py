"""dev: throughput of the sparse model on a SURVEY-probe-like problem (5000x1250 counts, 95 % zeros, K=50)."""
import sys, os, time, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'oracle'))
from cogaps_amd import _capi
import parity_util as pu
g, s = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (5000, 1250)
data = pu.synthetic_counts(g, s, zeros=0.95, rank=10, seed=1)
for sparse in (True, False):
    S = _capi.Session(data, nPatterns=50, nIterations=60, seed=42, sparseOptimization=sparse)
    S.run_iterations(1, 0, 20)
    t0 = time.time(); upd = S.run_iterations(1, 20, 40); dt = time.time() - t0
    print('sparse' if sparse else 'dense ', '%dx%d' % (g, s), 'proposals/s %.3g' % (upd / dt), 'atoms', S.natoms('A'), S.natoms('P'), 'avg queue', S.avg_queue('A'), S.avg_queue('P'))
    S.close()
