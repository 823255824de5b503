import sys, os, time, numpy as np
R='/root/repo'
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests')); sys.path.insert(0, os.path.join(R, 'oracle'))
from cogaps_amd import _capi
import parity_util as pu
data = pu.synthetic_counts(5000, 1250, zeros=0.95, rank=10, seed=1)
S = _capi.Session(data, nPatterns=50, nIterations=60, seed=42, sparseOptimization=True)
for k in range(8):
    t=time.time(); u=S.run_iterations(1, 5*k, 5); print('5 iterations', time.time()-t, u, S.natoms('A'), S.natoms('P'), flush=True)
