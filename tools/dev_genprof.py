"""dev: per-phase cycle breakdown of the generator kernel (library built with -DGEN_PROFILE)."""
import sys, ctypes as C, numpy as np, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
import torch
d = synthetic_dense(20000, 2000)
S = _capi.Session(d, nPatterns=50, nIterations=100, seed=42)
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S.run_iterations(1, 0, warm)
p0 = S.perf()
t0 = time.time(); upd = S.run_iterations(1, warm, 10); dt = time.time() - t0
p1 = S.perf()
print('props/s %.3g' % (upd / dt), 'batches', p1['batches'] - p0['batches'], 'us/batch', 1e6 * dt / (p1['batches'] - p0['batches']), 'atoms', S.natoms('A'), S.natoms('P'), 'avgq', S.avg_queue('A'), S.avg_queue('P'))
for w in 'AP':
    pr = S.debug_prof(w)
    tot = sum(pr[:8]) or 1
    names = ['flush', 'A1 type+scan', 'A2 draws', 'B1 register', 'B2 checks', 'C scan', 'C commit', 'serial births+bookkeeping']
    print(w, 'rounds', pr[15], 'total Mcycles %.1f' % (tot / 1e6))
    for i, n in enumerate(names):
        print('   %-28s %6.1f%%  %8.0f cycles/round' % (n, 100 * pr[i] / tot, pr[i] / max(1, pr[15])))
for w in 'AP':
    pr = S.debug_prof(w)
    print(w, 'eval block0 marks (cycles total): queue-rec %d | scalars+sync %d | death: alpha+reduce %d | death: gibbs+log %d | rest(update etc) %d' % tuple(pr[8:13]))
for w in 'AP':
    pr = S.debug_prof(w)
    print(w, 'prologue (entry loads + sync) cycles/round: %.0f' % (pr[14] / max(1, pr[15])))
