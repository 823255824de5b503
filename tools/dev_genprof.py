"""dev: per-phase cycle breakdown of the generator kernel (library built with -DGEN_PROFILE)."""
import sys, ctypes as C, numpy as np, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
import torch
d = synthetic_dense(20000, 2000)
import ctypes
PL = _capi.bind(ctypes.CDLL(os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_REPLAY_DEV.so')))
S = _capi.Session(d, lib=PL, nPatterns=50, nIterations=100, seed=42)
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 20
S.run_iterations(1, 0, warm)
p0 = S.perf()
q0 = {w: S.debug_prof(w) for w in 'AP'}
S.set_timing(True)
t0 = time.time(); upd = S.run_iterations(1, warm, 10); dt = time.time() - t0
S.set_timing(False)
p1 = S.perf()
nb = p1['batches'] - p0['batches']
print('props/s %.3g' % (upd / dt), 'batches', nb, 'us/batch', 1e6 * dt / nb, 'atoms', S.natoms('A'), S.natoms('P'), 'avgq', S.avg_queue('A'), S.avg_queue('P'))
print('event-bracket per batch: gen %.2f us (empty %.2f)  eval %.2f us (empty %.2f)' % (
    1e3 * (p1['genMs'] - p0['genMs']) / nb, 1e3 * (p1['genNoopMs'] - p0['genNoopMs']) / max(1, p1['genNoopTimed'] - p0['genNoopTimed']),
    1e3 * (p1['evalMs'] - p0['evalMs']) / nb, 1e3 * (p1['evalNoopMs'] - p0['evalNoopMs']) / max(1, p1['evalNoopTimed'] - p0['evalNoopTimed'])))
names = ['flush', 'A1 type+scan', 'A2 draws', 'B1 register', 'B2 checks', 'C scan', 'C commit', 'serial births+bookkeeping', 'end-of-batch write-back']
allc = 0
for w in 'AP':
    pr = [b - a for a, b in zip(q0[w], S.debug_prof(w))]
    sub = pr[8:13]
    pr[8] = pr[13]
    tot = sum(pr[:9]) or 1
    allc += tot + pr[14]
    print(w, 'rounds', pr[15], 'total Mcycles %.1f' % (tot / 1e6), 'prologue cycles/round %.0f' % (pr[14] / max(1, pr[15])))
    for i, n in enumerate(names):
        print('   %-28s %6.1f%%  %8.0f cycles/round' % (n, 100 * pr[i] / tot, pr[i] / max(1, pr[15])))
print('rounds per batch: %.3f; generator cycles per batch %.0f' % (sum(S.debug_prof(w)[15] - q0[w][15] for w in 'AP') / nb, allc / nb))
