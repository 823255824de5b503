"""dev: per-wave timeline (cycles since kernel entry) of one typical generator launch (profile build)."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
PL = _capi.bind(ctypes.CDLL(os.environ.get('COGAPS_PROFILE_LIB', os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_PROFILE_DEV.so'))))
SPARSE = "--sparse" in sys.argv
if SPARSE: sys.argv.remove("--sparse")
data = synthetic_dense(20000, 2000)
if SPARSE: data *= (np.random.Generator(np.random.MT19937(777)).random(data.shape) >= 0.95)
S = _capi.Session(data, lib=PL, nPatterns=50, nIterations=100, seed=42, sparseOptimization=SPARSE)
S.run_iterations(1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
names = {0: 'entry', 1: 'entry loads+sync (B0)', 2: 'helper: round vars set, flush loads issued', 3: 'helper: flush written back (before join)', 4: 'round top', 5: 'A1 pcg+guess', 6: 'A1 count3 #1 (C1)',
         7: 'A1 exact decide', 8: 'A1 count3 #2+perm (C2)', 9: 'A1 sync (C3)', 10: 'A2 stage1 rng/addr', 25: 'A2 join with the flush', 11: 'A2 stage2 (vec/bits0 used)', 12: 'A2 stage3 (atoms/binHead used)',
         13: 'A2 neighbour loads issued', 14: 'A2 finish', 15: 'B1 registrations', 16: 'B1 sync', 17: 'B2 lookups', 18: 'B2 logic', 19: 'B2 sync (Bp)', 20: 'C masks sync (Bc1)', 21: 'C commit issued',
         22: 'attempt wave ends', 26: 'entry: loads issued', 27: 'entry: conflict table preset', 28: 'entry: kernel arguments in', 29: 'entry: first trip landed, scalars in LDS', 23: 'helper: bookkeeping done', 24: 'helper: write-back issued',
         36: 'A2 birth block done', 30: 'chain: before the fetch', 31: 'chain: record + atoms fetched', 32: 'chain: seeds / table window issued', 33: 'chain: granules in (poll done)', 34: 'chain: decisions applied (stores issued)', 35: 'chain: stores acknowledged + barrier', 36: 'chain: window drawn ahead', 37: 'chain: draws validated', 38: 'chain: joined with the flush'}
WAVES = 8
buf = (ctypes.c_uint64 * (WAVES * 64))()
PL.cogaps_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert PL.cogaps_debug_timeline(buf, WAVES * 64) == 0
a = np.array(buf).reshape(WAVES, 64)
seq = {w: [(int(x) & 0xFF, int(x) >> 8) for x in a[w] if x] for w in range(WAVES)}
print('cycles since the wave\'s own first mark (+ since its previous mark); waves 0-3: attempt lanes, wave 4: the helper wave, waves 5-7: further applier waves of a chained launch')
print('%-46s' % 'mark' + ''.join('      wave%d        ' % w for w in range(WAVES)))
order = []
for w in range(WAVES):
    for ident, _ in seq[w]:
        if ident not in order: order.append(ident)
pos = {ident: min(i for w in range(WAVES) for i, (d, _) in enumerate(seq[w]) if d == ident) for ident in order}
order.sort(key=lambda d: (pos[d], d))
for ident in order:
    row = '%-46s' % names.get(ident, str(ident))
    for w in range(WAVES):
        hit = [i for i, (d, _) in enumerate(seq[w]) if d == ident]
        if not hit: row += ' ' * 19; continue
        i = hit[-1]; c = seq[w][i][1] - seq[w][0][1]; d = seq[w][i][1] - seq[w][i - 1][1] if i else 0
        row += ' %7d (%6d)  ' % (c, d)
    print(row)

# ---- evaluation kernel: first 16 workgroups of the last big launch of each sampler (A: 64 lanes, P: 1024 lanes)
ebuf = (ctypes.c_uint64 * (2 * 16 * 2 * 12))()
PL.cogaps_debug_eval_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert PL.cogaps_debug_eval_timeline(ebuf, 2 * 16 * 2 * 12) == 0
ea = np.array(ebuf).reshape(2, 16, 2, 12)
en = {7: 'published', 0: 'entry', 1: 'record', 2: 'scalars', 3: 'reduced', 4: 'scalar math', 5: 'broadcast', 6: 'AP update', 10: 'partials', 11: 'parked', 12: 'barrier'}
if SPARSE: en = {7: 'published', 0: 'entry', 1: 'record', 2: 'rows in LDS', 3: 'terms folded', 4: 'totals', 5: 's, s_mu', 6: 'decision + update'}
g0 = min(seq[w][0][1] for w in range(WAVES) if seq[w])
print('generator waves: first mark, absolute (minus the earliest):', [seq[w][0][1] - g0 for w in range(WAVES) if seq[w]])
for which, e in (('narrow workgroups (A sampler)', ea[0]), ('wide workgroups (P sampler)', ea[1])):
    print()
    print('evaluation kernel, %s, cycles since workgroup entry:' % which)
    for b in range(16):
        if e[b, 0, 1] == 0: continue
        ty = int(e[b, 0, 0])
        for w in range(2):
            ts = [(int(x) & 0xFF, int(x) >> 8) for x in e[b, w, 1:] if x]
            if not ts: continue
            t0 = ts[0][1]
            print('  wg %2d %s nUpd %d sameRow %d wave %s (entry %+d vs the generator workgroup\'s): ' % (b, chr(ty & 0xFF), (ty >> 8) & 0xFF, ty >> 16, 'first' if w == 0 else 'last ', t0 - g0) + '  '.join('%s %d' % (en.get(i, str(i)), c - t0) for i, c in ts[1:]))
