"""dev: per-wave mark sequence of a generator launch whose batch needed two rounds (build: make -C cogaps_amd/csrc libcogaps_hip_ROUND2_DEV.so)."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
PL = _capi.bind(ctypes.CDLL(os.path.join(os.path.dirname(_capi.LIB_PATH), sys.argv[2] if len(sys.argv) > 2 else 'libcogaps_hip_ROUND2_DEV.so')))
data = synthetic_dense(20000, 2000)
S = _capi.Session(data, lib=PL, nPatterns=50, nIterations=100, seed=42)
S.run_iterations(1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
names = {0: 'entry', 1: 'B0', 2: 'helper: round vars', 3: 'helper: flush written back', 4: 'round top', 5: 'A1 pcg+guess', 6: 'A1 count3 #1', 7: 'A1 exact decide', 8: 'A1 count3 #2+perm', 9: 'A1 sync', 10: 'A2 stage1',
         25: 'A2 join', 11: 'A2 stage2 (vec)', 12: 'A2 stage3 (atoms)', 13: 'A2 nb loads issued', 14: 'A2 finish', 15: 'B1 registrations', 16: 'B1 sync', 17: 'B2 lookups', 18: 'B2 logic', 19: 'B2 sync', 20: 'C masks sync',
         21: 'C commit issued', 22: 'wave ends', 26: 'entry: loads issued', 27: 'entry: table preset', 28: 'entry: kernargs in', 29: 'entry: scalars in LDS', 23: 'helper: books done', 24: 'helper: write-back issued'}
WAVES = 5
buf = (ctypes.c_uint64 * (WAVES * 64))()
PL.cogaps_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_int]
assert PL.cogaps_debug_timeline(buf, WAVES * 64) == 0
a = np.array(buf).reshape(WAVES, 64)
for w in range(WAVES):
    print('wave %d' % w)
    t0 = None; last = None
    for x in a[w]:
        x = int(x)
        if not x: continue
        ident, c = x & 0xFF, x >> 8
        if ident not in names: continue
        if t0 is None: t0 = c; last = c
        if c < last or c - t0 > 400000: continue      # stale LDS words between the rounds' mark ranges
        print('   %-28s %7d (+%5d)' % (names[ident], c - t0, c - last)); last = c
