"""dev: one chained launch (csrc/chain_kernel.h) on the chip-wide 100 MHz clock (profile build): when every evaluation workgroup entered,
published its decision and ended, and the generator workgroup's marks -- all relative to the launch's earliest mark, in microseconds."""
import sys, os, ctypes, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
PL = _capi.bind(ctypes.CDLL(os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_PROFILE_DEV.so')))
data = synthetic_dense(20000, 2000)
S = _capi.Session(data, lib=PL, nPatterns=50, nIterations=100, seed=42)
S.run_iterations(1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 40)
w = (ctypes.c_uint64 * 1024)(); g = (ctypes.c_uint64 * 8)()
PL.cogaps_debug_chain_timeline.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
assert PL.cogaps_debug_chain_timeline(w, g) == 0
w = np.array(w, dtype=np.int64).reshape(256, 4); g = np.array(g, dtype=np.int64)
qlen = int(w[0, 3])
# keep the workgroups whose marks belong to the generator's launch (entry within 50 us of it)
ok = [b for b in range(255) if w[b, 0] and abs(int(w[b, 0]) - int(g[0])) < 5000]
t0 = min([int(w[b, 0]) for b in ok] + [int(g[0])])
us = lambda x: (int(x) - t0) / 100.0
print('queue length %d; %d evaluation workgroups of the generator\'s launch' % (qlen, len(ok)))
ent = np.array([us(w[b, 0]) for b in ok]); pub = np.array([us(w[b, 1]) for b in ok if w[b, 1] and b < qlen]); end = np.array([us(w[b, 2]) for b in ok])
for name, a in (('entry', ent), ('published', pub), ('end', end)):
    if len(a): print('  evaluation %-10s min %.2f  median %.2f  p90 %.2f  max %.2f us' % (name, a.min(), np.median(a), np.percentile(a, 90), a.max()))
names = ['entry', 'first barrier passed, fetch begins', 'record + atoms fetched, window staged', 'granules in (poll done)', 'decisions applied + barrier', 'attempt wave 0 ends']
for i, n in enumerate(names):
    print('  generator  %-40s %.2f us' % (n, us(g[i])))
print('  births of the window: looked up ahead %d, the usual way %d, slow path %d' % (int(g[6]) & 0xFFFF, int(g[6]) >> 16, int(g[7])))
sec = [b for b in range(240, 255) if w[b, 1] and abs(int(w[b, 1]) - int(g[0])) < 5000]
if sec: print('  second-pass proposals (queue slots 240..254) published: ' + ' '.join('%.2f' % us(w[b, 1]) for b in sec) + ' us')
