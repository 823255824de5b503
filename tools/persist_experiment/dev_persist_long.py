"""dev: the headline chain for N iterations in the persistent form (COGAPS_PERSIST=on), product or profile build: does it run through, how long does it take"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
lib = _capi.load() if '--profile' not in sys.argv else _capi.bind(ctypes.CDLL(os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_PROFILE_DEV.so')))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 120
S = _capi.Session(synthetic_dense(20000, 2000), lib=lib, nPatterns=50, nIterations=100, seed=42)
t0 = time.time()
done = 0
try:
    for i in range(0, n, 10):
        S.run_iterations(1, i, min(10, n - i)); done = i + 10
        print('iterations', done, 'form', S.launch_form('A'), S.launch_form('P'), 'seconds %.1f' % (time.time() - t0), flush=True)
except Exception as e:
    print('FAILED after', done, 'iterations:', e)
