"""dev: per-launch wall time of the evaluation kernel replayed on a populated queue, with parts switched off
(library built with -DGEN_PROFILE)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
import ctypes
PL = _capi.bind(ctypes.CDLL(os.path.join(os.path.dirname(_capi.LIB_PATH), 'libcogaps_hip_REPLAY_DEV.so')))
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for which in 'AP':
    for flags, name in [(0, 'full'), (2, 'no AP update'), (6, 'no row loads, no update'), (14, '... and no scalar step'), (30, '... and no reduction'), (1, 'record only')]:
        S = _capi.Session(synthetic_dense(20000, 2000), lib=PL, nPatterns=50, nIterations=100, seed=42)
        S.run_iterations(1, 0, warm)
        us = S.debug_replay(which, 1, 200, flags)
        print(which, '%-26s %7.2f us/launch' % (name, us), flush=True)
        S.close()
    S = _capi.Session(synthetic_dense(20000, 2000), lib=PL, nPatterns=50, nIterations=100, seed=42)
    S.run_iterations(1, 0, warm)
    print(which, 'gen+eval pair              %7.2f us' % S.debug_replay(which, 0, 200, 0), flush=True)
    S.close()
