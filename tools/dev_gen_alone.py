"""dev: wall time per launch of the generator alone (kind 2: batches are generated but never evaluated) against
(generator, evaluation) pairs (kind 3) from the same populated state -- what the evaluation kernel's passage through
the caches costs the generator."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
warm = int(sys.argv[1]) if len(sys.argv) > 1 else 40
for which in 'AP':
    for kind, name in ((2, 'generator alone'), (3, 'generator + evaluation pairs')):
        S = _capi.Session(synthetic_dense(20000, 2000), nPatterns=50, nIterations=100, seed=42)
        S.run_iterations(1, 0, warm)
        us = S.debug_replay(which, kind, 150, 0)
        print(which, '%-30s %7.2f us/launch' % (name, us), flush=True)
        S.close()
