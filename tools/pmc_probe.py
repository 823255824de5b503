"""Workload for the rocprofv3 --pmc passes (profiles/README.md): the headline problem, a short warm-up, then a few
iterations.  Counter collection serialises every dispatch it covers, so the passes restrict it with
--kernel-include-regex / --kernel-iteration-range to a window of evaluation launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cogaps_amd import _capi
from bench import synthetic_dense
S = _capi.Session(synthetic_dense(20000, 2000), nPatterns=50, nIterations=100, seed=42)
upd = S.run_iterations(1, 0, int(sys.argv[1]) if len(sys.argv) > 1 else 12)
print('proposals', upd, 'atoms', S.natoms('A'), S.natoms('P'), 'avg queue', S.avg_queue('A'), S.avg_queue('P'))
S.close()
