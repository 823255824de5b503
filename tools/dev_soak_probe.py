"""dev: the chained launch beside a foreign kernel, iteration by iteration, with the counters printed as it goes (what was the last thing that worked)"""
import sys, os, json, subprocess, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tools")]
import dev_soak, bench
from cogaps_amd import _capi
data = bench.synthetic_dense(20000, 2000)
proc = subprocess.Popen([sys.executable, "-c", dev_soak.FOREIGN, "120"], stdout=subprocess.PIPE, text=True)
assert proc.stdout.readline().strip() == "ready"
S = _capi.Session(data, nIterations=100, nPatterns=50, seed=42, outputFrequency=10)
for it in range(0, 120, 4):
    t0 = time.time()
    S.run_iterations(1 if it < 100 else 2, it if it < 100 else it - 100, 4)
    print(it, "%.2f s" % (time.time() - t0), {w: (S.chained(w), S.chain_recoveries(w), S.natoms(w)) for w in "AP"}, flush=True)
