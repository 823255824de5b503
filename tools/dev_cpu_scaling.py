"""dev: the CPU port's rate on the headline chain per OpenMP thread count, over iterations [lo, hi) (populated chain)."""
import sys, os, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "oracle"))
import pyoracle as po
from bench import synthetic_dense
lo, hi = int(sys.argv[1]), int(sys.argv[2])
data = synthetic_dense(20000, 2000)
for threads in [int(x) for x in sys.argv[3:]]:
    O = po.Session(data, omp=True, maxThreads=threads, math_mode=po.MATH_LIBM, redW_A=1, redW_P=1, redG=1, nPatterns=50, nIterations=100, seed=42, outputFrequency=10)
    props = 0; t0 = None
    for it in range(hi):
        if it == lo: t0 = time.time(); props = 0
        O.set_annealing(min(1.0, 2.0 * it / 100)); nA, nP = O.draw_steps(); O.iterate(nA, nP); props += nA + nP
    dt = time.time() - t0
    print("threads %3d: iterations %d-%d: %d proposals in %.1f s = %.3g proposals/s" % (threads, lo + 1, hi, props, dt, props / dt), flush=True)
    O.close()
