"""dev: aggregate proposals/s of C independent chains (sessions) run concurrently on one GPU, one host thread and one stream each."""
import sys, os, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from cogaps_amd import _capi
from bench import synthetic_dense
C = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_iter = int(sys.argv[2]) if len(sys.argv) > 2 else 30
S = [_capi.Session(synthetic_dense(20000, 2000, seed=12345 + c), nPatterns=50, nIterations=n_iter, seed=42 + c) for c in range(C)]
upd = [0] * C
def work(c, first, n):
    upd[c] = S[c].run_iterations(1, first, n)
def phase(first, n):
    th = [threading.Thread(target=work, args=(c, first, n)) for c in range(C)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    return time.perf_counter() - t0
phase(0, 5)
dt = phase(5, n_iter - 5)
print("chains %d: %.3f s, %d proposals, aggregate %.3f M proposals/s (%.3f M per chain)" % (C, dt, sum(upd), sum(upd) / dt / 1e6, sum(upd) / dt / 1e6 / C))
for s in S: s.close()
