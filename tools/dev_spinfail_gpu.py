"""dev: the lost-hand-over recovery on the HARDWARE.  A library built with -DGEN_TEST_SPIN_FAIL_EPOCH=N (every third applier lane gives up at batch N of each
sampler, as tests/test_emul_parity.py injects it on the emulator) against the product library: the same chain, state for state.
    COGAPS_SPINFAIL_LIB=cogaps_amd/csrc/libcogaps_hip_AB_spinfail.so python tools/dev_spinfail_gpu.py"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from cogaps_amd import _capi
import bench
inj = _capi.bind(ctypes.CDLL(os.environ["COGAPS_SPINFAIL_LIB"]))
data = bench.synthetic_dense(4000, 1600)
out = []
for lib in (None, inj):
    kw = dict(nPatterns=20, seed=7, nIterations=40)
    S = _capi.Session(data, lib=lib, **kw) if lib is not None else _capi.Session(data, **kw)
    S.run_iterations(1, 0, 40); S.run_iterations(2, 0, 40)
    st = {w: (S.atoms(w)["pos"].copy(), S.atoms(w)["mass"].copy(), S.matrix(w).copy(), S.ap(w).copy()) for w in "AP"}
    out.append(st)
    print("recoveries", {w: S.chain_recoveries(w) for w in "AP"}, "chained", {w: S.chained(w) for w in "AP"}, "atoms", S.natoms("A"), S.natoms("P"))
    S.close()
same = all(np.array_equal(a, b) for w in "AP" for a, b in zip(out[0][w], out[1][w]))
print("same state:", same)
sys.exit(0 if same else 1)
