"""Round-2 golden vectors (run in the build container: python tools/make_golden_r2.py).

  glibc235_logf_expf.npz    inputs and the outputs of THIS image's libm (GNU libc 2.35, the -mfma ifunc variant on
                            this host) for logf over uniform() outputs and expf over non-positive arguments: the pin
                            of the kernels' COGAPS_MATH_GLIBC_FMA mode.
  gist_k5_s123_i300_seq.npz, gist_k4_s77_i200_sparse_seq.npz
                            full results of the oracle in the reference's arithmetic (sequential sums, libm) on two
                            further configurations whose atom histories / totalUpdates / meanChiSq were printed by the
                            reference binary itself ("Judge's independent oracle check"); asserted
                            below before anything is written.
"""
import ctypes
import ctypes.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import pyoracle as po  # noqa: E402
from make_golden import G, save  # noqa: E402

REFERENCE_PRINTED = {   # the reference core's own output (same evidentiary class as SURVEY.md section 8c)
    "dense": dict(kw=dict(nPatterns=5, nIterations=300, seed=123, outputFrequency=30),
                  atomsA=[512, 1398, 2213, 2768, 3232, 3622, 3569, 3530, 3494, 3387, 3363, 3340, 3204, 3072, 3041, 2949, 2852, 2901, 2938, 2884],
                  atomsP=[13, 21, 28, 30, 33, 37, 42, 43, 45, 47, 48, 48, 51, 53, 55, 58, 57, 55, 57, 60], totalUpdates=1727325),
    "sparse": dict(kw=dict(nPatterns=4, nIterations=200, seed=77, outputFrequency=20, sparseOptimization=True),
                   atomsA=[248, 694, 1152, 1579, 2038, 2428, 2726, 3050, 3299, 3519, 3680, 3777, 3788, 3724, 3566, 3434, 3353, 3324, 3330, 3292],
                   atomsP=[10, 13, 13, 15, 19, 22, 27, 27, 28, 34, 33, 32, 34, 33, 36, 33, 37, 38, 38, 42], totalUpdates=1097330),
}


def libm_vectors():
    libm = ctypes.CDLL(ctypes.util.find_library("m") or "libm.so.6")
    for f in (libm.logf, libm.expf):
        f.restype, f.argtypes = ctypes.c_float, [ctypes.c_float]
    rng = np.random.default_rng(2024)
    u = (rng.integers(0, 2 ** 32, 8000, dtype=np.uint64).astype(np.float32) / np.float32(4294967296.0))      # GapsRng::uniform()
    edges = np.array([0.0, 1.0, 2.0 ** -32, 2.0 ** -31, 1.0 - 2.0 ** -24, 0.5, 0.70710677, 0.70710683, 1e-38, 1e-45, 3.0e-39], dtype=np.float32)
    xl = np.concatenate([u, edges, rng.random(181, dtype=np.float32) * np.float32(1e-3)])
    xe = np.concatenate([-(rng.random(7000, dtype=np.float32) * np.float32(40.0)), -(rng.random(1000, dtype=np.float32) * np.float32(110.0)),
                         np.array([0.0, -0.0, -88.0, -103.0, -103.5, -103.97, -104.0, -200.0, -np.inf, -1e-10, -87.33655], dtype=np.float32)])
    with np.errstate(divide="ignore"):
        yl = np.array([libm.logf(float(v)) for v in xl], dtype=np.float32)
    ye = np.array([libm.expf(float(v)) for v in xe], dtype=np.float32)
    L = po.lib()
    bad = sum(np.float32(L.go_glibc_logf(float(v), 1)).tobytes() != w.tobytes() for v, w in zip(xl, yl)) + \
        sum(np.float32(L.go_glibc_expf(float(v), 1)).tobytes() != w.tobytes() for v, w in zip(xe, ye))
    assert bad == 0, "this host's libm is not the glibc 2.35 -mfma variant the restatement follows (%d differences)" % bad
    np.savez_compressed(os.path.join(G, "glibc235_logf_expf.npz"), x_log=xl, y_log=yl, x_exp=xe, y_exp=ye)
    print("libm vectors:", xl.size, xe.size)


def main():
    libm_vectors()
    gist = po.read_mtx(os.path.join(G, "GIST.mtx"))
    for name, fp in REFERENCE_PRINTED.items():
        r = po.run(gist, **fp["kw"])
        assert r["atomsA"].tolist() == fp["atomsA"] and r["atomsP"].tolist() == fp["atomsP"] and r["totalUpdates"] == fp["totalUpdates"], name
        kw = fp["kw"]
        save("gist_k%d_s%d_i%d_%sseq.npz" % (kw["nPatterns"], kw["seed"], kw["nIterations"], "sparse_" if name == "sparse" else ""), r)
        print(name, "ok", r["meanChiSq"], r["chisq"][-1])


if __name__ == "__main__":
    main()
