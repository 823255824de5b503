"""Golden vector of the BENCHMARKED chain itself (run in the build container: python tools/make_golden_c3.py; about four
minutes of the eight host cores).

BASELINE configs[2] -- bench.synthetic_dense(20000, 2000), nPatterns = 50, seed 42, nIterations = 100 (+100), outputFrequency 10 --
run end to end by the CPU oracle in the kernels' lane order (reduction widths 512 / 8192 lanes x float4, portable log / exp,
OpenMP over the queue): what `cogaps_run` on the MI355X must reproduce bit for bit, including the iterations bench.py times
(181-200 of the schedule).  The loop is runOnePhase's (reference src/GapsRunner.cpp:272-327).

tests/golden/c3_k50_s42_i100_lane.npz holds
  stepsA / stepsP [200]     the Poisson step counts drawn per iteration (equilibration 0-99, sampling 100-199)
  natomsA / natomsP [200]   domain sizes after every iteration
  atomsA / atomsP / chisq   the histories at outputFrequency 10 (diagnostics$atomsA, $atomsP, $chisq)
  totalUpdates, meanChiSq, avgQueueA / avgQueueP
  sha256_{Amean,Pmean,Asd,Psd}, sha256 of the final atom positions / masses (vector order), factor matrices and A*P caches
  sample_idx_* / sample_*   a 1 % sample of the entries of the four statistics matrices (so that a mismatch can be located)
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pyoracle as po  # noqa: E402
import bench  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "c3_k50_s42_i100_lane.npz")
N_ITER = 100


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def final_state_digests(S):
    """digests of the chain state after the last iteration; S: any session with atoms / matrix / ap (oracle or library)"""
    out = {}
    for w in "AP":
        a = S.atoms(w)
        out["sha256_atoms_pos_" + w] = sha(a["pos"])
        out["sha256_atoms_mass_" + w] = sha(a["mass"])
        out["sha256_matrix_" + w] = sha(S.matrix(w))
        out["sha256_ap_" + w] = sha(S.ap(w))
    return out


def main():
    data = bench.synthetic_dense(20000, 2000)
    O = po.Session(data, omp=True, maxThreads=min(8, os.cpu_count() or 1), math_mode=po.MATH_PORTABLE, redW_A=512, redW_P=8192, redG=4,
                   nPatterns=50, nIterations=N_ITER, seed=42, outputFrequency=10)
    stepsA, stepsP, natA, natP = [], [], [], []
    t0 = time.time()
    for phase in (1, 2):
        for it in range(N_ITER):
            a, b = O.run_iterations(phase, it, 1)
            stepsA.append(int(a[0])), stepsP.append(int(b[0]))
            natA.append(O.natoms("A")), natP.append(O.natoms("P"))
            if it % 10 == 9:
                print("phase %d iteration %d: atoms %d / %d, %.0f s" % (phase, it + 1, natA[-1], natP[-1], time.time() - t0), flush=True)
    state = final_state_digests(O)
    r = O.finish()
    O.close()
    assert r["totalUpdates"] == sum(stepsA) + sum(stepsP)
    rng = np.random.Generator(np.random.MT19937(20260929))
    extra = {}
    for f in ("Amean", "Pmean", "Asd", "Psd"):
        flat = r[f].ravel()
        idx = np.sort(rng.choice(flat.size, size=max(1, flat.size // 100), replace=False)).astype(np.uint32)
        extra["sha256_" + f] = sha(r[f])
        extra["sample_idx_" + f] = idx
        extra["sample_" + f] = flat[idx].copy()
    np.savez_compressed(OUT, stepsA=np.array(stepsA, np.uint32), stepsP=np.array(stepsP, np.uint32), natomsA=np.array(natA, np.uint32), natomsP=np.array(natP, np.uint32),
                        atomsA=r["atomsA"], atomsP=r["atomsP"], chisq=r["chisq"], totalUpdates=np.uint64(r["totalUpdates"]), meanChiSq=np.float32(r["meanChiSq"]),
                        avgQueueA=np.float32(r["averageQueueLengthA"]), avgQueueP=np.float32(r["averageQueueLengthP"]), **state, **extra)
    print("written", OUT, "totalUpdates", r["totalUpdates"], "meanChiSq", r["meanChiSq"], "queue", r["averageQueueLengthA"], r["averageQueueLengthP"], "%.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
