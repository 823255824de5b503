"""The oracle is only trusted once it reproduces the reference: SURVEY.md section 8c records the atom
histories, totalUpdates and meanChiSq of the reference C++ core (scalar -O2 build) on its own test data
(GIST.mtx K=7, modsimdata K=3; seed 42; 1000+1000 iterations; outputFrequency 100).  The oracle in
sequential-reduction / libm mode must reproduce them exactly."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

FINGERPRINTS = {   # SURVEY.md section 8c, "Oracle facts established", item (3)
    "gist": dict(k=7, atomsA=[2875, 3838, 3940, 3123, 3268, 3215, 3372, 3344, 3394, 3422, 3441, 3450, 3507, 3565, 3597, 3496, 3597, 3532, 3600, 3608],
                 atomsP=[36, 51, 58, 66, 70, 75, 78, 80, 81, 87, 85, 90, 88, 94, 100, 100, 101, 95, 96, 95],
                 totalUpdates=6902140, meanChiSq=3200.784, qA=34.9, qP=2.9),
    "modsim": dict(k=3, atomsA=[29, 50, 59, 58, 57, 56, 56, 49, 57, 60, 64, 62, 60, 60, 58, 56, 58, 54, 55, 52],
                   atomsP=[31, 41, 47, 53, 63, 69, 57, 63, 62, 69, 62, 69, 70, 67, 72, 59, 57, 60, 58, 58],
                   totalUpdates=225435, meanChiSq=36.233, qA=4.4, qP=3.9),
}


# Two further outputs of the reference core itself ("Judge's independent oracle check": the same reference
# build as the section 8c probe, run on configurations the oracle had not been pinned on): GIST.mtx dense K=5 seed=123 300+300
# iterations outputFrequency 30, and -- the only reference-derived numbers for the SPARSE model -- K=4 seed=77 200+200
# outputFrequency 20 with sparseOptimization.
REFERENCE_PRINTED = {
    "dense": dict(kw=dict(nPatterns=5, nIterations=300, seed=123, outputFrequency=30), golden="gist_k5_s123_i300_seq.npz",
                  atomsA=[512, 1398, 2213, 2768, 3232, 3622, 3569, 3530, 3494, 3387, 3363, 3340, 3204, 3072, 3041, 2949, 2852, 2901, 2938, 2884],
                  atomsP=[13, 21, 28, 30, 33, 37, 42, 43, 45, 47, 48, 48, 51, 53, 55, 58, 57, 55, 57, 60],
                  totalUpdates=1727325, meanChiSq=4856.156, lastChisq=6161.283, qA=33.599, qP=2.875,
                  probe=("Amean", 0, [0.0009664664, 0.005498413, 0.009075806])),
    "sparse": dict(kw=dict(nPatterns=4, nIterations=200, seed=77, outputFrequency=20, sparseOptimization=True), golden="gist_k4_s77_i200_sparse_seq.npz",
                   atomsA=[248, 694, 1152, 1579, 2038, 2428, 2726, 3050, 3299, 3519, 3680, 3777, 3788, 3724, 3566, 3434, 3353, 3324, 3330, 3292],
                   atomsP=[10, 13, 13, 15, 19, 22, 27, 27, 28, 34, 33, 32, 34, 33, 36, 33, 37, 38, 38, 42],
                   totalUpdates=1097330, meanChiSq=7804.247, lastChisq=9881.170, qA=None, qP=None,
                   probe=("Pmean", 0, [0.9394006, 0.7834437, 0.1044978])),
}


def check_reference_printed(r, fp):
    """every digit the reference binary printed"""
    assert r["atomsA"].tolist() == fp["atomsA"] and r["atomsP"].tolist() == fp["atomsP"]
    assert r["totalUpdates"] == fp["totalUpdates"]
    assert abs(r["meanChiSq"] - fp["meanChiSq"]) < 6e-4 and abs(float(r["chisq"][-1]) - fp["lastChisq"]) < 6e-4
    if fp["qA"] is not None:
        assert abs(r["averageQueueLengthA"] - fp["qA"]) < 6e-4 and abs(r["averageQueueLengthP"] - fp["qP"]) < 6e-4
    f, row, vals = fp["probe"]
    assert np.allclose(r[f][row, :3], vals, rtol=2e-7, atol=0)


@pytest.mark.parametrize("name", ["dense", "sparse"])
def test_reference_printed_outputs(oracle, gist, name):
    fp = REFERENCE_PRINTED[name]
    r = oracle.run(gist, **fp["kw"])
    check_reference_printed(r, fp)
    g = np.load(os.path.join(GOLDEN, fp["golden"]))
    for f in ("Amean", "Pmean", "Asd", "Psd", "chisq"):
        assert np.array_equal(r[f], g[f]), f


def test_glibc_restatement_is_this_hosts_libm(oracle):
    """oracle/gaps_oracle.c restates glibc 2.35's logf / expf (the reference's libm, a dependency outside /root/reference):
    it must agree bit for bit with the C library on a host that runs the -mfma variant -- every 97th float of uniform()'s
    range for logf and of (-inf, 0] for expf here (tools/make_golden_r2.py's run covered every float), and with the
    committed vectors anywhere"""
    L = oracle.lib()
    g = np.load(os.path.join(GOLDEN, "glibc235_logf_expf.npz"))
    for v, w in zip(g["x_log"], g["y_log"]):
        assert np.float32(L.go_glibc_logf(float(v), 1)).tobytes() == w.tobytes(), float(v)
    for v, w in zip(g["x_exp"], g["y_exp"]):
        assert np.float32(L.go_glibc_expf(float(v), 1)).tobytes() == w.tobytes(), float(v)
    fused_here = L.go_glibc_mismatches(0, 1, 0, 0x3f800000, 9973) + L.go_glibc_mismatches(1, 1, 0x80000000, 0xff800000, 9973) == 0
    if not fused_here:
        pytest.skip("this host's libm is not the glibc 2.35 -mfma build; the committed vectors above are the pin")
    assert L.go_glibc_mismatches(0, 1, 0, 0x3f800000, 97) == 0
    assert L.go_glibc_mismatches(1, 1, 0x80000000, 0xff800000, 97) == 0


def test_glibc_math_mode_gives_the_libm_chain(oracle, gist):
    """the restated math is what makes the difference on seed 123 (the portable log flips an accept test near iteration 370)"""
    kw = REFERENCE_PRINTED["dense"]["kw"]
    a = oracle.run(gist, math_mode=oracle.MATH_GLIBC_FMA, **kw)
    check_reference_printed(a, REFERENCE_PRINTED["dense"])
    b = oracle.run(gist, math_mode=oracle.MATH_PORTABLE, **kw)
    assert b["atomsA"].tolist()[:12] == a["atomsA"].tolist()[:12] and b["totalUpdates"] != a["totalUpdates"]


@pytest.mark.parametrize("name", ["modsim", "gist"])
def test_reference_fingerprint(oracle, gist, modsim, name):
    fp = FINGERPRINTS[name]
    data = gist if name == "gist" else modsim
    r = oracle.run(data, nPatterns=fp["k"], nIterations=1000, seed=42, outputFrequency=100)
    assert r["atomsA"].tolist() == fp["atomsA"]
    assert r["atomsP"].tolist() == fp["atomsP"]
    assert r["totalUpdates"] == fp["totalUpdates"]
    assert abs(r["meanChiSq"] - fp["meanChiSq"]) < 5e-4 * fp["meanChiSq"]
    assert abs(r["averageQueueLengthA"] - fp["qA"]) < 0.06 and abs(r["averageQueueLengthP"] - fp["qP"]) < 0.06
    g = np.load(os.path.join(GOLDEN, "%s_k%d_s42_i1000_seq.npz" % (name, fp["k"])))
    for f in ("Amean", "Pmean", "Asd", "Psd", "chisq"):
        assert np.array_equal(r[f], g[f]), f


def test_openmp_queue_loop_is_deterministic(oracle, gist):
    """reference tests/testthat/test_seed_consistency.R:41-70: the result does not depend on nThreads"""
    a = oracle.run(gist, nPatterns=7, nIterations=60, seed=42, outputFrequency=10)
    b = oracle.run(gist, nPatterns=7, nIterations=60, seed=42, outputFrequency=10, omp=True, maxThreads=3)
    for f in ("atomsA", "atomsP", "Amean", "Pmean", "Asd", "Psd", "chisq"):
        assert np.array_equal(a[f], b[f]), f
    assert a["totalUpdates"] == b["totalUpdates"]


def test_same_seed_same_result_other_seed_differs(oracle, modsim):
    """test_seed_consistency.R:23-39"""
    a = oracle.run(modsim, nPatterns=3, nIterations=100, seed=5, outputFrequency=50)
    b = oracle.run(modsim, nPatterns=3, nIterations=100, seed=5, outputFrequency=50)
    c = oracle.run(modsim, nPatterns=3, nIterations=100, seed=6, outputFrequency=50)
    assert np.array_equal(a["Amean"], b["Amean"]) and np.array_equal(a["Pmean"], b["Pmean"])
    assert not np.array_equal(a["Amean"], c["Amean"])


def test_portable_math_matches_libm_chain(oracle, gist):
    """the portable log/exp only changes last-bit ties: same chain on the reference data"""
    a = oracle.run(gist, nPatterns=7, nIterations=150, seed=42, outputFrequency=50)
    b = oracle.run(gist, nPatterns=7, nIterations=150, seed=42, outputFrequency=50, math_mode=oracle.MATH_PORTABLE)
    assert a["atomsA"].tolist() == b["atomsA"].tolist() and a["totalUpdates"] == b["totalUpdates"]


def test_chisq_contract(oracle, modsim):
    """tests/testthat/test_chisq.R: meanChiSq == sum(((D - Amean Pmean^T)/S)^2)"""
    r = oracle.run(modsim, nPatterns=3, nIterations=200, seed=1, outputFrequency=100)
    S = np.maximum(modsim * 0.1, 0.1)
    ref = float((((modsim - r["Amean"].astype(np.float64) @ r["Pmean"].astype(np.float64).T) / S) ** 2).sum())
    assert abs(r["meanChiSq"] - ref) < 1e-4 * ref + 1e-7 * modsim.size


def test_fixed_matrix_contract(oracle, modsim):
    """tests/testthat/test_fixed_matrix.R: the fixed side's mean is all zero, meanChiSq is 0, chisq history moves"""
    fixedP = np.abs(np.random.default_rng(0).normal(size=(20, 3))).astype(np.float32)
    r = oracle.run(modsim, nPatterns=3, nIterations=100, seed=2, outputFrequency=10, whichMatrixFixed="P", fixedPatterns=fixedP)
    assert not r["Pmean"].any() and r["Amean"].any() and r["meanChiSq"] == 0.0
    assert len(set(r["chisq"].tolist())) == len(r["chisq"])


def test_lookup_tables(oracle):
    """src/cpp_tests/testRandom.cpp:54-86 design: LUT error bound 0.03; monotone"""
    from math import erf
    e, ei, qg = oracle.luts()
    assert e.shape == (3001,) and ei.shape == (5001,) and qg.shape == (5001,)
    xs = np.arange(3001) / 1000.0
    assert np.max(np.abs(e - np.array([erf(x) for x in xs]))) < 1e-6
    assert np.all(np.diff(e) >= 0) and np.all(np.diff(ei[:-1]) > 0) and np.all(np.diff(qg[1:-1]) > 0) and qg[-1] == qg[-2]
    assert e[0] == 0.0 and ei[0] == 0.0 and qg[0] == 0.0 and e[-1] < 1.0


def test_portable_log_exp_accuracy(oracle):
    L = oracle.lib()
    rng = np.random.default_rng(3)
    u = (rng.integers(1, 2 ** 32, 20000).astype(np.float32) / np.float32(4294967296.0))
    for x in u:
        assert L.go_portable_logf(float(x)) == np.float32(np.log(np.float64(x)))
    for x in -rng.random(5000).astype(np.float32) * 60:
        assert L.go_portable_expf(float(x)) == np.float32(np.exp(np.float64(x)))
    assert L.go_portable_logf(0.0) == -np.inf and L.go_portable_logf(1.0) == 0.0
