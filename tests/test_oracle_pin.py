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


def test_luts_are_the_correctly_rounded_tables(oracle):
    """The three lookup tables come from double-precision normal / gamma(2,1) cdf and quantile calls (Boost.Math in the reference,
    Math.cpp:55-81 -- a dependency absent from /root/reference and from this image; libm + Newton in the oracle and the product),
    rounded to fp32 (Random.cpp:269-295).  Rebuilt here with mpmath at 50 digits following the same fp32 rounding sequence: every one
    of the 13 003 entries equals the oracle's, and no exact value lies within 1e-13 (relative; ~1000 double ulps) of an fp32 rounding
    midpoint -- so ANY double-accurate implementation, Boost.Math included, rounds to these same fp32 tables.  That closes the
    "LUT bit-parity with a real Boost build" caveat of SURVEY.md section 8c in substance."""
    import mpmath as mp
    mp.mp.dps = 50
    f32 = np.float32
    e, ei, qg = oracle.luts()
    sqrt2f = f32(1.4142135623730951)
    margin = [mp.mpf(1)]

    def to_f32(v):
        """round the exact value to fp32; record its relative distance to the nearer rounding midpoint"""
        r = f32(float(v))                      # (double rounding is harmless exactly when the margin below holds)
        if v != 0:
            lo, hi = np.nextafter(r, f32(-np.inf)), np.nextafter(r, f32(np.inf))
            d = min(abs(v - (mp.mpf(float(r)) + mp.mpf(float(lo))) / 2), abs(v - (mp.mpf(float(r)) + mp.mpf(float(hi))) / 2)) / abs(v)
            margin[0] = min(margin[0], d)
        return r

    def newton(f, df, x0):
        x = mp.mpf(float(x0))
        for _ in range(8):
            x = x - f(x) / df(x)
        return x
    bad = 0
    # erf: 2.f * float(cdf(N(0,1), x * sqrt2f)) - 1.f
    for i in range(3001):
        x = f32(i) / f32(1000.0)
        p = to_f32(mp.ncdf(mp.mpf(float(f32(x * sqrt2f)))))
        bad += int(f32(f32(2.0) * p - f32(1.0)).tobytes() != e[i].tobytes())
    # erfinv: float(quantile(N(0,1), (1.f + x) / 2.f)) / sqrt2f; the last entry from q = 1.9998f / 2.f
    qs = [f32(f32(1.0) + f32(i) / f32(5000.0)) / f32(2.0) for i in range(5000)] + [f32(1.9998) / f32(2.0)]
    for i, q in enumerate(qs):
        qq = mp.mpf(float(q))
        z = mp.mpf(0) if q == f32(0.5) else newton(lambda t: mp.ncdf(t) - qq, mp.npdf, float(ei[i]) * 1.4142135623730951)
        bad += int(f32(to_f32(z) / sqrt2f).tobytes() != ei[i].tobytes())
    # qgamma: float(quantile(Gamma(2,1), x)), 0 below 1e-6; the last entry from 0.9998f
    xs = [f32(i) / f32(5000.0) for i in range(5000)] + [f32(0.9998)]
    for i, x in enumerate(xs):
        if i == 0 or x < f32(0.000001):
            bad += int(qg[i] != 0.0)
            continue
        xx = mp.mpf(float(x))
        z = newton(lambda t: 1 - mp.exp(-t) * (1 + t) - xx, lambda t: t * mp.exp(-t), max(float(qg[i]), 1e-3))
        bad += int(to_f32(z).tobytes() != qg[i].tobytes())
    assert bad == 0
    assert margin[0] >= mp.mpf("1e-13"), margin[0]


# Six further outputs of the reference core ("Judge's own checks": the reference's sources built per SURVEY.md
# Appendix B and run on configurations nobody had tuned anything on).  Among them the only reference-printed numbers for K = 50 --
# forward gaps::dot order, large-lambda Poisson -- dense and sparse (the bench's generator at 4000 x 400), and for the two-round shard
# flow (subset of genes, then whichMatrixFixed = 'P' with the first round's Pmean).  kw: oracle / library keyword arguments.
def _judge_data(name, gist, modsim):
    import bench
    if name in ("gist_tsv_k3", "gist_csv_sparse_k6", "shard_round1", "shard_round2"):
        return gist
    if name == "modsim_sparse_k4":
        return modsim
    d = bench.synthetic_dense(4000, 400)
    if name == "k50_sparse":
        d = d * (np.random.Generator(np.random.MT19937(777)).random(d.shape) >= 0.95)
    return np.ascontiguousarray(d, dtype=np.float32)


JUDGE_R2 = {
    "gist_tsv_k3": dict(kw=dict(nPatterns=3, nIterations=400, seed=2024, outputFrequency=40),
                        atomsA=[517, 1201, 1788, 2225, 2596, 2625, 2512, 2313, 2228, 2208, 2141, 2099, 2160, 2193, 2199, 2200, 2231, 2246, 2290, 2269],
                        atomsP=[11, 17, 21, 25, 24, 25, 25, 32, 36, 37, 39, 44, 46, 47, 45, 47, 48, 46, 49, 51],
                        totalUpdates=1671263, meanChiSq=5399.322, lastChisq=6716.472, qA=31.665, qP=2.670,
                        probes=[("Amean", 0, [1.06566, 0.009782198, 0.002830506]), ("Pmean", 0, [0.9526445, 0.9599464, 1.0])]),
    "modsim_sparse_k4": dict(kw=dict(nPatterns=4, nIterations=500, seed=9, outputFrequency=50, sparseOptimization=True),
                             atomsA=[35, 47, 51, 68, 75, 91, 90, 90, 86, 85, 86, 98, 91, 105, 109, 105, 107, 101, 91, 93],
                             atomsP=[19, 41, 47, 50, 49, 59, 58, 63, 56, 56, 54, 45, 52, 55, 54, 57, 52, 55, 58, 56],
                             totalUpdates=133060, meanChiSq=49.368, lastChisq=114.531, qA=4.645, qP=4.084,
                             probes=[("Amean", 0, [0.3686761, 0.05144705, 11.68852])]),
    "gist_csv_sparse_k6": dict(kw=dict(nPatterns=6, nIterations=250, seed=31, outputFrequency=25, sparseOptimization=True),
                               atomsA=[520, 1358, 2068, 2520, 2747, 2839, 2889, 2949, 3016, 3055, 3090, 3097, 3158, 3155, 3194, 3244, 3245, 3201, 3342, 3289],
                               atomsP=[12, 17, 23, 27, 30, 39, 41, 43, 42, 45, 46, 43, 44, 48, 50, 52, 55, 58, 54, 54],
                               totalUpdates=1375546, meanChiSq=3562.224, lastChisq=7074.692, qA=33.215, qP=2.893, probes=[]),
    "k50_dense": dict(kw=dict(nPatterns=50, nIterations=25, seed=42, outputFrequency=2),
                      atomsA=[17, 36, 69, 164, 360, 697, 1238, 2002, 3023, 4205, 5455, 6826, 9054, 10577, 12037, 13640, 15269, 16881, 18509, 20189, 21820, 23372, 24987, 26625],
                      atomsP=[13, 26, 47, 88, 176, 280, 420, 541, 677, 851, 1018, 1162, 1415, 1575, 1748, 1934, 2117, 2289, 2440, 2594, 2777, 2949, 3110, 3278],
                      totalUpdates=535815, meanChiSq=67113072.0, lastChisq=58949680.0, qA=62.417, qP=20.467, probes=[("Amean", 0, [0.04871542])]),
    "k50_sparse": dict(kw=dict(nPatterns=50, nIterations=30, seed=42, outputFrequency=3, sparseOptimization=True),
                       atomsA=[22, 67, 212, 551, 1144, 2192, 3469, 4836, 6276, 7681, 9169, 10576, 12031, 13404, 14807, 16177, 17423, 18496, 19568, 20607],
                       atomsP=[20, 47, 111, 244, 393, 554, 749, 947, 1170, 1372, 1623, 1884, 2112, 2326, 2548, 2789, 2980, 3219, 3407, 3598],
                       totalUpdates=583624, meanChiSq=4587509.0, lastChisq=5578760.0, qA=61.292, qP=20.624, probes=[("Amean", 0, [None, None, 0.05574835])]),
    "shard_round1": dict(kw=dict(nPatterns=3, nIterations=200, seed=42, outputFrequency=50, subsetIndices=np.arange(1, 401, dtype=np.uint32), subsetDim=1),
                         atomsA=[231, 474, 719, 890, 1042, 980, 950, 902], atomsP=None, totalUpdates=None, meanChiSq=None, lastChisq=None, qA=None, qP=None, probes=[]),
    "shard_round2": dict(kw=dict(nPatterns=3, nIterations=200, seed=42, outputFrequency=50, subsetIndices=np.arange(1, 401, dtype=np.uint32), subsetDim=1, whichMatrixFixed="P"),
                         atomsA=[210, 410, 535, 605, 630, 679, 728, 750], atomsP=[0] * 8, totalUpdates=None, meanChiSq=0.0, lastChisq=None, qA=None, qP=None,
                         probes=[("Pmean", 0, [0.0])]),
}


def check_judge_case(r, fp):
    """every digit the reference binary printed (seven significant digits; chi2 values as printed)"""
    assert r["atomsA"].tolist() == fp["atomsA"]
    if fp["atomsP"] is not None:
        assert r["atomsP"].tolist() == fp["atomsP"]
    if fp["totalUpdates"] is not None:
        assert r["totalUpdates"] == fp["totalUpdates"]
    for got, want in ((r["meanChiSq"], fp["meanChiSq"]), (float(r["chisq"][-1]), fp["lastChisq"])):
        if want is not None:
            assert abs(got - want) <= max(6e-4, 6e-8 * abs(want)), (got, want)
    if fp["qA"] is not None:
        assert abs(r["averageQueueLengthA"] - fp["qA"]) < 6e-4 and abs(r["averageQueueLengthP"] - fp["qP"]) < 6e-4
    for f, row, vals in fp["probes"]:
        for c, v in enumerate(vals):
            if v is not None:
                assert abs(float(r[f][row, c]) - v) <= 6e-7 * abs(v), (f, row, c)


def run_judge_case(run, name, gist, modsim, cache):
    """`run(data, **kw)` -> result dict; the second shard round takes the first round's Pmean as its fixed patterns"""
    fp = JUDGE_R2[name]
    kw = dict(fp["kw"])
    if name == "shard_round2":
        if "shard_round1" not in cache:
            cache["shard_round1"] = run(_judge_data("shard_round1", gist, modsim), **JUDGE_R2["shard_round1"]["kw"])
        kw["fixedPatterns"] = cache["shard_round1"]["Pmean"]
    r = run(_judge_data(name, gist, modsim), **kw)
    cache[name] = r
    return r


@pytest.mark.parametrize("name", list(JUDGE_R2))
def test_judge_round2_fingerprints(oracle, gist, modsim, name, _judge_cache={}):
    r = run_judge_case(oracle.run, name, gist, modsim, _judge_cache)
    check_judge_case(r, JUDGE_R2[name])
