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


# ---- the oracle against the reference BUILD on random configurations (round 4) ----------------------------------------------------
# tools/refprobe compiles the reference's core where it lies (/root/reference/src) with this repository's stand-ins for the Boost
# headers the image lacks, into /tmp -- a probe, not oracle/_ref (parity stays "partial" by rule).  Every number
# the probe prints (%.9g: atom / chi2 histories, totalUpdates, meanChiSq, queue lengths, three rows of each statistic) and an FNV hash of
# each full statistics matrix are compared with pyoracle.run in the reference's arithmetic (sequential sums, libm) bit for bit.
# Skipped where the reference sources or g++ are absent (the GPU box).
N_RANDOM_REFERENCE_CONFIGS = 40


def _refprobe():
    sys_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "refprobe")
    import sys
    if sys_path not in sys.path:
        sys.path.insert(0, sys_path)
    import refprobe
    return refprobe


@pytest.fixture(scope="session")
def refprobe_bin():
    rp = _refprobe()
    if not rp.available():
        pytest.skip("no /root/reference sources or no g++ here: the reference-build probe runs in the build container only")
    return rp.build()


def random_reference_config(i):
    """configuration number i (seeded): data, file format, and both parameter sets (probe arguments / pyoracle keywords)"""
    g = np.random.Generator(np.random.MT19937(20240 + i))
    fmt = (".mtx", ".csv", ".tsv", ".gct")[i % 4]
    sparse = bool(i % 3 == 1)
    nG, nS = int(g.integers(24, 150)), int(g.integers(6, 40))
    rank = int(g.integers(2, 5))
    d = (g.gamma(2.0, 0.5, (nG, rank)) * (g.random((nG, rank)) > 0.4)) @ (g.gamma(2.0, 0.5, (rank, nS)) * (g.random((rank, nS)) > 0.3))
    d = d * (0.8 + 0.4 * g.random((nG, nS))) + 0.01
    if sparse:
        d = d * (g.random((nG, nS)) >= g.uniform(0.6, 0.95))
    d = np.ascontiguousarray(d, dtype=np.float32)
    KS = [1, 2, 3, 4, 5, 7, 11, 13, 24, 25, 26, 33, 50, 60]          # (25 / 26: both directions of gaps::dot, VectorMath.h:41-134)
    K = KS[i % len(KS)] if i < 2 * len(KS) else int(g.choice(KS))
    nIter = int(g.integers(20, 110))
    outFreq = max(1, nIter // int(g.integers(2, 11)))
    cfg = dict(fmt=fmt, sparse=sparse, K=K, nIter=nIter, outFreq=outFreq, seed=int(g.integers(1, 2 ** 31 - 1)), threads=int(g.integers(1, 5)),
               transpose=bool(g.random() < 0.3), subsetDim=0, subset=None, unc=None, fixed="N", fixedPatterns=None,
               pump=bool(g.random() < 0.2), snaps=0, snapPhase=3, alphaA=0.01, alphaP=0.01, maxA=100.0, maxP=100.0)
    genes, samples = nG, nS
    if g.random() < 0.4:
        cfg["subsetDim"] = int(g.integers(1, 3))
        n = genes if cfg["subsetDim"] == 1 else samples
        keep = max(4, int(n * g.uniform(0.4, 0.9)))
        cfg["subset"] = np.sort(g.choice(n, size=keep, replace=False)).astype(np.uint32) + 1       # 1-based, sorted (Matrix.cpp:112 sorts them for file inputs)
        if cfg["subsetDim"] == 1:
            genes = keep
        else:
            samples = keep
    if not sparse and g.random() < 0.3:
        cfg["unc"] = np.maximum(0.2 * d, 0.3).astype(np.float32)
    if g.random() < 0.25:
        cfg["fixed"] = "A" if g.random() < 0.5 else "P"
        rows = genes if cfg["fixed"] == "A" else samples
        cfg["fixedPatterns"] = (g.gamma(2.0, 0.5, (rows, K)) * (g.random((rows, K)) > 0.2)).astype(np.float32)
    if g.random() < 0.2:
        cfg["snaps"] = int(g.integers(2, 5)); cfg["snapPhase"] = int(g.integers(1, 4))
    if g.random() < 0.25:
        cfg["alphaA"], cfg["alphaP"] = float(np.float32(g.uniform(0.005, 0.5))), float(np.float32(g.uniform(0.005, 0.5)))
        cfg["maxA"], cfg["maxP"] = float(np.float32(g.uniform(5, 200))), float(np.float32(g.uniform(5, 200)))
    cfg["data"] = d
    return cfg


def run_reference_config(rp, binary, cfg, tmp):
    stored = np.ascontiguousarray(cfg["data"].T) if cfg["transpose"] else cfg["data"]
    path = os.path.join(tmp, "d" + cfg["fmt"])
    rp.write_matrix(path, stored)
    kw = dict(nPatterns=cfg["K"], nIterations=cfg["nIter"], seed=cfg["seed"], threads=cfg["threads"], outFreq=cfg["outFreq"], sparse=cfg["sparse"],
              transpose=cfg["transpose"], pump=cfg["pump"], alphaA=repr(cfg["alphaA"]), alphaP=repr(cfg["alphaP"]), maxGibbsA=repr(cfg["maxA"]), maxGibbsP=repr(cfg["maxP"]))
    extra = {}
    if cfg["subsetDim"]:
        sp = os.path.join(tmp, "subset.txt")
        np.savetxt(sp, cfg["subset"], fmt="%d")
        extra.update(subsetDim=cfg["subsetDim"], subset=sp)
    if cfg["unc"] is not None:
        up = os.path.join(tmp, "u" + cfg["fmt"])
        rp.write_matrix(up, np.ascontiguousarray(cfg["unc"].T) if cfg["transpose"] else cfg["unc"])
        extra["unc"] = up
    if cfg["fixed"] != "N":
        fpth = os.path.join(tmp, "fixed.csv")
        rp.write_matrix(fpth, cfg["fixedPatterns"])
        extra.update(fixed=cfg["fixed"], fixedFile=fpth)
    if cfg["snaps"]:
        kw.update(snapshots=cfg["snaps"], snapshotPhase=cfg["snapPhase"])
    return rp.run(binary, path, timeout=600, **extra, **kw), stored


def oracle_for_config(oracle, cfg, stored):
    unc = None
    if cfg["unc"] is not None:
        unc = np.ascontiguousarray(cfg["unc"].T) if cfg["transpose"] else cfg["unc"]
    return oracle.run(stored, unc=unc, omp=cfg["threads"] > 1, nPatterns=cfg["K"], nIterations=cfg["nIter"], seed=cfg["seed"], outputFrequency=cfg["outFreq"],
                      maxThreads=cfg["threads"], alphaA=cfg["alphaA"], alphaP=cfg["alphaP"], maxGibbsMassA=cfg["maxA"], maxGibbsMassP=cfg["maxP"],
                      transposeData=cfg["transpose"], subsetIndices=cfg["subset"], subsetDim=cfg["subsetDim"], whichMatrixFixed=cfg["fixed"],
                      fixedPatterns=cfg["fixedPatterns"], sparseOptimization=cfg["sparse"], takePumpSamples=cfg["pump"],
                      snapshotFrequency=(cfg["nIter"] // cfg["snaps"]) if cfg["snaps"] else 0, snapshotPhase=(cfg["snapPhase"] % 3) if cfg["snaps"] else 0)      # (the oracle numbers "all phases" 0, the reference GAPS_ALL_PHASES = 3)


def compare_with_reference_build(rp, ref, o):
    """every printed value, bit for bit (the probe prints %.9g, which identifies an fp32 value)"""
    assert ref["atomsA"].tolist() == o["atomsA"].tolist() and ref["atomsP"].tolist() == o["atomsP"].tolist()
    assert ref["totalUpdates"] == o["totalUpdates"]
    assert np.array_equal(ref["chisq"], o["chisq"].astype(np.float32))
    for key, okey in (("meanChiSq", "meanChiSq"), ("qA", "averageQueueLengthA"), ("qP", "averageQueueLengthP")):
        assert ref[key] == np.float32(o[okey]), (key, ref[key], o[okey])
    for name in ("Amean", "Asd", "Pmean", "Psd"):
        for (n, r), vals in ref["rows"].items():
            if n == name:
                assert np.array_equal(vals, o[name][r]), (name, r)
        h, cnt = ref["hashes"][name]
        assert cnt == o[name].size and h == rp.fnv_matrix(o[name]), name
    if "pump" in ref["hashes"]:
        assert ref["hashes"]["pump"][0] == rp.fnv_matrix(o["pumpMatrix"]) and ref["hashes"]["meanPattern"][0] == rp.fnv_matrix(o["meanPatternAssignment"])
    assert len(ref["snapE"]) == o["equilibrationSnapshotsA"].shape[0] and len(ref["snapS"]) == o["samplingSnapshotsA"].shape[0]
    for k, (ha, hp) in enumerate(ref["snapE"]):
        assert ha == rp.fnv_matrix(o["equilibrationSnapshotsA"][k]) and hp == rp.fnv_matrix(o["equilibrationSnapshotsP"][k])
    for k, (ha, hp) in enumerate(ref["snapS"]):
        assert ha == rp.fnv_matrix(o["samplingSnapshotsA"][k]) and hp == rp.fnv_matrix(o["samplingSnapshotsP"][k])


@pytest.mark.parametrize("i", range(N_RANDOM_REFERENCE_CONFIGS))
def test_oracle_equals_reference_build_on_random_configs(oracle, refprobe_bin, tmp_path, i):
    rp = _refprobe()
    cfg = random_reference_config(i)
    ref, stored = run_reference_config(rp, refprobe_bin, cfg, str(tmp_path))
    compare_with_reference_build(rp, ref, oracle_for_config(oracle, cfg, stored))


# ---- VERDICT.md round 3, "Judge's own checks": seven more configurations of the reference build (the judge's own stand-in headers and
# driver), every printed float given to nine digits.  1-6 run here and on the GPU in the verification mode; 7 -- BASELINE configs[2]
# whole, 100 + 100 iterations, 53 M proposals, four minutes of eight cores -- only with COGAPS_LONG_TESTS=1 (tools/ref_vs_port_c3.py makes the
# same comparison against the probe and keeps its record under profiles/).  Regenerate / extend: tools/refprobe/regen_fingerprints.py.
def _judge3_data(name, gist):
    import bench
    if name in ("gist_gct_k9", "gist_tsv_sparse_k11_samples", "gist_csv_sparse_k4_genes"):
        return gist, None
    if name == "synth1500_k50_unc":
        d = bench.synthetic_dense(1500, 350)
        return d, np.maximum(np.float32(0.2) * d, np.float32(0.3)).astype(np.float32)
    if name == "synth2500_sparse_k50":
        d = bench.synthetic_dense(2500, 600)
        return np.ascontiguousarray(d * (np.random.Generator(np.random.MT19937(99)).random(d.shape) >= 0.95), dtype=np.float32), None
    return bench.synthetic_dense(20000, 2000), None


JUDGE_R3 = {
    "gist_gct_k9": dict(kw=dict(nPatterns=9, nIterations=180, seed=777, outputFrequency=18),
                        atomsA=[392, 1272, 2109, 2892, 3635, 4013, 4330, 4625, 4901, 5051, 5116, 5068, 5045, 4948, 4866, 4723, 4625, 4511, 4465, 4429],
                        atomsP=[12, 17, 26, 33, 36, 43, 46, 53, 55, 58, 57, 60, 61, 64, 66, 64, 67, 70, 70, 68],
                        totalUpdates=1433511, meanChiSq=5047.65088, qA=36.055397, qP=2.97920275),
    "gist_tsv_sparse_k11_samples": dict(kw=dict(nPatterns=11, nIterations=120, seed=5, outputFrequency=12, sparseOptimization=True, subsetIndices=np.arange(2, 9, dtype=np.uint32), subsetDim=2),
                                        atomsA=[250, 881, 1629, 2313, 2914, 3304, 3440, 3488, 3556, 3639, 3670, 3710, 3729, 3805, 3820, 3846, 3844, 3850, 3935, 3902],
                                        atomsP=[7, 12, 17, 20, 24, 27, 31, 33, 34, 36, 38, 38, 40, 40, 40, 41, 44, 49, 53, 50],
                                        totalUpdates=745027, meanChiSq=3149.55542, qA=34.574707, qP=2.4916687),
    "synth1500_k50_unc": dict(kw=dict(nPatterns=50, nIterations=20, seed=7, outputFrequency=2),
                              atomsA=[13, 21, 50, 107, 210, 415, 692, 1064, 1477, 1978, 2509, 3054, 3622, 4229, 4796, 5354, 5996, 6589, 7200, 7818],
                              atomsP=[9, 22, 40, 73, 135, 198, 287, 392, 512, 620, 735, 873, 1016, 1150, 1276, 1443, 1579, 1735, 1887, 2040],
                              totalUpdates=131487, meanChiSq=5212348.0, qA=35.5075798, qP=17.6584644),
    "gist_csv_sparse_k4_genes": dict(kw=dict(nPatterns=4, nIterations=150, seed=1234, outputFrequency=15, sparseOptimization=True, subsetIndices=np.arange(200, 901, dtype=np.uint32), subsetDim=1),
                                     atomsA=[122, 293, 472, 647, 826, 1008, 1193, 1356, 1458, 1564, 1643, 1740, 1785, 1801, 1875, 1898, 1941, 1967, 1976, 2025],
                                     atomsP=[6, 9, 11, 12, 15, 17, 17, 19, 19, 21, 21, 21, 21, 22, 23, 25, 27, 26, 28, 32],
                                     totalUpdates=401805, meanChiSq=16062.7295, qA=23.7872849, qP=2.56697607),
    "synth2500_sparse_k50": dict(kw=dict(nPatterns=50, nIterations=15, seed=3, outputFrequency=3, sparseOptimization=True),
                                 atomsA=[18, 54, 149, 371, 747, 1328, 2057, 2928, 3909, 4890], atomsP=[12, 27, 75, 162, 300, 480, 702, 963, 1227, 1514],
                                 totalUpdates=52553, meanChiSq=4371113.5, qA=36.8032532, qP=18.0268326),
    "headline_18": dict(kw=dict(nPatterns=50, nIterations=18, seed=42, outputFrequency=3),
                        atomsA=[22, 69, 246, 802, 2384, 5938, 12012, 20386, 30052, 40759, 51989, 63811], atomsP=[24, 65, 197, 586, 1168, 2020, 3047, 4212, 5406, 6596, 7868, 9130],
                        totalUpdates=658260, meanChiSq=2147591936.0, qA=None, qP=None),      # (the fp32 chi2 accumulator saturating at 2^31 is the reference's own behaviour)
    "headline_100": dict(kw=dict(nPatterns=50, nIterations=100, seed=42, outputFrequency=10),
                         atomsA=[376, 9731, 40869, 80358, 121928, 163065, 201598, 235212, 264287, 289635, 312314, 331845, 347230, 358467, 365076, 368751, 370696, 371881, 371160, 370151],
                         atomsP=None, totalUpdates=53070346, meanChiSq=64217980.0, qA=157.304565, qP=50.5546989),
}


def check_judge3_case(r, fp):
    """nine printed digits identify an fp32 value: equality of the floats"""
    assert r["atomsA"].tolist() == fp["atomsA"]
    if fp["atomsP"] is not None:
        assert r["atomsP"].tolist() == fp["atomsP"]
    assert r["totalUpdates"] == fp["totalUpdates"]
    assert np.float32(r["meanChiSq"]) == np.float32(fp["meanChiSq"]), r["meanChiSq"]
    if fp["qA"] is not None:
        assert np.float32(r["averageQueueLengthA"]) == np.float32(fp["qA"]) and np.float32(r["averageQueueLengthP"]) == np.float32(fp["qP"])


def run_judge3_case(run, name, gist):
    d, unc = _judge3_data(name, gist)
    return run(d, unc=unc, **JUDGE_R3[name]["kw"])


@pytest.mark.parametrize("name", list(JUDGE_R3))
def test_judge_round3_fingerprints(oracle, gist, name):
    if name == "headline_100" and os.environ.get("COGAPS_LONG_TESTS") != "1":
        pytest.skip("four minutes of eight cores: COGAPS_LONG_TESTS=1 (tools/ref_vs_port_c3.py keeps the record of the same comparison)")
    big = name.startswith("headline")
    r = run_judge3_case(lambda d, unc=None, **kw: oracle.run(d, unc=unc, omp=big, maxThreads=8 if big else 1, **kw), name, gist)
    check_judge3_case(r, JUDGE_R3[name])


def test_typed_fingerprints_regenerate_from_the_reference_build(refprobe_bin, tmp_path):
    """the reference-printed tables above (judge rounds 1-3) are not just typed in: tools/refprobe/regen_fingerprints.py
    re-runs the reference build on each configuration (the K = 50 and headline-sized ones only with --all) and finds the same numbers"""
    rp = _refprobe()
    import regen_fingerprints as rg
    cs = rg.cases(str(tmp_path), skip=rg.BIG)
    n = 0
    for name, (data, kw, fp) in cs.items():
        assert rg.check(rp.run(refprobe_bin, data, **kw), fp), name
        n += 1
    assert n >= 13
