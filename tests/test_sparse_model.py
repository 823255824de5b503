"""SparseNormalModel (useSparseOptimization): the oracle's restatement against the dense model and against known
answers, and the kernels (test-only emulator build) against the oracle, bit for bit.  The reference holds no
compiled test of this model; its stale Catch design (src/cpp_tests/testSparseGibbsSampler.cpp:41-249: sparse and
dense alpha parameters agree within 0.1 %, incl. 2-site symmetry and with-change) is re-created here."""
import numpy as np
import pytest

import parity_util as pu


def _pair(oracle, data, k, **kw):
    od = oracle.Session(data, nPatterns=k, seed=1, **kw)
    os_ = oracle.Session(data, nPatterns=k, seed=1, sparseOptimization=True, **kw)
    return od, os_


def test_initial_chisq_is_100_per_nonzero(oracle):
    """all-zero factors: chi2 = sum (d / (0.1 d))^2 = 100 * nnz in both models (the reference's dense design checks
    100 * nRow * nCol for D(i,j) = i+j+1, testDenseGibbsSampler.cpp:26-35)"""
    data = pu.synthetic_counts(120, 30, zeros=0.7)
    od, os_ = _pair(oracle, data, 4)
    nnz = int((data > 0).sum())
    for w in "AP":
        assert os_.chisq(w) == 100.0 * nnz
        assert od.chisq(w) == pytest.approx(100.0 * nnz, rel=1e-5)
    d_full = (np.arange(20)[:, None] + np.arange(7)[None, :] + 1).astype(np.float32)
    od, os_ = _pair(oracle, d_full, 3)
    assert od.chisq("A") == 100.0 * 20 * 7 and os_.chisq("A") == 100.0 * 20 * 7


def test_sparse_and_dense_alpha_parameters_agree(oracle):
    rng = np.random.default_rng(5)
    g, s, k = 300, 40, 5
    data = pu.synthetic_counts(g, s, zeros=0.7, seed=3)
    a = (rng.gamma(2, .5, (g, k)) * (rng.random((g, k)) > .5)).astype(np.float32)
    p = (rng.gamma(2, .5, (s, k)) * (rng.random((s, k)) > .3)).astype(np.float32)
    od, os_ = _pair(oracle, data, k)
    od.debug_set_matrices(a, p), os_.debug_set_matrices(a, p)
    worst = 0.0
    for w, m in (("A", g), ("P", s)):
        for _ in range(200):
            r1, r2 = (int(x) for x in rng.integers(0, m, 2))
            c1, c2 = (int(x) for x in rng.integers(0, k, 2))
            for mode, args in ((0, (r1, c1)), (1, (r1, c1, 0, 0, -0.7)), (2, (r1, c1, r1, c2)), (2, (r1, c1, r2, c2))):
                x, y = od.debug_alpha(w, mode, *args), os_.debug_alpha(w, mode, *args)
                worst = max(worst, max(abs(u - v) / max(1.0, abs(u)) for u, v in zip(x, y)))
            # 2-site symmetry: swapping the sites keeps s and flips the sign of s_mu
            if r1 != r2 or c1 != c2:
                f, b = os_.debug_alpha(w, 2, r1, c1, r2, c2), os_.debug_alpha(w, 2, r2, c2, r1, c1)
                assert f[0] == pytest.approx(b[0], rel=1e-4, abs=1e-2) and f[1] == pytest.approx(-b[1], rel=1e-4, abs=1e-2)
    assert worst < 1e-3
    assert os_.chisq("A") == pytest.approx(od.chisq("A"), rel=1e-4)


def test_sparse_chain_is_thread_count_independent(oracle):
    """test_seed_consistency.R:41-70 (sparse rows): 1 vs 3 threads, equal atom histories"""
    data = pu.synthetic_counts(200, 40, zeros=0.8)
    r1 = oracle.run(data, nPatterns=5, nIterations=60, seed=42, outputFrequency=20, sparseOptimization=True, omp=True, maxThreads=1)
    r3 = oracle.run(data, nPatterns=5, nIterations=60, seed=42, outputFrequency=20, sparseOptimization=True, omp=True, maxThreads=3)
    assert np.array_equal(r1["atomsA"], r3["atomsA"]) and np.array_equal(r1["atomsP"], r3["atomsP"])
    assert r1["totalUpdates"] == r3["totalUpdates"] and np.array_equal(r1["Amean"], r3["Amean"])


@pytest.mark.parametrize("genes,samples,k,iters,zeros,win", [
    (60, 40, 3, 120, 0.85, 256),      # one flag word per vector, K <= 25: gaps::dot adds last-to-first
    (300, 50, 30, 30, 0.85, 256),     # K > 25: first-to-last
    (9000, 12, 4, 10, 0.9, 256),      # 141 flag words: 256-thread workgroups, cross-wave butterfly
    (20, 5000, 3, 10, 0.9, 256),      # the long side on the other sampler
    (200, 70, 7, 40, 0.6, 64),        # 64-attempt windows: hazards and multi-round batches
])
def test_sparse_stepwise(emul_lib, genes, samples, k, iters, zeros, win):
    data = pu.synthetic_counts(genes, samples, zeros=zeros, seed=genes + samples)
    pu.run_stepwise(emul_lib(win), data, iters, trace=genes * samples < 50000, nPatterns=k, seed=11, total_iter=max(iters, 40), sparseOptimization=True)


def test_sparse_transposed_subset_fixed(emul_lib):
    data = pu.synthetic_counts(240, 36, zeros=0.8, seed=9)
    pu.run_stepwise(emul_lib(256), np.ascontiguousarray(data.T), 20, trace=False, nPatterns=4, seed=3, total_iter=40, sparseOptimization=True, transposeData=True)
    idx = np.arange(5, 125, dtype=np.uint32)
    pu.run_stepwise(emul_lib(256), data, 20, trace=False, nPatterns=4, seed=3, total_iter=40, sparseOptimization=True, subsetIndices=idx, subsetDim=1)
    fixed = np.abs(np.random.default_rng(2).normal(0.5, 0.4, (36, 4))).astype(np.float32)
    fixed[fixed < 0.3] = 0.0
    pu.run_stepwise(emul_lib(256), data, 20, trace=False, nPatterns=4, seed=3, total_iter=40, sparseOptimization=True, whichMatrixFixed="P", fixedPatterns=fixed)


def test_sparse_full_run_matches_oracle(emul_lib, oracle):
    from cogaps_amd import _capi
    data = pu.synthetic_counts(150, 30, zeros=0.8, seed=21)
    lib = emul_lib(256)
    kw = dict(nPatterns=4, nIterations=40, seed=42, outputFrequency=10, sparseOptimization=True)
    r = _capi.run(data, lib=lib, **kw)
    w_a, w_p = lib.cogaps_reduction_width(30), lib.cogaps_reduction_width(150)
    o = oracle.run(data, math_mode=oracle.MATH_PORTABLE, redW_A=w_a, redW_P=w_p, redG=4, **kw)
    for f in ("Amean", "Asd", "Pmean", "Psd", "chisq", "atomsA", "atomsP"):
        assert np.array_equal(r[f], o[f]), f
    assert r["totalUpdates"] == o["totalUpdates"] and r["meanChiSq"] == o["meanChiSq"]
