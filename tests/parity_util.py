"""Shared step-wise comparison of a library session (HIP, or the test-only emulator build) with the
oracle: per-batch proposal traces (type, rows, columns, positions, PCG states, atom indices), atom
vectors (position, mass, neighbour links in vector order), factor matrices, AP caches, chi2,
average queue length -- all bit-exact."""
import numpy as np

from cogaps_amd import _capi
import pyoracle as po


def make_pair(lib, data, **kw):
    S = _capi.Session(data, lib=lib, **kw)
    wA = lib.cogaps_reduction_width(S.dims("A")[1])
    wP = lib.cogaps_reduction_width(S.dims("P")[1])
    okw = dict(kw)
    okw.pop("device", None)
    if okw.pop("reductionMode", "lanes") == "seq":
        # verification mode: the reference's own order (one accumulator) and the session's math mode
        math = {"portable": po.MATH_PORTABLE, "glibc-fma": po.MATH_GLIBC_FMA, "glibc-sse2": po.MATH_GLIBC_SSE2}[okw.pop("mathMode", "portable")]
        O = po.Session(data, math_mode=math, redW_A=1, redW_P=1, redG=1, **okw)
    else:
        O = po.Session(data, math_mode=po.MATH_PORTABLE, redW_A=wA, redW_P=wP, redG=4, **okw)
    return S, O


def assert_trace_equal(a, b, tag):
    assert np.array_equal(a["nproc"], b["nproc"]), tag + ": batch sizes (nProcessed) differ"
    assert np.array_equal(a["qlen"], b["qlen"]), tag + ": queue lengths differ"
    assert len(a["rec"]) == len(b["rec"]), tag + ": number of queued proposals differs"
    ra, rb = a["rec"], b["rec"]
    for f in ("type", "r1", "c1", "r2", "c2", "rng_state", "atom1", "batch"):
        assert np.array_equal(ra[f], rb[f]), "%s: field %s differs first at %d" % (tag, f, int(np.nonzero(ra[f] != rb[f])[0][0]))
    m = ra["type"] == ord("M")
    assert np.array_equal(ra["pos"][m], rb["pos"][m]), tag + ": move destinations differ"
    e = ra["type"] == ord("E")
    assert np.array_equal(ra["atom2"][e], rb["atom2"][e]), tag + ": exchange partners differ"


def assert_state_equal(S, O, tag, chisq=True):
    for w in "AP":
        a, b = S.atoms(w), O.atoms(w)
        for f in ("pos", "mass", "left", "right"):
            assert np.array_equal(a[f], b[f]), "%s %s: atom %s differs" % (tag, w, f)
        assert np.array_equal(S.matrix(w), O.matrix(w)), "%s %s: factor matrix differs" % (tag, w)
        assert np.array_equal(S.rows(w), O.rows(w)), "%s %s: factor matrix (HybridMatrix row copy) differs" % (tag, w)
        assert np.array_equal(S.ap(w), O.ap(w)), "%s %s: AP cache differs" % (tag, w)
        assert S.avg_queue(w) == O.avg_queue(w), "%s %s: average queue length differs" % (tag, w)
        assert S.check_domain(w) == 0, "%s %s: the atomic domain's cached neighbour positions / masses or links are inconsistent" % (tag, w)
        if chisq:
            assert S.chisq(w) == O.chisq(w), "%s %s: chi2 differs" % (tag, w)


def run_stepwise(lib, data, n_iter, trace=True, total_iter=None, check_every=1, **kw):
    total_iter = total_iter or max(n_iter, 2)
    kw.setdefault("nIterations", total_iter)
    S, O = make_pair(lib, data, **kw)
    fixed = kw.get("whichMatrixFixed", "N")
    props = 0
    for it in range(n_iter):
        t = min(1.0, 2.0 * it / total_iter)
        S.set_annealing(t), O.set_annealing(t)
        nA, nP = S.draw_steps()
        assert (nA, nP) == O.draw_steps(), "Poisson step counts differ at iteration %d" % it
        props += nA + nP
        if trace and fixed == "N":
            assert_trace_equal(S.update("A", nA, 1 << 16), O.update("A", nA, 1 << 16), "it%d A" % it)
            S.sync("P"), O.sync("P")
            assert_trace_equal(S.update("P", nP, 1 << 16), O.update("P", nP, 1 << 16), "it%d P" % it)
            S.sync("A"), O.sync("A")
        else:
            S.iterate(nA, nP), O.iterate(nA, nP)
        if (it + 1) % check_every == 0 or it == n_iter - 1:
            assert_state_equal(S, O, "it%d" % it)
    out = (S.natoms("A"), S.natoms("P"), props)
    S.close(), O.close()
    return out


def synthetic(genes, samples, rank=3, seed=7):
    rng = np.random.default_rng(seed)
    a0 = rng.gamma(2.0, 0.5, (genes, rank)) * (rng.random((genes, rank)) > 0.5)
    p0 = rng.gamma(2.0, 0.5, (samples, rank)) * (rng.random((samples, rank)) > 0.3)
    return ((a0 @ p0.T) * (0.9 + 0.2 * rng.random((genes, samples))) + 0.01).astype(np.float32)


def synthetic_counts(genes, samples, zeros=0.85, rank=4, seed=7):
    """count-like data (positive entries >= 1, `zeros` of the entries 0): the regime the sparse model's fixed
    uncertainty (0.1 on zeros, 0.1*d elsewhere) coincides with the default max(0.1*d, 0.1)"""
    rng = np.random.default_rng(seed)
    a0 = rng.gamma(2.0, 0.5, (genes, rank)) * (rng.random((genes, rank)) > 0.5)
    p0 = rng.gamma(2.0, 0.5, (samples, rank)) * (rng.random((samples, rank)) > 0.4)
    d = np.ceil((a0 @ p0.T) * (0.9 + 0.2 * rng.random((genes, samples)))) * (rng.random((genes, samples)) > zeros)
    return d.astype(np.float32)


def option_cases():
    """combinations of the run options cogaps_cpp forwards (Cogaps.cpp:64-139): transposed input, a subset in either dimension, a fixed
    factor, an uncertainty matrix, the sparse model, nPatterns from 1 up"""
    cases, i = [], 0
    for sparse in (False, True):
        for transpose in (False, True):
            for fixed in ("N", "A", "P"):
                for subset in (0, 1, 2):
                    i += 1
                    cases.append((sparse, transpose, fixed, subset, 1 + (i % 6), bool(i % 2) and not sparse))
    return cases


def run_option_case(lib, sparse, transpose, fixed, subset, k, with_unc):
    genes, samples = 83, 37
    base = synthetic_counts(genes, samples, zeros=0.6, seed=5 + k) if sparse else synthetic(genes, samples, seed=5 + k)
    data = np.ascontiguousarray(base.T) if transpose else base          # transposeData: the file holds samples x genes
    kw = dict(nPatterns=k, seed=100 + k, total_iter=40, check_every=5, transposeData=transpose, sparseOptimization=sparse)
    n_genes, n_samples = genes, samples
    if subset == 1:
        kw.update(subsetIndices=np.arange(3, 3 + 50, dtype=np.uint32), subsetDim=1); n_genes = 50
    elif subset == 2:
        kw.update(subsetIndices=np.arange(2, 2 + 20, dtype=np.uint32), subsetDim=2); n_samples = 20
    if fixed != "N":
        rows = n_genes if fixed == "A" else n_samples
        kw.update(whichMatrixFixed=fixed, fixedPatterns=np.abs(np.random.default_rng(k).normal(size=(rows, k))).astype(np.float32))
    if with_unc:
        kw.update(unc=np.maximum(data * np.float32(0.2), np.float32(0.3)).astype(np.float32))
    run_stepwise(lib, data, 40, trace=(fixed == "N"), **kw)
