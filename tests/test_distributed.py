"""GWCoGAPS / scCoGAPS driver: consensus matching on hand-built cases (the reference's R clustering is
unpinned -- SURVEY.md H7 -- so the checks are structural) and the world_size-2 data path over gloo with
the per-subset chains run on the test-only emulator build."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _params(n_patterns, n_sets, **kw):
    from cogaps_amd import CogapsParams
    p = CogapsParams(nPatterns=n_patterns, seed=3, **kw)
    p.distributed = "genome-wide"
    p.setDistributedParams(nSets=n_sets)
    return p


def test_complete_linkage_cutree_known_answer():
    from cogaps_amd.distributed import _complete_linkage_cutree
    pts = np.array([0.0, 0.1, 0.2, 5.0, 5.1, 9.0])
    d = np.abs(pts[:, None] - pts[None, :])
    assert _complete_linkage_cutree(d, 3).tolist() == [1, 1, 1, 2, 2, 3]
    assert _complete_linkage_cutree(d, 2).tolist() == [1, 1, 1, 2, 2, 2]
    assert _complete_linkage_cutree(d, 6).tolist() == [1, 2, 3, 4, 5, 6]


def test_pattern_match_recovers_planted_patterns():
    from cogaps_amd.distributed import find_consensus_matrix
    rng = np.random.default_rng(0)
    base = np.abs(rng.normal(size=(60, 4)))
    sets = [base[:, rng.permutation(4)] * (1 + 0.03 * rng.random((60, 4))) for _ in range(4)]
    r = find_consensus_matrix(sets, _params(4, 4))
    cons = r["consensus"]
    assert cons.shape == (60, 4) and np.allclose(cons.max(axis=0), 1.0)
    assert all(2 <= c.shape[1] <= 6 for c in r["clusteredPatterns"])          # minNS <= size <= maxNS
    corr = np.corrcoef(cons.T, (base / base.max(axis=0)).T)[:4, 4:]
    assert (np.abs(corr).max(axis=1) > 0.99).all() and len(set(np.abs(corr).argmax(axis=1).tolist())) == 4


def test_create_sets_cover_and_explicit():
    from cogaps_amd.distributed import create_sets
    p = _params(3, 4)
    sets = create_sets(103, p)
    allv = np.concatenate(sets)
    assert sorted(allv.tolist()) == list(range(1, 104)) and [len(s) for s in sets] == [25, 25, 25, 28]
    p.explicitSets = [list(range(1, 11)), list(range(11, 21)), list(range(21, 31)), list(range(31, 41))]
    assert [s.tolist() for s in create_sets(40, p)] == p.explicitSets      # tests/testthat/test_subset_data.R:27-40


def test_oversized_clusters_are_split_until_none_is_left():
    """R/DistributedCogaps.R:160-166: splitCluster is called until no cluster exceeds maxNS, also when a split keeps one half
    only (the other falls below minNS) and that half is still too large"""
    from cogaps_amd.distributed import pattern_match
    rng = np.random.default_rng(5)
    t = np.linspace(0, 1, 80)
    a, b = np.sin(6 * t) + 1.2, np.cos(5 * t) + 1.2
    # 11 near-copies of pattern a with one of them an outlier (a 2-way cut of the cluster peels it off as a singleton), 4 of b
    cols = [a * (1 + 0.002 * rng.random(80)) for _ in range(10)] + [a * (1 + 0.35 * rng.random(80))] + [b * (1 + 0.002 * rng.random(80)) for _ in range(4)]
    p = _params(2, 4)
    p.setDistributedParams(nSets=4, cut=2, minNS=2, maxNS=6)
    r = pattern_match(np.stack(cols, axis=1), p)
    sizes = [c.shape[1] for c in r["clusteredPatterns"]]
    assert max(sizes) <= 6 and min(sizes) >= 2, sizes
    assert np.allclose(r["consensus"].max(axis=0), 1.0)


def test_named_and_weighted_sets():
    """sampleWithExplictSets with names (SubsetData.R:14-27) and sampleWithAnnotationWeights (:37-55)"""
    from cogaps_amd.distributed import create_sets
    p = _params(2, 4)
    p.setDistributedParams(nSets=2, minNS=2)
    names = ["g%d" % i for i in range(1, 9)]
    p.explicitSets = [["g3", "g1", "g2", "g8"], ["g4", "g5", "g6", "g7"]]
    assert [s.tolist() for s in create_sets(8, p, names)] == [[1, 2, 3, 8], [4, 5, 6, 7]]
    p.explicitSets = [["g3", "nope"], ["g4"]]
    with pytest.raises(ValueError, match="not found"):
        create_sets(8, p, names)
    p.explicitSets = [[3, 1, 2], [4, 9]]
    with pytest.raises(ValueError, match="outside"):
        create_sets(8, p, names)
    p.explicitSets = [[3, 1, 2], [4, 8]]
    assert create_sets(8, p)[0].tolist() == [3, 1, 2]                       # index sets are used as given
    q = _params(2, 3)
    ann = ["a"] * 10 + ["b"] * 30 + ["c"] * 20
    q.setAnnotationWeights(ann, {"a": 2.0, "b": 1.0, "c": 0.0})
    sets = create_sets(60, q)
    assert len(sets) == 3 and all(len(s) == 20 and s.min() >= 1 and s.max() <= 40 and np.all(np.diff(s) >= 0) for s in sets)    # group c has weight 0
    with pytest.raises(ValueError):
        q.setAnnotationWeights(ann, {"a": 1.0})                             # a weight per group
    with pytest.raises(ValueError, match="setAnnotationWeights"):
        q.setParam("samplingWeight", {"a": 1})


def test_cogaps_does_not_modify_the_callers_params(emul_lib, modsim, monkeypatch):
    from cogaps_amd import _capi, CoGAPS, CogapsParams
    lib = emul_lib(256)
    monkeypatch.setattr(_capi, "load", lambda: lib)
    p = CogapsParams(nPatterns=3, seed=2, nIterations=20)
    r = CoGAPS(modsim, p, nPatterns=2, messages=False, outputFrequency=10, alpha=0.02)
    assert (p.nPatterns, p.alphaA, p.distributed) == (3, 0.01, None) and r.featureLoadings.shape == (25, 2)
    assert r.metadata["params"].nPatterns == 2 and r.metadata["params"].alphaA == 0.02


WORKER = r'''
import os, sys, ctypes, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import torch.distributed as dist
from cogaps_amd import _capi, CogapsParams
from cogaps_amd.distributed import distributedCogaps
import pyoracle as po
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=2)
lib = _capi.bind(ctypes.CDLL(os.path.join(%(root)r, "tests", "emul", "libcogaps_emul_TESTONLY_w256.so")))
data = po.read_mtx(os.path.join(%(root)r, "tests", "golden", "GIST.mtx"))[:240]
p = CogapsParams(nPatterns=3, seed=5, nIterations=12)
p.distributed = "genome-wide"; p.setDistributedParams(nSets=2, minNS=2)
p.explicitSets = [list(range(1, 121)), list(range(121, 241))]
run = lambda d, unc=None, **kw: _capi.run(d, unc=unc, lib=lib, **{k: v for k, v in kw.items() if k != "device"})
out = distributedCogaps(data, p, run_fn=run, outputFrequency=6)
np.savez(sys.argv[2], Amean=out["Amean"], Pmean=out["Pmean"], consensus=out["consensus"], meanChiSq=out["meanChiSq"],
         u0=out["unmatchedPatterns"][0], u1=out["unmatchedPatterns"][1])
dist.destroy_process_group()
'''


def test_world2_gloo_matches_single_process(tmp_path, emul_lib, gist):
    emul_lib(256)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "port": port})
    outs = [str(tmp_path / ("r%d.npz" % r)) for r in range(2)]
    procs = [subprocess.Popen([sys.executable, str(script), str(r), outs[r]]) for r in range(2)]
    assert all(p.wait(timeout=600) == 0 for p in procs)
    a, b = np.load(outs[0]), np.load(outs[1])
    for k in a.files:
        assert np.array_equal(a[k], b[k]), "ranks disagree on " + k
    # the same flow in one process (no collective): identical result
    from cogaps_amd import _capi, CogapsParams
    from cogaps_amd.distributed import distributedCogaps
    lib = emul_lib(256)
    p = CogapsParams(nPatterns=3, seed=5, nIterations=12)
    p.distributed = "genome-wide"; p.setDistributedParams(nSets=2, minNS=2)
    p.explicitSets = [list(range(1, 121)), list(range(121, 241))]
    run = lambda d, unc=None, **kw: _capi.run(d, unc=unc, lib=lib, **{k: v for k, v in kw.items() if k != "device"})
    ref = distributedCogaps(gist[:240], p, run_fn=run, outputFrequency=6)
    assert np.array_equal(ref["Amean"], a["Amean"]) and np.array_equal(ref["consensus"], a["consensus"])
    assert ref["Amean"].shape == (240, ref["consensus"].shape[1]) and not ref["Pmean"].any()     # fixed side comes back zero
    assert np.allclose(ref["consensus"].max(axis=0), 1.0)
    # pass 1 of shard 0 equals the oracle's asynchronous chain on the same gene subset
    import pyoracle as po
    o = po.run(gist[:240], nPatterns=3, nIterations=12, seed=5, outputFrequency=6, math_mode=po.MATH_PORTABLE, redW_A=64, redW_P=64, redG=4,
               subsetIndices=np.arange(1, 121, dtype=np.uint32), subsetDim=1)
    assert np.array_equal(o["Pmean"], a["u0"])


WORKER_N = r'''
import os, sys, ctypes, numpy as np
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "oracle"))
import torch.distributed as dist
from cogaps_amd import _capi, CogapsParams
from cogaps_amd.distributed import distributedCogaps
import pyoracle as po
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%(port)d", rank=int(sys.argv[1]), world_size=%(world)d)
lib = _capi.bind(ctypes.CDLL(os.path.join(%(root)r, "tests", "emul", "libcogaps_emul_TESTONLY_w256.so")))
data = po.read_mtx(os.path.join(%(root)r, "tests", "golden", "GIST.mtx"))[:%(rows)d]
p = CogapsParams(nPatterns=2, seed=11, nIterations=40)
p.distributed = "genome-wide"; p.setDistributedParams(nSets=%(nsets)d, minNS=2)
p.explicitSets = [list(range(1 + k * %(per)d, 1 + (k + 1) * %(per)d)) for k in range(%(nsets)d)]
run = lambda d, unc=None, **kw: _capi.run(d, unc=unc, lib=lib, **{k: v for k, v in kw.items() if k != "device"})
out = distributedCogaps(data, p, run_fn=run, outputFrequency=20)
np.savez(sys.argv[2], Amean=out["Amean"], Asd=out["Asd"], Pmean=out["Pmean"], consensus=out["consensus"], meanChiSq=out["meanChiSq"],
         **{"u%%d" %% k: u for k, u in enumerate(out["unmatchedPatterns"])})
dist.destroy_process_group()
'''


@pytest.mark.parametrize("world,nsets", [(8, 8), (3, 8)])
def test_many_ranks_gloo_match_single_process(tmp_path, emul_lib, gist, world, nsets):
    """the N > 1 path as the 8-GPU scaling run will drive it -- one subset per rank (world 8 / nSets 8) -- and with uneven ownership
    (world 3 / nSets 8: ranks own 3, 3 and 2 subsets, the gather buffers carry padded slots): every rank's stitched result, consensus
    and per-subset first-pass factors equal the single-process run bit for bit (reference flow: R/DistributedCogaps.R:57-97)"""
    emul_lib(256)
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    per = 40
    script = tmp_path / "worker.py"
    script.write_text(WORKER_N % {"root": ROOT, "port": port, "world": world, "nsets": nsets, "per": per, "rows": per * nsets})
    outs = [str(tmp_path / ("r%d.npz" % r)) for r in range(world)]
    env = dict(os.environ, OMP_NUM_THREADS="1", MKL_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), str(r), outs[r]], env=env) for r in range(world)]
    assert all(p.wait(timeout=900) == 0 for p in procs)
    from cogaps_amd import _capi, CogapsParams
    from cogaps_amd.distributed import distributedCogaps
    lib = emul_lib(256)
    p = CogapsParams(nPatterns=2, seed=11, nIterations=40)
    p.distributed = "genome-wide"; p.setDistributedParams(nSets=nsets, minNS=2)
    p.explicitSets = [list(range(1 + k * per, 1 + (k + 1) * per)) for k in range(nsets)]
    run = lambda d, unc=None, **kw: _capi.run(d, unc=unc, lib=lib, **{k: v for k, v in kw.items() if k != "device"})
    ref = distributedCogaps(gist[:per * nsets], p, run_fn=run, outputFrequency=20)
    for r in range(world):
        a = np.load(outs[r])
        for k in ("Amean", "Asd", "Pmean", "consensus", "meanChiSq"):
            assert np.array_equal(ref[k], a[k]), "rank %d: %s" % (r, k)
        for k in range(nsets):
            assert np.array_equal(ref["unmatchedPatterns"][k], a["u%d" % k]), "rank %d: first-pass factor of subset %d" % (r, k)


def test_clustering_agrees_with_scipy_on_random_matrices():
    """the restated cluster::agnes(method = "complete") + stats::cutree against an independent implementation (scipy's
    linkage('complete') + cut_tree) on 240 random 1 - cor matrices: the same partition for every cut, labels numbered by first
    appearance; and the restated stats::cor against numpy's corrcoef"""
    from scipy.cluster.hierarchy import cut_tree, linkage
    from scipy.spatial.distance import squareform
    from cogaps_amd.distributed import _complete_linkage_cutree, _cor
    rng = np.random.default_rng(11)
    for case in range(240):
        n, rows = int(rng.integers(3, 26)), int(rng.integers(5, 60))
        m = np.abs(rng.normal(size=(rows, n))) + (rng.random((rows, 1)) if case % 3 else 0.0)
        c = _cor(m)
        assert np.allclose(c, np.corrcoef(m.T), rtol=0, atol=1e-12)
        d = 1.0 - c
        d = 0.5 * (d + d.T)
        np.fill_diagonal(d, 0.0)
        z = linkage(squareform(d, checks=False), method="complete")
        for k in {1, 2, 3, max(1, n // 2), n - 1, n}:
            if k < 1:
                continue
            ours = _complete_linkage_cutree(d, k)
            ref = cut_tree(z, n_clusters=k).ravel()
            first = {}
            ref_lab = np.array([first.setdefault(int(v), len(first) + 1) for v in ref])          # cutree numbers clusters by first appearance
            assert np.array_equal(ours, ref_lab), (case, n, k)


def _gw_params(n_sets, sets):
    from cogaps_amd import CogapsParams
    p = CogapsParams(nPatterns=3, seed=5, nIterations=40)
    p.distributed = "genome-wide"
    p.setDistributedParams(nSets=n_sets, minNS=2)
    p.explicitSets = sets
    return p


def test_shard_sources_array_file_and_loader_agree(emul_lib, gist):
    """distributedCogaps takes the whole matrix, a FILE (every rank reads only its shards' rows through
    cogaps_read_matrix_file_subset, as the reference's workers do, Matrix.cpp:70-134) or a LOADER i -> shard: the same result, and
    with a file or a loader the whole matrix is never built (checked: the loader is asked for exactly the shards, once per pass)"""
    from conftest import GOLDEN
    from cogaps_amd import _capi
    from cogaps_amd.distributed import distributedCogaps
    lib = emul_lib(256)
    run = lambda d, unc=None, **kw: _capi.run(d, unc=unc, lib=lib, **{k: v for k, v in kw.items() if k != "device"})
    sets = [list(range(1, 101)), list(range(101, 181)), list(range(181, 301))]                  # uneven on purpose
    ref = distributedCogaps(gist[:300], _gw_params(3, sets), run_fn=run, outputFrequency=5)
    # (a file holds all 1363 genes: the sets name the first 300)
    for ext in ("mtx", "csv"):
        r = distributedCogaps(os.path.join(GOLDEN, "GIST." + ext), _gw_params(3, sets), run_fn=run, outputFrequency=5)
        for k in ("Amean", "Asd", "consensus"):
            assert np.array_equal(r[k], ref[k]), (ext, k)
    asked = []

    def loader(i, idx):
        asked.append((i, len(idx)))
        return np.ascontiguousarray(gist[np.asarray(idx) - 1])
    r = distributedCogaps(loader, _gw_params(3, sets), run_fn=run, outputFrequency=5, shape=(300, 9))
    assert np.array_equal(r["Amean"], ref["Amean"]) and np.array_equal(r["consensus"], ref["consensus"])
    assert sorted(asked) == sorted([(0, 100), (1, 80), (2, 120)] * 2)
    with pytest.raises(ValueError, match="shape"):
        distributedCogaps(loader, _gw_params(3, sets), run_fn=run)
    with pytest.raises(ValueError, match="expected"):
        distributedCogaps(lambda i, idx: gist[:5], _gw_params(3, sets), run_fn=run, shape=(300, 9))
    # a file's subset arrives in sorted index order (Matrix.cpp:113): unsorted explicit sets give the sorted sets' result
    shuffled = [list(reversed(s)) for s in sets]
    r = distributedCogaps(os.path.join(GOLDEN, "GIST.tsv"), _gw_params(3, shuffled), run_fn=run, outputFrequency=5)
    assert np.array_equal(r["Amean"], ref["Amean"])


def test_shards_are_grouped_by_launch_shape():
    """the batches of a rank's shards are formed from the shards' shapes up front (cogaps_reduction_width / cogaps_sparse_width of
    both samplers' vector lengths), not by parsing an error message"""
    from cogaps_amd.distributed import _launch_shape
    assert _launch_shape(20000, 2000, False) == _launch_shape(19990, 2000, False)                # same widths and slice counts
    assert _launch_shape(20000, 2000, False) != _launch_shape(9000, 2000, False)                 # 8192 vs 4096 lanes on the P side
    assert _launch_shape(50000, 12500, True) != _launch_shape(50000, 12500, False)
    assert _launch_shape(50000, 12500, True) == _launch_shape(50000, 12400, True)
