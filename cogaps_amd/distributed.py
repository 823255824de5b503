"""GWCoGAPS / scCoGAPS -- distributedCogaps (reference R/DistributedCogaps.R:40-127) with one data
subset per GPU.

  createSets                       R/SubsetData.R:85-116  (explicitSets, or a uniform partition)
  pass 1: one chain per subset     R/DistributedCogaps.R:64-68   -> rank r runs the subsets i with i % world == r
  all-gather of the shared factor  :71-74  (sampleFactors for genome-wide, featureLoadings for single-cell)
                                   -> torch.distributed.all_gather (RCCL over xGMI on GPUs, gloo in CPU tests)
  findConsensusMatrix              :129-217 (cor -> 1-cor -> complete linkage -> cutree -> split -> cubic-weighted
                                   mean -> max-normalise), computed redundantly on every rank
  pass 2: fixed matrix             :86-97
  stitchTogether                   :226-278

Differences from the reference, both documented in DESIGN.md: (1) the reference forces its sequential
sampler inside workers (:28-29); this library is the asynchronous sampler, so each shard is an asynchronous
chain (the per-shard oracle is the asynchronous reference run with the same dataIndicesSubset); (2) the
uniform partition uses numpy's generator, not R's sample(): results with explicitSets are comparable, the
random partition is not.  The reference's quirk that the shared factor comes back all-zero from pass 2
(:236-237, src/GapsRunner.cpp:301-306) is reproduced in Pmean/Amean; the consensus actually used is
returned under diagnostics["consensus"].
"""
import os

import numpy as np

from . import _capi


def _ipc_mode_for_rccl():
    """RCCL between the ranks' processes needs dmabuf IPC on this driver (HSA_ENABLE_IPC_MODE_LEGACY=0, read when HIP initialises).  The
    launcher or the rank's entry point exports it (bench.py does); a rank of a multi-process job that arrives here without it gets it
    set now -- nothing is touched on import or in a single-process call -- and is told when HIP is already up and will not see it."""
    try:
        import torch.distributed as dist
        multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    except Exception:
        multi = False
    if not multi or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0":
        return
    import warnings
    had = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if had is None:
        os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    try:
        import torch
        up = torch.cuda.is_initialized()
    except Exception:
        up = False
    if had is not None or up:
        warnings.warn("HSA_ENABLE_IPC_MODE_LEGACY is %s and HIP is %s: RCCL between processes needs it to be 0 before HIP initialises "
                      "(export it in the launcher)" % (repr(had), "already initialised" if up else "not yet initialised"))


# ------------------------------------------------------------------------------------------------
def _explicit_sets(params, names):
    """sampleWithExplictSets, R/SubsetData.R:7-29: index sets as given (not sorted, as in the reference), or sets of gene /
    sample names mapped through the names of the partitioned dimension (`which(allNames %in% set)`: ascending indices)"""
    sets = list(params.explicitSets)
    is_name = [all(isinstance(x, str) for x in s) and len(s) > 0 for s in sets]
    if all(is_name):
        if names is None:
            raise ValueError("explicitSets holds names but no %s were given" % ("geneNames" if params.distributed == "genome-wide" else "sampleNames"))
        lookup = {}
        for i, n in enumerate(names):
            lookup.setdefault(n, []).append(i + 1)
        out = []
        for st in sets:
            if any(x not in lookup for x in st):
                raise ValueError("some named genes in explicitSets not found")
            out.append(np.sort(np.array([i for x in set(st) for i in lookup[x]], dtype=np.int64)))
        return out
    if any(is_name):
        raise ValueError("explicitSets must be all index sets or all name sets")
    return [np.asarray(s, dtype=np.int64) for s in sets]


def _annotation_weight_sets(params, set_size, rng):
    """sampleWithAnnotationWeights, R/SubsetData.R:37-55: every set draws `set_size` group labels with the given weights, then that
    many members of each group with replacement; sorted.  (numpy's generator, not R's sample(): same distribution, other draws.)"""
    ann = np.asarray(params.samplingAnnotation)
    weight = dict(params.samplingWeight)
    groups = sorted(set(ann.tolist()))
    if sorted(weight) != groups:
        raise ValueError("samplingWeight must name every group of samplingAnnotation")
    prob = np.array([float(weight[g]) for g in groups], dtype=np.float64)
    prob = prob / prob.sum()
    sets = []
    for _ in range(params.nSets):
        counts = rng.multinomial(set_size, prob)
        picks = [rng.choice(np.nonzero(ann == g)[0] + 1, size=c, replace=True) for g, c in zip(groups, counts) if c]
        sets.append(np.sort(np.concatenate(picks)).astype(np.int64))
    return sets


def create_sets(total, params, names=None):
    """1-based index sets (R convention), R/SubsetData.R:85-116"""
    if params.explicitSets is not None:
        if len(params.explicitSets) != params.nSets:
            raise ValueError("nSets does not match number of explicit sets given")
        sets = _explicit_sets(params, names)
        for st in sets:
            if st.size == 0 or st.min() < 1 or st.max() > total:
                raise ValueError("explicitSets holds an index outside 1 .. %d" % total)
        return sets
    rng = np.random.Generator(np.random.MT19937(int(params.seed)))
    set_size = total // params.nSets
    if params.samplingAnnotation is not None:
        if len(params.samplingAnnotation) != total:
            raise ValueError("samplingAnnotation must label every one of the %d partitioned rows / columns" % total)
        return _annotation_weight_sets(params, set_size, rng)
    remaining = np.arange(1, total + 1)
    sets = []
    for _ in range(params.nSets - 1):                      # sampleUniformly, SubsetData.R:63-76
        sel = rng.choice(remaining, set_size, replace=False)
        sets.append(np.sort(sel))
        remaining = np.setdiff1d(remaining, sel)
    sets.append(np.sort(remaining))
    return sets


# ------------------------------------------------------------------------------------------------
def _complete_linkage_cutree(dist, k):
    """cluster::agnes(diss, method="complete") + stats::cutree(k): labels numbered by first appearance"""
    n = dist.shape[0]
    if k >= n:
        return np.arange(1, n + 1)
    members = {i: [i] for i in range(n)}
    d = dist.astype(np.float64).copy()
    np.fill_diagonal(d, np.inf)
    active = list(range(n))
    while len(active) > k:
        sub = d[np.ix_(active, active)]
        a, b = np.unravel_index(np.argmin(sub), sub.shape)
        i, j = active[min(a, b)], active[max(a, b)]
        for o in active:                                   # complete linkage: distance = max over members
            if o != i and o != j:
                d[i, o] = d[o, i] = max(d[i, o], d[j, o])
        members[i] += members.pop(j)
        active.remove(j)
    labels = np.zeros(n, dtype=np.int64)
    nxt = 1
    for obs in range(n):
        if labels[obs] == 0:
            root = next(r for r, m in members.items() if obs in m)
            labels[members[root]] = nxt
            nxt += 1
    return labels


def _cor(m):
    """stats::cor of the columns (Pearson)"""
    x = m.astype(np.float64)
    x = x - x.mean(axis=0, keepdims=True)
    s = np.sqrt((x * x).sum(axis=0))
    return (x.T @ x) / np.outer(s, s)


def corcut(all_patterns, cut, min_ns):
    """R/DistributedCogaps.R:197-217: list of column-index arrays, one per kept cluster"""
    corr_dist = 1.0 - _cor(all_patterns)
    if np.isnan(corr_dist).any():
        raise ValueError("NA values in correlation of patterns")
    ids = _complete_linkage_cutree(corr_dist, cut)
    out = []
    for c in dict.fromkeys(ids.tolist()):                  # unique(), order of first appearance
        cols = np.nonzero(ids == c)[0]
        if cols.size >= min_ns:
            out.append(cols)
    return out


def corr_to_mean_pattern(cluster):
    """R/DistributedCogaps.R:186-190: round(cor(column, rowMeans), 3)"""
    mean_pat = cluster.astype(np.float64).mean(axis=1)
    out = []
    for j in range(cluster.shape[1]):
        x = cluster[:, j].astype(np.float64)
        c = np.corrcoef(x, mean_pat)[0, 1]
        out.append(np.round(c, 3))
    return np.array(out)


def pattern_match(all_patterns, params):
    """R/DistributedCogaps.R:145-177"""
    clusters = [all_patterns[:, c] for c in corcut(all_patterns, params.cut, params.minNS)]
    while True:                                            # split clusters larger than maxNS in two
        big = [i for i, c in enumerate(clusters) if c.shape[1] > params.maxNS]
        if not big:
            break
        i = big[0]
        split = corcut(clusters[i], 2, params.minNS)
        parts = [clusters[i][:, s] for s in split]
        if not parts:                                      # neither half reaches minNS: R would index NULL; drop the cluster
            clusters.pop(i)
            continue
        # splitCluster (:151-158): the first half replaces the cluster, the second is appended; the loop goes on until no
        # cluster exceeds maxNS (cutree(k = 2) always takes at least one column away, so it ends)
        clusters[i] = parts[0]
        if len(parts) > 1:
            clusters.append(parts[1])
    if not clusters:
        raise ValueError("no cluster of patterns reaches minNS members")
    mean_patterns = np.stack([
        (c.astype(np.float64) * (corr_to_mean_pattern(c) ** 3)[None, :]).sum(axis=1) / (corr_to_mean_pattern(c) ** 3).sum()
        for c in clusters], axis=1)                        # weighted.mean(row, corrToMeanPattern^3)
    consensus = mean_patterns / mean_patterns.max(axis=0, keepdims=True)
    return {"clusteredPatterns": clusters, "consensus": consensus.astype(np.float32)}


def find_consensus_matrix(unmatched, params):
    """R/DistributedCogaps.R:129-135"""
    return pattern_match(np.concatenate(unmatched, axis=1), params)


# ------------------------------------------------------------------------------------------------
def _dist():
    try:
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        pass
    return None


def _all_gather_arrays(local, shapes, dist, device):
    """all-gather a list of equally shaped fp32 matrices per owned subset; returns them in subset order"""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    n_sets = len(shapes)
    per_rank = (n_sets + world - 1) // world
    buf = torch.zeros((per_rank,) + tuple(shapes[0]), dtype=torch.float32, device=device)
    for slot, i in enumerate(range(rank, n_sets, world)):
        buf[slot] = torch.from_numpy(np.ascontiguousarray(local[i], dtype=np.float32)).to(device)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf)
    res = [None] * n_sets
    for r in range(world):
        for slot, i in enumerate(range(r, n_sets, world)):
            res[i] = out[r][slot].cpu().numpy()
    return res


def _current_device(run_fn):
    """ordinal of the calling thread's current GPU: torch's when it has initialised HIP, else hipGetDevice through the library"""
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            return int(torch.cuda.current_device())
    except Exception:
        pass
    if run_fn is _capi.run:
        return _capi.current_device()
    return 0


def _launch_shape(n_genes, n_samples, sparse):
    """what cogaps_batch_create requires the chains of a batch to share (cogaps_hip.cpp): the reduction widths and slice counts of the
    two samplers' data vectors (A: samples long, P: genes long), the sparse model's workgroup widths"""
    L = _capi.load()
    key = []
    for n in (n_samples, n_genes):
        npad = (n + 3) & ~3
        key += [L.cogaps_reduction_width(n), ((npad >> 2) + 511) // 512, L.cogaps_sparse_width(n) if sparse else 0]
    return tuple(key)


def _run_shards(spec, ids, in_flight, run_fn, shape_of=None):
    """This rank's shards, `in_flight` at a time (the role of BPPARAM's workers, DistributedCogaps.R:60-63, 84-87).  spec(i) gives
    (data, uncertainty, keyword arguments) of shard i's cogaps_run call -- it is called when the shard's turn comes and its matrices
    are dropped as soon as the shard's session holds them in HBM, so the host never keeps more than one group's shards;
    shape_of(i) = (genes, samples, sparse, has uncertainty) of the shard without loading it.

    With the product library the shards in flight run as batches (cogaps_batch_*, batched multi-chain launches: one generator
    launch with a workgroup per chain and one evaluation launch over all chains' queues per step) -- a single chain keeps one
    workgroup busy in its generator kernel and a few hundred in its evaluation kernel, alternately, so the chains of a batch cost
    about the time of one; from four shards on they run as two such batches on two host threads.  A batch needs one evaluation launch
    shape: the shards are grouped by it up front (`_launch_shape`; very uneven subsets give several groups, a group of one runs as
    a plain session), nothing is decided by parsing error messages.  A group that does not fit the GPU's memory after all is retried
    with half as many shards in flight; finished shards are kept.  Another run_fn (tests) runs one host thread per shard in flight.
    Either way every shard's chain is bit-identical to the chain it runs alone."""
    ids = list(ids)

    def one(i):
        d, u, k = spec(i)
        return run_fn(d, unc=u, **k)
    if in_flight <= 1 or len(ids) <= 1:
        return {i: one(i) for i in ids}
    if run_fn is _capi.run and shape_of is not None:
        out = {}
        # no more shards in flight than the GPU holds: a session keeps ~9 matrices of its shard's size resident (data, A*P cache,
        # uncertainty, for both samplers) -- counted as 10 (13 with an uncertainty matrix) against 85 % of the free
        # HBM; the sparse model keeps the dense data and uncertainty for meanChiSq plus the packed vectors: counted as 5
        g0_, s0_, sparse0, unc0 = shape_of(ids[0])
        per_shard = (5 if sparse0 else (13 if unc0 else 10)) * 4 * max(g_ * s_ for g_, s_, _, _ in (shape_of(i) for i in ids)) + (64 << 20)
        device = spec.device if hasattr(spec, "device") else -1
        free_bytes, _ = _capi.device_memory(device)
        in_flight = max(1, min(in_flight, int(0.85 * free_bytes // per_shard)))
        by_shape = {}
        for i in ids:
            g_, s_, sp_, _ = shape_of(i)
            by_shape.setdefault(_launch_shape(g_, s_, sp_), []).append(i)

        def batch(grp):
            if len(grp) == 1:
                return [one(grp[0])]
            return _capi.run_batch((spec(i) for i in grp))
        for members in by_shape.values():
            g0 = 0
            while g0 < len(members):
                grp = members[g0:g0 + in_flight]
                # Two batches on two host threads and streams once there are chains enough: one batch's generator launch (one
                # workgroup per chain, the latency-bound step) then runs under the other's evaluation launches.  Measured with the C3
                # shape: 8 chains 21.2 -> 23.3 M proposals/s, 16 chains 29.0 -> 34.4 M, 32 chains 35.7 -> 44.0 M.
                halves = [grp[0::2], grp[1::2]] if len(grp) >= 4 else [grp]

                def attempt(h):
                    """a half's results, or the typed error that ended it (device memory: the footprint estimate was too low)"""
                    try:
                        return batch(h)
                    except _capi.OutOfDeviceMemory as e:      # COGAPS_ERR_OUT_OF_DEVICE_MEMORY from the C ABI, not a message text
                        return e
                if len(halves) == 1:
                    res = [attempt(grp)]
                else:
                    from concurrent.futures import ThreadPoolExecutor
                    with ThreadPoolExecutor(max_workers=2) as pool:
                        res = list(pool.map(attempt, halves))
                failed = []
                for h, rr in zip(halves, res):
                    if isinstance(rr, _capi.OutOfDeviceMemory):
                        failed += h
                        continue
                    for i, r in zip(h, rr):       # a half that finished is kept whatever happened to the other
                        out[i] = r
                if failed:
                    if in_flight <= 1:
                        raise next(rr for rr in res if isinstance(rr, _capi.OutOfDeviceMemory))
                    in_flight = max(1, in_flight // 2)
                    # only the shards of the half that failed are run again, with half as many in flight
                    members = members[:g0] + failed + [i for i in members[g0 + len(grp):]]
                    continue
                g0 += len(grp)
        return out
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(in_flight, 4)) as pool:       # (HIP gives a process four hardware queues: DESIGN.md section 5)
        futs = {i: pool.submit(one, i) for i in ids}
        return {i: f.result() for i, f in futs.items()}


class _Source:
    """Where the shards of a distributed run come from.  `data` is
      * a 2-D array: shard i = its rows / columns sets[i] (in the order given, Matrix(mat, ...), Matrix.cpp:30-69);
      * a path to a .mtx / .csv / .tsv / .gct file: every rank reads only ITS shards' rows / columns of the file with the library's
        reader (cogaps_read_matrix_file_subset) -- as the reference's workers do (Matrix(path, ...), Matrix.cpp:70-134), which sorts
        the indices first: the sets are sorted to match;
      * a callable loader(i, indices) -> matrix or (matrix, uncertainty): shard i's own contiguous sub-matrix for the 1-based
        `indices` (rows of the data when the partitioned dimension is the rows, else columns), with `shape` = dimensions of the whole.
    No rank ever materialises the whole matrix unless the caller hands it over as an array."""

    def __init__(self, data, uncertainty, shape, subset_rows):
        self.data, self.unc, self.subset_rows = data, uncertainty, subset_rows
        self.kind = "path" if isinstance(data, (str, bytes)) else ("loader" if callable(data) else "array")
        if self.kind == "array":
            self.shape = tuple(data.shape)
        elif self.kind == "path":
            nr, nc, self.row_names, self.col_names = _capi.file_info(data)
            self.shape = (nr, nc)
            if uncertainty is not None and not isinstance(uncertainty, (str, bytes)):
                raise ValueError("data given as a file needs the uncertainty as a file, too")
        else:
            if shape is None:
                raise ValueError("a shard loader needs shape=(rows, columns) of the whole data matrix")
            if uncertainty is not None:
                raise ValueError("a shard loader returns (matrix, uncertainty) itself")
            self.shape = (int(shape[0]), int(shape[1]))

    def has_unc(self):
        return self.unc is not None

    def shard(self, i, idx1):
        if self.kind == "array":
            idx = idx1 - 1
            cut = (lambda m: np.ascontiguousarray(m[idx, :] if self.subset_rows else m[:, idx], dtype=np.float32))
            return cut(self.data), (None if self.unc is None else cut(self.unc))
        if self.kind == "path":
            kw = {"rows": idx1} if self.subset_rows else {"cols": idx1}
            return _capi.read_matrix_file(self.data, **kw), (None if self.unc is None else _capi.read_matrix_file(self.unc, **kw))
        r = self.data(i, idx1)
        d, u = r if isinstance(r, tuple) else (r, None)
        d = np.ascontiguousarray(d, dtype=np.float32)
        want = (len(idx1), self.shape[1]) if self.subset_rows else (self.shape[0], len(idx1))
        if d.shape != want:
            raise ValueError("the shard loader returned %s for shard %d, expected %s" % (d.shape, i, want))
        return d, (None if u is None else np.ascontiguousarray(u, dtype=np.float32))


def distributedCogaps(data, params, uncertainty=None, messages=False, outputFrequency=1000, transposeData=False,
                      device=-1, run_fn=None, comm_device=None, shardsInFlight=16, nSnapshots=0, snapshotPhase="sampling", shape=None):
    _ipc_mode_for_rccl()
    run_fn = run_fn or _capi.run
    shardsInFlight = max(1, int(shardsInFlight))
    genome_wide = params.distributed == "genome-wide"
    subset_rows = bool(transposeData) != genome_wide          # xor, SubsetData.R:87-88
    src = _Source(data, uncertainty, shape, subset_rows)
    total = src.shape[0] if subset_rows else src.shape[1]
    sets = create_sets(total, params, params.geneNames if genome_wide else params.sampleNames)
    if src.kind == "path":
        sets = [np.sort(st) for st in sets]                    # Matrix.cpp:113: a worker reading a file sorts its indices
    if min(len(s) for s in sets) < params.nPatterns:
        raise ValueError("data subset dimension less than nPatterns")
    dist = _dist()
    world, rank = (dist.get_world_size(), dist.get_rank()) if dist else (1, 0)
    if world > len(sets):                                      # every rank sees this before the first collective: none is left waiting
        raise ValueError("more ranks (%d) than subsets (%d)" % (world, len(sets)))
    if device is None or device < 0:
        # The HIP current device belongs to the calling host thread; the shards in flight run on pool threads that start on
        # device 0.  Resolve the ordinal once, here, and hand it to every shard explicitly.
        device = _current_device(run_fn)
    if comm_device is None:
        comm_device = "cpu"
        if dist is not None and dist.get_backend() == "nccl":
            import torch
            comm_device = torch.device("cuda", torch.cuda.current_device())
    mine = list(range(rank, len(sets), world))
    common = dict(nIterations=params.nIterations, seed=params.seed, outputFrequency=outputFrequency, alphaA=params.alphaA,
                  alphaP=params.alphaP, maxGibbsMassA=params.maxGibbsMassA, maxGibbsMassP=params.maxGibbsMassP,
                  transposeData=transposeData, sparseOptimization=params.sparseOptimization, device=device,
                  takePumpSamples=params.takePumpSamples, nSnapshots=nSnapshots, snapshotPhase=snapshotPhase)      # allParams reaches every worker unchanged (:12-35)

    def shape_of(i):
        """(genes, samples, sparse model, uncertainty given) of shard i, from the index sets alone"""
        r, c = (len(sets[i]), src.shape[1]) if subset_rows else (src.shape[0], len(sets[i]))
        g_, s_ = (c, r) if transposeData else (r, c)
        return g_, s_, bool(params.sparseOptimization), src.has_unc()

    def make_spec(n_patterns, fixed=None, which="N"):          # callInternalCoGAPS, DistributedCogaps.R:12-35
        def spec(i):
            # rows (columns) sets[i] of the data / uncertainty as their own contiguous matrices, cut (or read, or loaded) when the
            # shard's turn comes: the library then holds and uploads one shard, never the whole matrix
            d, u = src.shard(i, sets[i])
            return d, u, dict(nPatterns=n_patterns, runningDistributed=True, workerID=i + 1, messages=messages, whichMatrixFixed=which,
                              fixedPatterns=fixed, **common)
        spec.device = device
        return spec

    initial, unmatched, matched = None, None, None
    if params.fixedPatterns is None:
        initial = _run_shards(make_spec(params.nPatterns), mine, shardsInFlight, run_fn, shape_of)
        key = "Pmean" if genome_wide else "Amean"
        local = {i: initial[i][key] for i in mine}
        if dist is not None:
            shape = next(iter(local.values())).shape
            unmatched = _all_gather_arrays(local, [shape] * len(sets), dist, comm_device)
        else:
            unmatched = [local[i] for i in range(len(sets))]
        matched = find_consensus_matrix(unmatched, params)
        if dist is not None:
            # the consensus is computed redundantly on every rank from the gathered patterns: the second pass is only meaningful
            # if all ranks hold the same bits -- checked, not assumed
            import hashlib
            import torch
            dig = np.frombuffer(hashlib.sha256(np.ascontiguousarray(matched["consensus"]).tobytes()).digest()[:8], dtype=np.int64).copy()
            mine_t = torch.from_numpy(dig).to(comm_device)
            every = [torch.empty_like(mine_t) for _ in range(world)]
            dist.all_gather(every, mine_t)
            if any(int(t.item()) != int(dig[0]) for t in every):
                raise RuntimeError("the ranks computed different consensus matrices from the same gathered patterns")
    else:
        matched = {"consensus": np.asarray(params.fixedPatterns, dtype=np.float32), "clusteredPatterns": None}

    consensus = matched["consensus"]
    which = "P" if genome_wide else "A"
    final = _run_shards(make_spec(consensus.shape[1], fixed=consensus, which=which), mine, shardsInFlight, run_fn, shape_of)

    # stitchTogether (DistributedCogaps.R:226-278): collect the per-subset free factor on every rank
    free_key, free_sd = ("Amean", "Asd") if genome_wide else ("Pmean", "Psd")
    if dist is not None:
        # tensor all-gathers (RCCL on GPUs): the free factor's mean and standard deviation, padded to the longest subset, and
        # one small record per subset (meanChiSq); the shared factor comes back from every shard as the same all-zero matrix
        rows = max(len(st) for st in sets)
        k2 = consensus.shape[1]

        def padded(key):
            return {i: np.concatenate([final[i][key], np.zeros((rows - final[i][key].shape[0], k2), np.float32)], axis=0) for i in mine}
        g_mean = _all_gather_arrays(padded(free_key), [(rows, k2)] * len(sets), dist, comm_device)
        g_sd = _all_gather_arrays(padded(free_sd), [(rows, k2)] * len(sets), dist, comm_device)
        # (meanChiSq travels as two fp32 halves of its float64 value: the multi-rank sum equals the single-process one to the last digit)
        def split64(x):
            hi = np.float32(x)
            return np.array([[hi, np.float32(float(x) - float(hi))]], dtype=np.float32)
        g_chi = _all_gather_arrays({i: split64(final[i]["meanChiSq"]) for i in mine}, [(1, 2)] * len(sets), dist, comm_device)
        shared_zero = np.zeros_like(final[mine[0]]["Pmean" if genome_wide else "Amean"])
        allf = {i: {free_key: g_mean[i][:len(sets[i])], free_sd: g_sd[i][:len(sets[i])], "meanChiSq": float(g_chi[i][0, 0]) + float(g_chi[i][0, 1]),
                    ("Pmean" if genome_wide else "Amean"): shared_zero} for i in range(len(sets))}
    else:
        allf = final
    order = list(range(len(sets)))
    free_mean = np.concatenate([allf[i][free_key] for i in order], axis=0)
    free_dev = np.concatenate([allf[i][free_sd] for i in order], axis=0)
    set_indices = np.concatenate(sets)
    if free_mean.shape[0] == set_indices.size and np.array_equal(np.sort(set_indices), np.arange(1, free_mean.shape[0] + 1)):
        reorder = np.argsort(set_indices, kind="stable")           # match(1:n, setIndices)
        free_mean, free_dev = free_mean[reorder], free_dev[reorder]
    shared = allf[0]["Pmean" if genome_wide else "Amean"]           # "same for all sets": zeros, see module docstring
    out = {
        "Amean": free_mean if genome_wide else shared, "Asd": free_dev if genome_wide else np.zeros_like(shared),
        "Pmean": shared if genome_wide else free_mean, "Psd": np.zeros_like(shared) if genome_wide else free_dev,
        "seed": params.seed, "meanChiSq": float(sum(allf[i]["meanChiSq"] for i in order)),
        "subsets": sets, "consensus": consensus,
    }
    if initial is not None:
        out["firstPass"] = initial
        out["unmatchedPatterns"] = unmatched
        out["clusteredPatterns"] = matched["clusteredPatterns"]
        out["CorrToMeanPattern"] = [corr_to_mean_pattern(c) for c in matched["clusteredPatterns"]]
    return out
