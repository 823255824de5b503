"""CoGAPS() -- the reference's user entry point (R/CoGAPS.R:90-155) over the HIP library.

Same arguments and defaults; `data` is a 2-D array (genes x samples unless transposeData) or a path to a
.mtx/.csv/.tsv file.  The standard run dispatches to cogaps_run (the C-ABI replacement of gaps::run);
`distributed` = "genome-wide" / "single-cell" dispatches to cogaps_amd.distributed (GWCoGAPS / scCoGAPS).
"""
import warnings

import numpy as np

from . import _capi
from .io import read_matrix
from .params import CogapsParams
from .result import CogapsResult


def check_inputs(data, uncertainty, params, snapshotPhase="sampling", nSnapshots=0, checkpointInFile=None, nThreads=1):
    """R/HelperFunctions.R:194-249 (checkDataMatrix + checkInputs)"""
    if uncertainty is not None and params.sparseOptimization:
        raise ValueError("must use default uncertainty when enabling sparseOptimization")
    if checkpointInFile is not None:
        raise ValueError("CoGAPS was built with checkpoints disabled")
    if snapshotPhase not in ("equilibration", "sampling", "all"):
        raise ValueError("snapshotPhase must be either equilibration, sampling, or all")
    if params.distributed is not None and nThreads > 1:
        warnings.warn("can't run multi-threaded and distributed CoGAPS at the same time, ignoring nThreads")
    if np.isnan(data).any():
        raise ValueError("NA values in data")
    if (data < 0).any() or (uncertainty is not None and (uncertainty < 0).any()):
        raise ValueError("negative values in data and/or uncertainty matrix")
    if data.shape[0] <= params.nPatterns or data.shape[1] <= params.nPatterns:
        raise ValueError("nPatterns must be less than dimensions of data")
    if uncertainty is not None and (uncertainty < 1e-5).any():
        warnings.warn("small values in uncertainty matrix detected")


def CoGAPS(data, params=None, nPatterns=None, nThreads=1, messages=True, outputFrequency=1000, uncertainty=None,
           checkpointOutFile="gaps_checkpoint.out", checkpointInterval=0, checkpointInFile=None, transposeData=False,
           BPPARAM=None, workerID=1, asynchronousUpdates=True, nSnapshots=0, snapshotPhase="sampling", device=-1, **extra):
    if params is None:
        params = CogapsParams(**({} if nPatterns is None else {"nPatterns": nPatterns}))
    else:
        params = params.copy()                                   # value semantics of the S4 object: the caller's params stay as they are
        if nPatterns is not None:
            params.setParam("nPatterns", nPatterns)
    for k, v in extra.items():                                   # parseExtraParams: named CogapsParams slots in ...
        params.setParam(k, v)
    params.validate()
    if isinstance(data, str):
        data = read_matrix(data)
    data = np.ascontiguousarray(data, dtype=np.float32)
    unc = None if uncertainty is None else np.ascontiguousarray(read_matrix(uncertainty) if isinstance(uncertainty, str) else uncertainty, dtype=np.float32)
    check_inputs(data, unc, params, snapshotPhase, nSnapshots, checkpointInFile, nThreads)
    if not asynchronousUpdates:
        raise ValueError("asynchronousUpdates=FALSE selects the reference's sequential sampler; this library is the asynchronous one")
    if params.distributed is not None:
        from .distributed import distributedCogaps
        # BPPARAM: the reference hands the subsets to that many BiocParallel workers (R/DistributedCogaps.R:60-63); here: shards in
        # flight per GPU, run as batches of lock-stepped chains (an int, or an object with a `workers` attribute; default 16 = two batches of eight)
        in_flight = 16 if BPPARAM is None else int(getattr(BPPARAM, "workers", BPPARAM))
        raw = distributedCogaps(data, params, unc, messages=messages, outputFrequency=outputFrequency, transposeData=transposeData, device=device,
                                shardsInFlight=in_flight, nSnapshots=nSnapshots, snapshotPhase=snapshotPhase)
    else:
        raw = _capi.run(data, unc=unc, nPatterns=params.nPatterns, nIterations=params.nIterations, seed=params.seed,
                        outputFrequency=outputFrequency, nThreads=nThreads, alphaA=params.alphaA, alphaP=params.alphaP,
                        maxGibbsMassA=params.maxGibbsMassA, maxGibbsMassP=params.maxGibbsMassP, transposeData=transposeData,
                        subsetIndices=params.subsetIndices, subsetDim=params.subsetDim, whichMatrixFixed=params.whichMatrixFixed,
                        fixedPatterns=params.fixedPatterns, sparseOptimization=params.sparseOptimization, messages=messages,
                        workerID=workerID, device=device, takePumpSamples=params.takePumpSamples,
                        nSnapshots=nSnapshots, snapshotPhase=snapshotPhase)
    return CogapsResult(raw, params=params, geneNames=params.geneNames, sampleNames=params.sampleNames)


def GWCoGAPS(data, params=None, nPatterns=None, **kw):
    """R/CoGAPS.R:213-224"""
    params = params.copy() if params is not None else CogapsParams(**({} if nPatterns is None else {"nPatterns": nPatterns}))
    params.distributed = "genome-wide"
    return CoGAPS(data, params, nPatterns, **kw)


def scCoGAPS(data, params=None, nPatterns=None, **kw):
    """R/CoGAPS.R:173-184"""
    params = params.copy() if params is not None else CogapsParams(**({} if nPatterns is None else {"nPatterns": nPatterns}))
    params.distributed = "single-cell"
    params.sparseOptimization = kw.pop("sparseOptimization", params.sparseOptimization)
    return CoGAPS(data, params, nPatterns, **kw)
