"""CogapsParams -- mirror of the reference's S4 class (R/class-CogapsParams.R:44-123, methods in
R/methods-CogapsParams.R): same slot names, same defaults, same validity rules."""
import numpy as np


class CogapsParams:
    # R/class-CogapsParams.R:99-123 (initialize) -- defaults
    _DEFAULTS = dict(
        nPatterns=7, nIterations=50000, alphaA=0.01, alphaP=0.01, maxGibbsMassA=100.0, maxGibbsMassP=100.0,
        seed=None, sparseOptimization=False, distributed=None, nSets=4, cut=None, minNS=None, maxNS=None,
        explicitSets=None, samplingAnnotation=None, samplingWeight=None, subsetIndices=None, subsetDim=0,
        geneNames=None, sampleNames=None, fixedPatterns=None, whichMatrixFixed="N", takePumpSamples=False,
        checkpointInterval=0, checkpointInFile="", checkpointOutFile="",
    )

    def __init__(self, **kwargs):
        for k, v in self._DEFAULTS.items():
            setattr(self, k, v)
        if self.seed is None:
            import time
            self.seed = int(time.time() * 1000) % 10000          # getMilliseconds(as.POSIXlt(Sys.time()))
        self._sync_derived()
        self.setParams(**kwargs)

    def _sync_derived(self):
        # class-CogapsParams.R:112-118: cut = nPatterns, minNS = ceiling(nSets/2), maxNS = minNS + nSets
        if self.cut is None:
            self.cut = self.nPatterns
        if self.minNS is None:
            self.minNS = int(np.ceil(self.nSets / 2))
        if self.maxNS is None:
            self.maxNS = self.minNS + self.nSets

    def _guarded(self, fn):
        """R's setters return a modified copy and fail validObject() without touching the original"""
        saved = dict(self.__dict__)
        try:
            fn()
            self.validate()
        except Exception:
            self.__dict__.clear()
            self.__dict__.update(saved)
            raise
        return self

    def setParam(self, name, value):
        """methods-CogapsParams.R:35-76"""
        return self._guarded(lambda: self._set(name, value))

    def _set(self, name, value):
        if name == "alpha":
            self.alphaA = self.alphaP = value
        elif name == "maxGibbsMass":
            self.maxGibbsMassA = self.maxGibbsMassP = value
        elif name in ("nSets", "cut", "minNS", "maxNS"):
            raise ValueError("please set this parameter with setDistributedParams")
        elif name in ("samplingAnnotation", "samplingWeight"):
            raise ValueError("please set this parameter with setAnnotationWeights")
        elif name in ("fixedPatterns", "whichMatrixFixed"):
            raise ValueError("please set this parameter with setFixedPatterns")
        elif name == "distributed":
            if value is not None and value == "none":
                value = None
            self.distributed = value
        elif name == "nPatterns":
            self.nPatterns = int(value)
            self.cut = min(self.cut, self.nPatterns) if self.cut is not None else self.nPatterns
        elif name in self._DEFAULTS:
            setattr(self, name, value)
        else:
            raise ValueError("unknown parameter: %s" % name)

    def setParams(self, **kw):
        for k, v in kw.items():
            self.setParam(k, v)
        return self

    def setDistributedParams(self, nSets=None, cut=None, minNS=None, maxNS=None):
        """methods-CogapsParams.R:84-110"""
        if self.distributed is None:
            import warnings
            warnings.warn("setting distributed parameters while distributed is NULL")
        if nSets is not None:
            self.nSets = int(nSets)
        self.cut = self.nPatterns if cut is None else int(cut)
        self.minNS = int(np.ceil(self.nSets / 2)) if minNS is None else int(minNS)
        self.maxNS = self.minNS + self.nSets if maxNS is None else int(maxNS)
        self.validate()
        return self

    def setAnnotationWeights(self, annotation, weights):
        """methods-CogapsParams.R:179-187: group label per row / column of the partitioned dimension, and a weight per group (a
        mapping group -> weight, R's named vector); used by createSets (R/SubsetData.R:37-55)"""
        def fn():
            self.samplingAnnotation = None if annotation is None else list(annotation)
            self.samplingWeight = None if weights is None else dict(weights)
        return self._guarded(fn)

    def copy(self):
        """R's S4 objects have value semantics: CoGAPS() works on its own copy"""
        import copy
        return copy.deepcopy(self)

    def setFixedPatterns(self, fixedPatterns, whichMatrixFixed):
        """methods-CogapsParams.R:139-150"""
        def fn():
            self.fixedPatterns = np.asarray(fixedPatterns, dtype=np.float64)
            self.whichMatrixFixed = whichMatrixFixed
        return self._guarded(fn)

    def getParam(self, name):
        return getattr(self, name)

    def validate(self):
        """class-CogapsParams.R:131-193 (validity)"""
        if self.nPatterns <= 0 or int(self.nPatterns) != self.nPatterns:
            raise ValueError("number of patterns must be an integer greater than zero")
        if self.nIterations <= 0 or int(self.nIterations) != self.nIterations:
            raise ValueError("number of iterations must be an integer greater than zero")
        if self.alphaA <= 0 or self.alphaP <= 0:
            raise ValueError("alpha parameter must be greater than zero")
        if self.maxGibbsMassA <= 0 or self.maxGibbsMassP <= 0:
            raise ValueError("maxGibbsMass must be greater than zero")
        if self.seed is not None and self.seed <= 0:
            raise ValueError("random seed must be an integer greater than zero")
        if self.nSets is not None and self.minNS is not None and self.minNS <= 1 and self.nSets > 1 and self.distributed is not None:
            raise ValueError("minNS must be greater than one")
        if self.nSets is not None and self.nSets <= 1:
            raise ValueError("number of sets must be greater than 1")
        if self.whichMatrixFixed not in ("A", "P", "N"):
            raise ValueError("Invalid choice of whichMatrixFixed, must be 'A' or 'P'")
        if self.whichMatrixFixed in ("A", "P") and self.fixedPatterns is None:
            raise ValueError("whichMatrixFixed is set without passing a fixedPatterns matrix")
        if self.distributed is not None and self.distributed not in ("genome-wide", "single-cell"):
            raise ValueError("distributed method must be either 'genome-wide' or 'single-cell'")
        if (self.samplingAnnotation is None) != (self.samplingWeight is None):
            raise ValueError("samplingAnnotation and samplingWeight must be set together (setAnnotationWeights)")
        if self.samplingWeight is not None:
            # class-CogapsParams.R validity: named, non-negative weights, one per annotation group
            if any(float(w) < 0 for w in self.samplingWeight.values()):
                raise ValueError("samplingWeight must be non-negative")
            if set(self.samplingWeight) != set(self.samplingAnnotation):
                raise ValueError("names of samplingWeight must match the groups of samplingAnnotation")
        if self.fixedPatterns is not None and np.any(np.asarray(self.fixedPatterns) < 0):
            raise ValueError("fixedPatterns must be non-negative")
        return True
