"""CogapsResult -- the numeric core of the reference's S4 result (R/class-CogapsResult.R): the four
factor matrices under their LinearEmbeddingMatrix names and the metadata list createCogapsResult fills
(R/methods-CogapsResult.R:8-20).  Plots / gene-set statistics are out of scope."""
import numpy as np


class CogapsResult:
    def __init__(self, raw, params=None, geneNames=None, sampleNames=None):
        self.featureLoadings = np.asarray(raw["Amean"])      # Amean
        self.loadingStdDev = np.asarray(raw["Asd"])          # Asd
        self.sampleFactors = np.asarray(raw["Pmean"])        # Pmean
        self.factorStdDev = np.asarray(raw["Psd"])           # Psd
        self.geneNames = geneNames
        self.sampleNames = sampleNames
        diag = {k: raw[k] for k in ("chisq", "atomsA", "atomsP", "averageQueueLengthA", "averageQueueLengthP",
                                    "totalUpdates", "totalRunningTime") if k in raw}
        # Cogaps.cpp:176-185: pumpStat, meanPatternAssignment, the four snapshot lists
        if "pumpMatrix" in raw:
            diag["pumpStat"] = raw["pumpMatrix"]; diag["meanPatternAssignment"] = raw["meanPatternAssignment"]
        for k in ("equilibrationSnapshotsA", "equilibrationSnapshotsP", "samplingSnapshotsA", "samplingSnapshotsP"):
            if k in raw:
                diag[k] = list(raw[k])
        for k in ("firstPass", "unmatchedPatterns", "clusteredPatterns", "CorrToMeanPattern", "subsets", "consensus"):
            if k in raw:
                diag[k] = raw[k]
        self.metadata = {"meanChiSq": raw.get("meanChiSq"), "seed": raw.get("seed"), "diagnostics": diag, "params": params}

    # R/methods-CogapsResult.R getters
    def getFeatureLoadings(self):
        return self.featureLoadings

    def getSampleFactors(self):
        return self.sampleFactors

    def getAmplitudeMatrix(self):
        return self.featureLoadings

    def getPatternMatrix(self):
        return self.sampleFactors

    def getMeanChiSq(self):
        return self.metadata["meanChiSq"]

    def getSubsets(self):
        return self.metadata["diagnostics"].get("subsets")

    def getUnmatchedPatterns(self):
        return self.metadata["diagnostics"].get("unmatchedPatterns")

    def getClusteredPatterns(self):
        return self.metadata["diagnostics"].get("clusteredPatterns")
