"""cogaps_amd -- the CoGAPS asynchronous Gibbs sampler hot path on MI355X (HIP, gfx950).

Public surface mirrors the reference's: CoGAPS(), GWCoGAPS(), scCoGAPS(), CogapsParams, CogapsResult,
buildReport(), checkpointsEnabled(), compiledWithOpenMPSupport().  All compute goes through
csrc/libcogaps_hip.so (include/cogaps_hip.h); importing works without a GPU, running does not."""
from .params import CogapsParams
from .result import CogapsResult
from .api import CoGAPS, GWCoGAPS, scCoGAPS


def buildReport():
    from . import _capi
    return _capi.load().cogaps_build_report().decode()


def checkpointsEnabled():
    return False


def compiledWithOpenMPSupport():
    from . import _capi
    return bool(_capi.load().cogaps_compiled_with_openmp())


__all__ = ["CoGAPS", "GWCoGAPS", "scCoGAPS", "CogapsParams", "CogapsResult", "buildReport", "checkpointsEnabled",
           "compiledWithOpenMPSupport"]
