"""ctypes view of include/cogaps_hip.h.

`load()` returns the product library (libcogaps_hip.so, HIP/gfx950).  It fails loudly when the
library has not been built: there is no CPU fallback.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libcogaps_hip.so")

INTERRUPT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p)


class CogapsParamsC(C.Structure):
    _fields_ = [
        ("seed", C.c_uint32), ("nPatterns", C.c_uint32), ("nIterations", C.c_uint32),
        ("maxThreads", C.c_uint32), ("outputFrequency", C.c_uint32),
        ("checkpointInterval", C.c_uint32), ("snapshotFrequency", C.c_uint32),
        ("alphaA", C.c_float), ("alphaP", C.c_float),
        ("maxGibbsMassA", C.c_float), ("maxGibbsMassP", C.c_float),
        ("transposeData", C.c_int32), ("printMessages", C.c_int32),
        ("subsetData", C.c_int32), ("subsetGenes", C.c_int32),
        ("dataIndicesSubset", C.POINTER(C.c_uint32)), ("nSubset", C.c_uint32),
        ("useSparseOptimization", C.c_int32), ("takePumpSamples", C.c_int32),
        ("asynchronousUpdates", C.c_int32),
        ("whichMatrixFixed", C.c_char), ("fixedPatterns", C.POINTER(C.c_float)),
        ("fixedRows", C.c_uint32), ("workerID", C.c_uint32), ("runningDistributed", C.c_int32),
        ("device", C.c_int32), ("interrupt", INTERRUPT_FN), ("interruptArg", C.c_void_p),
        ("snapshotPhase", C.c_int32), ("pumpThreshold", C.c_int32), ("fixedCols", C.c_int32),
        ("reductionMode", C.c_int32), ("mathMode", C.c_int32),
    ]


class CogapsResultC(C.Structure):
    _fields_ = [
        ("nGenes", C.c_uint32), ("nSamples", C.c_uint32), ("nPatterns", C.c_uint32),
        ("Amean", C.POINTER(C.c_float)), ("Asd", C.POINTER(C.c_float)),
        ("Pmean", C.POINTER(C.c_float)), ("Psd", C.POINTER(C.c_float)),
        ("nHistory", C.c_uint32), ("chisqHistory", C.POINTER(C.c_float)),
        ("atomHistoryA", C.POINTER(C.c_uint32)), ("atomHistoryP", C.POINTER(C.c_uint32)),
        ("totalUpdates", C.c_uint64), ("seed", C.c_uint32), ("totalRunningTime", C.c_uint32),
        ("meanChiSq", C.c_float), ("averageQueueLengthA", C.c_float), ("averageQueueLengthP", C.c_float),
        ("samplerSeconds", C.c_double),
        ("pumpMatrix", C.POINTER(C.c_float)), ("meanPatternAssignment", C.POINTER(C.c_float)),
        ("nEquilibrationSnapshots", C.c_uint32), ("nSamplingSnapshots", C.c_uint32),
        ("equilibrationSnapshotsA", C.POINTER(C.c_float)), ("equilibrationSnapshotsP", C.POINTER(C.c_float)),
        ("samplingSnapshotsA", C.POINTER(C.c_float)), ("samplingSnapshotsP", C.POINTER(C.c_float)),
    ]


class CogapsPerfC(C.Structure):
    _fields_ = [
        ("evalBytes", C.c_uint64), ("evalLaunches", C.c_uint64), ("genLaunches", C.c_uint64),
        ("batches", C.c_uint64), ("proposalsQueued", C.c_uint64),
        ("evalMs", C.c_double), ("genMs", C.c_double), ("syncMs", C.c_double),
        ("evalNoopMs", C.c_double), ("genNoopMs", C.c_double), ("evalNoopTimed", C.c_uint64), ("genNoopTimed", C.c_uint64),
        ("timedBatches", C.c_uint64), ("evalTimed", C.c_uint64), ("genTimed", C.c_uint64), ("syncTimed", C.c_uint64), ("syncBytes", C.c_uint64),
    ]


TRACE_DTYPE = np.dtype([
    ("pos", "<u8"), ("rng_state", "<u8"), ("atom1", "<u4"), ("atom2", "<u4"),
    ("r1", "<u4"), ("c1", "<u4"), ("r2", "<u4"), ("c2", "<u4"), ("type", "<u4"), ("batch", "<u4"),
])

# every symbol include/cogaps_hip.h declares
EXPORTS = [
    "cogaps_default_params", "cogaps_run", "cogaps_result_free", "cogaps_last_error", "cogaps_last_error_code",
    "cogaps_build_report", "cogaps_source_hash", "cogaps_checkpoints_enabled", "cogaps_compiled_with_openmp",
    "cogaps_session_create", "cogaps_session_destroy", "cogaps_session_set_annealing",
    "cogaps_session_draw_steps", "cogaps_session_update", "cogaps_session_sync",
    "cogaps_session_iterate", "cogaps_session_run_iterations", "cogaps_session_natoms",
    "cogaps_session_chisq", "cogaps_session_get_matrix", "cogaps_session_get_ap",
    "cogaps_session_get_atoms", "cogaps_session_dims", "cogaps_session_avg_queue",
    "cogaps_session_finish", "cogaps_session_set_timing", "cogaps_session_perf",
    "cogaps_session_perf_sampler", "cogaps_session_chained", "cogaps_session_chain_recoveries", "cogaps_session_generator_window", "cogaps_session_launch_clock", "cogaps_session_launch_period", "cogaps_session_get_rows", "cogaps_sparse_width", "cogaps_reduction_width", "cogaps_session_debug_prof", "cogaps_session_debug_replay",
    "cogaps_run_from_file", "cogaps_read_matrix_file", "cogaps_read_matrix_file_subset", "cogaps_matrix_free", "cogaps_file_info", "cogaps_debug_math", "cogaps_current_device", "cogaps_device_memory",
    "cogaps_session_debug_check_domain", "cogaps_batch_create", "cogaps_batch_destroy", "cogaps_batch_run_iterations", "cogaps_batch_set_timing", "cogaps_batch_perf",
]

REDUCE_LANES, REDUCE_SEQ = 0, 1                              # cogaps_params.reductionMode
MATH_PORTABLE, MATH_GLIBC_FMA, MATH_GLIBC_SSE2 = 0, 1, 2     # cogaps_params.mathMode
_REDUCE = {"lanes": REDUCE_LANES, "seq": REDUCE_SEQ}
_MATH = {"portable": MATH_PORTABLE, "glibc-fma": MATH_GLIBC_FMA, "glibc-sse2": MATH_GLIBC_SSE2}
_PUMP = {"unique": 0, "cut": 1}

ERR_GENERIC, ERR_OUT_OF_DEVICE_MEMORY, ERR_OUT_OF_HOST_MEMORY = 1, 2, 3      # cogaps_last_error_code()


class CogapsError(RuntimeError):
    """a failing call of the C ABI: the library's message, and its kind as a code (include/cogaps_hip.h, cogaps_last_error_code)"""

    def __init__(self, message, code=ERR_GENERIC):
        super().__init__(message)
        self.code = code


class OutOfDeviceMemory(CogapsError):
    """hipErrorOutOfMemory inside the library (COGAPS_ERR_OUT_OF_DEVICE_MEMORY)"""


def _error(L, prefix=""):
    """the exception for the calling thread's last failing call"""
    code = int(L.cogaps_last_error_code())
    msg = prefix + L.cogaps_last_error().decode()
    return OutOfDeviceMemory(msg, code) if code == ERR_OUT_OF_DEVICE_MEMORY else CogapsError(msg, code)


def bind(L):
    """Attach prototypes to an opened library implementing include/cogaps_hip.h."""
    fp, u32p, vp = C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.c_void_p
    L.cogaps_default_params.argtypes = [C.POINTER(CogapsParamsC)]
    L.cogaps_default_params.restype = None
    L.cogaps_run.argtypes = [fp, C.c_uint32, C.c_uint32, C.POINTER(CogapsParamsC), fp, C.POINTER(CogapsResultC)]
    L.cogaps_result_free.argtypes = [C.POINTER(CogapsResultC)]
    L.cogaps_result_free.restype = None
    L.cogaps_last_error.restype = C.c_char_p
    L.cogaps_last_error_code.restype = C.c_int
    L.cogaps_build_report.restype = C.c_char_p
    L.cogaps_source_hash.restype = C.c_char_p
    L.cogaps_session_create.restype = vp
    L.cogaps_session_create.argtypes = [vp, C.c_uint32, C.c_uint32, C.POINTER(CogapsParamsC), vp, C.c_int]
    L.cogaps_session_destroy.argtypes = [vp]
    L.cogaps_session_destroy.restype = None
    L.cogaps_session_set_annealing.argtypes = [vp, C.c_float]
    L.cogaps_session_draw_steps.argtypes = [vp, u32p, u32p]
    L.cogaps_session_update.argtypes = [vp, C.c_char, C.c_uint32, vp, C.c_uint32, u32p, u32p, u32p, C.c_uint32, u32p]
    L.cogaps_session_sync.argtypes = [vp, C.c_char]
    L.cogaps_session_iterate.argtypes = [vp, C.c_uint32, C.c_uint32, C.c_int]
    L.cogaps_session_run_iterations.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.cogaps_session_natoms.argtypes = [vp, C.c_char, u32p]
    L.cogaps_session_chisq.argtypes = [vp, C.c_char, fp]
    L.cogaps_session_get_matrix.argtypes = [vp, C.c_char, fp]
    L.cogaps_session_get_ap.argtypes = [vp, C.c_char, fp]
    L.cogaps_session_get_atoms.argtypes = [vp, C.c_char, C.POINTER(C.c_uint64), fp, u32p, u32p]
    L.cogaps_session_dims.argtypes = [vp, C.c_char, u32p, u32p, u32p]
    L.cogaps_session_avg_queue.argtypes = [vp, C.c_char, fp]
    L.cogaps_session_finish.argtypes = [vp, C.POINTER(CogapsResultC)]
    L.cogaps_session_set_timing.argtypes = [vp, C.c_int]
    L.cogaps_session_perf.argtypes = [vp, C.POINTER(CogapsPerfC)]
    L.cogaps_session_perf_sampler.argtypes = [vp, C.c_char, C.POINTER(CogapsPerfC)]
    L.cogaps_session_chained.argtypes = [vp, C.c_char, C.POINTER(C.c_int)]
    L.cogaps_session_chain_recoveries.argtypes = [vp, C.c_char, C.POINTER(C.c_uint32)]
    L.cogaps_session_generator_window.argtypes = [vp, C.c_char, C.POINTER(C.c_uint32)]
    if hasattr(L, "cogaps_session_launch_period"):
        L.cogaps_session_launch_period.argtypes = [vp, C.c_char, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    if hasattr(L, "cogaps_session_launch_clock"):      # (an A/B build of an older source tree may lack it: launch_clock() then reports no launches)
        L.cogaps_session_launch_clock.argtypes = [vp, C.c_char, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.cogaps_session_debug_prof.argtypes = [vp, C.c_char, C.POINTER(C.c_uint64)]
    L.cogaps_session_debug_replay.argtypes = [vp, C.c_char, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_double)]
    L.cogaps_reduction_width.restype = C.c_uint32
    L.cogaps_sparse_width.restype = C.c_uint32
    L.cogaps_sparse_width.argtypes = [C.c_uint32]
    L.cogaps_session_get_rows.argtypes = [vp, C.c_char, fp]
    L.cogaps_reduction_width.argtypes = [C.c_uint32]
    L.cogaps_run_from_file.argtypes = [C.c_char_p, C.POINTER(CogapsParamsC), C.c_char_p, C.POINTER(CogapsResultC)]
    L.cogaps_read_matrix_file.argtypes = [C.c_char_p, u32p, u32p, C.POINTER(fp)]
    L.cogaps_read_matrix_file_subset.argtypes = [C.c_char_p, C.c_int, u32p, C.c_uint32, u32p, u32p, C.POINTER(fp)]
    L.cogaps_matrix_free.argtypes = [fp]
    L.cogaps_matrix_free.restype = None
    L.cogaps_file_info.argtypes = [C.c_char_p, u32p, u32p, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t), C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.cogaps_debug_math.argtypes = [C.c_int, C.c_int, fp, fp, C.c_uint32, C.c_int]
    L.cogaps_current_device.argtypes = [C.POINTER(C.c_int)]
    L.cogaps_device_memory.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.cogaps_session_debug_check_domain.argtypes = [vp, C.c_char, u32p]
    L.cogaps_batch_create.restype = vp
    L.cogaps_batch_create.argtypes = [C.POINTER(vp), C.c_uint32]
    L.cogaps_batch_destroy.argtypes = [vp]
    L.cogaps_batch_destroy.restype = None
    L.cogaps_batch_run_iterations.argtypes = [vp, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint64)]
    L.cogaps_batch_set_timing.argtypes = [vp, C.c_int]
    L.cogaps_batch_perf.argtypes = [vp, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    return L


_lib = None


def load():
    """The HIP library, or RuntimeError.  No fallback of any kind."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "cogaps_amd: %s is missing -- build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        # One HIP / HSA runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64; when the library
        # is loaded first it brings /opt/rocm's copies in, a later `import torch` adds the bundled ones, and whichever runtime touches
        # the device second finds it taken ("no ROCm-capable device is detected" -- in torch.cuda / RCCL, or here).  With torch
        # imported first the loader resolves this library's dependencies to the copies torch uses, so the front-end (device
        # tensors, torch.distributed over RCCL) and the sampler share one runtime.  Without PyTorch (a plain C / R client) the
        # library runs on the ROCm installation's runtime alone.
        # COGAPS_NO_TORCH=1 skips this (a Python client that never uses torch saves the import and stays on the ROCm installation's
        # runtime; it must then not import torch later in the same process).  A PyTorch that is not a ROCm build (CPU-only, CUDA)
        # bundles no HIP runtime: nothing to share, the library resolves against /opt/rocm as it would without PyTorch.
        import sys
        if "torch" in sys.modules or os.environ.get("COGAPS_NO_TORCH", "") in ("", "0"):
            try:
                import torch  # noqa: F401
            except ImportError:
                pass
        _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_params(L, nPatterns=3, nIterations=1000, seed=0, outputFrequency=500, nThreads=1,
                alphaA=0.01, alphaP=0.01, maxGibbsMassA=100.0, maxGibbsMassP=100.0,
                transposeData=False, subsetIndices=None, subsetDim=0, whichMatrixFixed="N",
                fixedPatterns=None, sparseOptimization=False, asynchronousUpdates=True,
                messages=False, workerID=1, device=-1, takePumpSamples=False,
                checkpointInterval=0, nSnapshots=0, snapshotPhase="sampling", snapshotFrequency=None,
                pumpThreshold="unique", reductionMode="lanes", mathMode="portable", runningDistributed=False):
    p = CogapsParamsC()
    L.cogaps_default_params(C.byref(p))
    p.nPatterns, p.nIterations, p.seed = int(nPatterns), int(nIterations), int(seed)
    p.outputFrequency, p.maxThreads = int(outputFrequency), int(nThreads)
    p.alphaA, p.alphaP = float(alphaA), float(alphaP)
    p.maxGibbsMassA, p.maxGibbsMassP = float(maxGibbsMassA), float(maxGibbsMassP)
    p.transposeData = int(bool(transposeData))
    p.printMessages = int(bool(messages)) if workerID == 1 else 0       # Cogaps.cpp:85
    p.useSparseOptimization = int(bool(sparseOptimization))
    p.asynchronousUpdates = int(bool(asynchronousUpdates))
    p.takePumpSamples = int(bool(takePumpSamples))
    p.checkpointInterval = int(checkpointInterval)
    # Cogaps.cpp:104-123: nSnapshots -> snapshotFrequency = nIterations / nSnapshots; phase names
    p.snapshotFrequency = int(snapshotFrequency) if snapshotFrequency is not None else (int(nIterations) // int(nSnapshots) if nSnapshots else 0)
    p.snapshotPhase = {"all": 0, "equilibration": 1, "sampling": 2}[snapshotPhase] if isinstance(snapshotPhase, str) else int(snapshotPhase)
    p.workerID = int(workerID)
    p.device = int(device)
    p.runningDistributed = int(bool(runningDistributed))
    p.pumpThreshold = _PUMP[pumpThreshold] if isinstance(pumpThreshold, str) else int(pumpThreshold)
    p.reductionMode = _REDUCE[reductionMode] if isinstance(reductionMode, str) else int(reductionMode)
    p.mathMode = _MATH[mathMode] if isinstance(mathMode, str) else int(mathMode)
    keep = []
    if subsetIndices is not None and subsetDim > 0:
        idx = np.ascontiguousarray(subsetIndices, dtype=np.uint32)
        keep.append(idx)
        p.subsetData = 1
        p.subsetGenes = 1 if subsetDim == 1 else 0
        p.dataIndicesSubset = idx.ctypes.data_as(C.POINTER(C.c_uint32))
        p.nSubset = idx.size
        p.runningDistributed = 1
    p.whichMatrixFixed = str(whichMatrixFixed).encode()[:1]
    if fixedPatterns is not None and whichMatrixFixed != "N":
        fx = np.ascontiguousarray(fixedPatterns, dtype=np.float32)
        keep.append(fx)
        if fx.ndim != 2 or fx.shape[1] != int(nPatterns):
            raise ValueError("fixedPatterns must have nPatterns = %d columns" % int(nPatterns))
        p.fixedPatterns = _fp(fx)
        p.fixedRows = fx.shape[0]
        p.fixedCols = fx.shape[1]
    p._keep = keep
    return p


def result_to_dict(L, r):
    g, s, k, h = r.nGenes, r.nSamples, r.nPatterns, r.nHistory
    out = {
        "Amean": np.ctypeslib.as_array(r.Amean, shape=(g, k)).copy(),
        "Asd": np.ctypeslib.as_array(r.Asd, shape=(g, k)).copy(),
        "Pmean": np.ctypeslib.as_array(r.Pmean, shape=(s, k)).copy(),
        "Psd": np.ctypeslib.as_array(r.Psd, shape=(s, k)).copy(),
        "chisq": np.ctypeslib.as_array(r.chisqHistory, shape=(max(h, 1),))[:h].copy(),
        "atomsA": np.ctypeslib.as_array(r.atomHistoryA, shape=(max(h, 1),))[:h].copy(),
        "atomsP": np.ctypeslib.as_array(r.atomHistoryP, shape=(max(h, 1),))[:h].copy(),
        "totalUpdates": int(r.totalUpdates), "meanChiSq": float(r.meanChiSq), "seed": int(r.seed),
        "averageQueueLengthA": float(r.averageQueueLengthA),
        "averageQueueLengthP": float(r.averageQueueLengthP),
        "totalRunningTime": int(r.totalRunningTime), "samplerSeconds": float(r.samplerSeconds),
    }

    def _arr(ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape).copy() if ptr else np.zeros(shape, dtype=np.float32)
    if r.pumpMatrix:
        out["pumpMatrix"] = _arr(r.pumpMatrix, (g, k)); out["meanPatternAssignment"] = _arr(r.meanPatternAssignment, (g, k))
    ne, ns = int(r.nEquilibrationSnapshots), int(r.nSamplingSnapshots)
    out["equilibrationSnapshotsA"] = _arr(r.equilibrationSnapshotsA, (ne, g, k)); out["equilibrationSnapshotsP"] = _arr(r.equilibrationSnapshotsP, (ne, s, k))
    out["samplingSnapshotsA"] = _arr(r.samplingSnapshotsA, (ns, g, k)); out["samplingSnapshotsP"] = _arr(r.samplingSnapshotsP, (ns, s, k))
    L.cogaps_result_free(C.byref(r))
    return out


class Session:
    """One sampler run, one step at a time (cogaps_session_* of include/cogaps_hip.h)."""

    def __init__(self, data, unc=None, lib=None, **kw):
        self.L = lib if lib is not None else load()
        self.d = np.ascontiguousarray(data, dtype=np.float32)
        self.u = None if unc is None else np.ascontiguousarray(unc, dtype=np.float32)
        self.p = make_params(self.L, **kw)
        self.h = self.L.cogaps_session_create(self.d.ctypes.data, self.d.shape[0], self.d.shape[1], C.byref(self.p),
                                              None if self.u is None else self.u.ctypes.data, 0)
        if not self.h:
            raise _error(self.L, "cogaps_session_create: ")

    def _ck(self, rc):
        if rc:
            raise _error(self.L)

    def close(self):
        if getattr(self, "h", None):
            self.L.cogaps_session_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_annealing(self, t):
        self._ck(self.L.cogaps_session_set_annealing(self.h, t))

    def draw_steps(self):
        a, b = C.c_uint32(), C.c_uint32()
        self._ck(self.L.cogaps_session_draw_steps(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def update(self, which, nsteps, trace_cap=0):
        if not trace_cap:
            self._ck(self.L.cogaps_session_update(self.h, which.encode(), nsteps, None, 0, None, None, None, 0, None))
            return None
        rec = np.zeros(trace_cap, dtype=TRACE_DTYPE)
        bn = np.zeros(trace_cap, dtype=np.uint32)
        bq = np.zeros(trace_cap, dtype=np.uint32)
        nt, nb = C.c_uint32(), C.c_uint32()
        u32p = C.POINTER(C.c_uint32)
        self._ck(self.L.cogaps_session_update(self.h, which.encode(), nsteps, rec.ctypes.data, trace_cap, C.byref(nt),
                                              bn.ctypes.data_as(u32p), bq.ctypes.data_as(u32p), trace_cap, C.byref(nb)))
        assert nt.value <= trace_cap and nb.value <= trace_cap, "trace overflow"
        return {"rec": rec[:nt.value], "nproc": bn[:nb.value], "qlen": bq[:nb.value]}

    def sync(self, which):
        self._ck(self.L.cogaps_session_sync(self.h, which.encode()))

    def iterate(self, nA, nP, sampling=False):
        self._ck(self.L.cogaps_session_iterate(self.h, nA, nP, int(sampling)))

    def run_iterations(self, phase, first, n):
        upd = C.c_uint64(0)
        self._ck(self.L.cogaps_session_run_iterations(self.h, phase, first, n, C.byref(upd)))
        return upd.value

    def natoms(self, which):
        n = C.c_uint32()
        self._ck(self.L.cogaps_session_natoms(self.h, which.encode(), C.byref(n)))
        return n.value

    def chisq(self, which):
        c = C.c_float()
        self._ck(self.L.cogaps_session_chisq(self.h, which.encode(), C.byref(c)))
        return c.value

    def dims(self, which):
        m, n, k = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._ck(self.L.cogaps_session_dims(self.h, which.encode(), C.byref(m), C.byref(n), C.byref(k)))
        return m.value, n.value, k.value

    def matrix(self, which):
        m, n, k = self.dims(which)
        out = np.zeros((m, k), dtype=np.float32)
        self._ck(self.L.cogaps_session_get_matrix(self.h, which.encode(), _fp(out)))
        return out

    def rows(self, which):
        m, n, k = self.dims(which)
        out = np.zeros((m, k), dtype=np.float32)
        self._ck(self.L.cogaps_session_get_rows(self.h, which.encode(), _fp(out)))
        return out

    def ap(self, which):
        m, n, k = self.dims(which)
        out = np.zeros((m, n), dtype=np.float32)
        self._ck(self.L.cogaps_session_get_ap(self.h, which.encode(), _fp(out)))
        return out

    def atoms(self, which, n=None):
        # size query: the library reports the current count through natoms after an update;
        # before any update it is 0
        cap = self.dims(which)[0] * self.dims(which)[2] + (1 << 17)
        pos = np.zeros(cap, dtype=np.uint64)
        mass = np.zeros(cap, dtype=np.float32)
        left = np.zeros(cap, dtype=np.uint32)
        right = np.zeros(cap, dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self._ck(self.L.cogaps_session_get_atoms(self.h, which.encode(), pos.ctypes.data_as(C.POINTER(C.c_uint64)), _fp(mass),
                                                 left.ctypes.data_as(u32p), right.ctypes.data_as(u32p)))
        n = self.natoms(which) if n is None else n
        return {"pos": pos[:n].copy(), "mass": mass[:n].copy(), "left": left[:n].copy(), "right": right[:n].copy()}

    def avg_queue(self, which):
        a = C.c_float()
        self._ck(self.L.cogaps_session_avg_queue(self.h, which.encode(), C.byref(a)))
        return a.value

    def set_timing(self, on):
        self._ck(self.L.cogaps_session_set_timing(self.h, int(on)))

    def launch_period(self, which):
        """the same launches by their period (entry to the next launch's entry: dispatcher start-up and end-of-kernel write-back included)"""
        m, n = C.c_double(0), C.c_uint64(0)
        pc = (C.c_double * 5)()
        if not hasattr(self.L, "cogaps_session_launch_period"):
            return {"mean_us": 0.0, "p10_us": 0.0, "p50_us": 0.0, "p75_us": 0.0, "p90_us": 0.0, "p99_us": 0.0, "launches": 0}
        self._ck(self.L.cogaps_session_launch_period(self.h, which.encode(), C.byref(m), pc, C.byref(n)))
        return {"mean_us": m.value, "p10_us": pc[0], "p50_us": pc[1], "p75_us": pc[2], "p90_us": pc[3], "p99_us": pc[4], "launches": int(n.value)}

    def launch_clock(self, which):
        """durations of the sampler's chained launches since set_timing(1), every launch (device clock): mean, percentiles, count"""
        m, n = C.c_double(0), C.c_uint64(0)
        pc = (C.c_double * 5)()
        if not hasattr(self.L, "cogaps_session_launch_clock"):
            return {"mean_us": 0.0, "p10_us": 0.0, "p50_us": 0.0, "p75_us": 0.0, "p90_us": 0.0, "p99_us": 0.0, "launches": 0}
        self._ck(self.L.cogaps_session_launch_clock(self.h, which.encode(), C.byref(m), pc, C.byref(n)))
        return {"mean_us": m.value, "p10_us": pc[0], "p50_us": pc[1], "p75_us": pc[2], "p90_us": pc[3], "p99_us": pc[4], "launches": int(n.value)}

    def chained(self, which):
        """whether the sampler's last update ran as chained launches (one launch per batch: csrc/chain_kernel.h)"""
        v = C.c_int()
        self._ck(self.L.cogaps_session_chained(self.h, which.encode(), C.byref(v)))
        return bool(v.value)

    def generator_window(self, which):
        """attempts per round of the sampler's generator launches as of its last update"""
        v = C.c_uint32()
        self._ck(self.L.cogaps_session_generator_window(self.h, which.encode(), C.byref(v)))
        return v.value

    def chain_recoveries(self, which):
        """how often a hand-over inside a chained launch of this sampler never arrived and the batch was completed by the recovery (the sampler
        keeps two launches per batch from the first such event on)"""
        v = C.c_uint32()
        self._ck(self.L.cogaps_session_chain_recoveries(self.h, which.encode(), C.byref(v)))
        return int(v.value)

    def perf(self, which=None):
        p = CogapsPerfC()
        if which is None:
            self._ck(self.L.cogaps_session_perf(self.h, C.byref(p)))
        else:
            self._ck(self.L.cogaps_session_perf_sampler(self.h, which.encode(), C.byref(p)))
        return {f[0]: getattr(p, f[0]) for f in CogapsPerfC._fields_}

    def debug_replay(self, which, kind, n, flags=0):
        us = C.c_double()
        self._ck(self.L.cogaps_session_debug_replay(self.h, which.encode(), kind, n, flags, C.byref(us)))
        return us.value

    def check_domain(self, which):
        """number of broken invariants of the atomic domain's redundant state (0 = consistent)"""
        v = C.c_uint32()
        self._ck(self.L.cogaps_session_debug_check_domain(self.h, which.encode(), C.byref(v)))
        return v.value

    def debug_prof(self, which):
        out = (C.c_uint64 * 16)()
        self._ck(self.L.cogaps_session_debug_prof(self.h, which.encode(), out))
        return list(out)

    def finish(self):
        r = CogapsResultC()
        self._ck(self.L.cogaps_session_finish(self.h, C.byref(r)))
        return result_to_dict(self.L, r)


def read_matrix_file(path, lib=None, rows=None, cols=None):
    """The library's own reader (csrc/file_reader.h): the file as a dense fp32 matrix.  Host only.  rows / cols (one of them): 1-based
    indices of the rows / columns to keep -- only that part of the file is materialised, in SORTED index order, as the reference's
    workers read their subset of a file (Matrix.cpp:70-134)."""
    L = lib or load()
    nr, nc, ptr = C.c_uint32(), C.c_uint32(), C.POINTER(C.c_float)()
    if rows is not None and cols is not None:
        raise ValueError("rows or cols, not both")
    if rows is None and cols is None:
        rc = L.cogaps_read_matrix_file(os.fsencode(path), C.byref(nr), C.byref(nc), C.byref(ptr))
    else:
        idx = np.ascontiguousarray(rows if rows is not None else cols, dtype=np.uint32)
        rc = L.cogaps_read_matrix_file_subset(os.fsencode(path), int(rows is not None), idx.ctypes.data_as(C.POINTER(C.c_uint32)), idx.size, C.byref(nr), C.byref(nc), C.byref(ptr))
    if rc:
        raise _error(L)
    try:
        return np.ctypeslib.as_array(ptr, shape=(nr.value, nc.value)).copy() if nr.value * nc.value else np.zeros((nr.value, nc.value), np.float32)
    finally:
        L.cogaps_matrix_free(ptr)


def file_info(path, lib=None):
    """getFileInfo_cpp: (nrow, ncol, rowNames, colNames)"""
    L = lib or load()
    nr, nc, rn, cn = C.c_uint32(), C.c_uint32(), C.c_size_t(), C.c_size_t()
    if L.cogaps_file_info(os.fsencode(path), C.byref(nr), C.byref(nc), None, 0, C.byref(rn), None, 0, C.byref(cn)):
        raise _error(L)
    rb, cb = C.create_string_buffer(rn.value), C.create_string_buffer(cn.value)
    if L.cogaps_file_info(os.fsencode(path), C.byref(nr), C.byref(nc), rb, rn.value, None, cb, cn.value, None):
        raise _error(L)
    names = lambda b: b.value.decode().split("\n") if b.value else []
    return nr.value, nc.value, names(rb), names(cb)


def run_from_file(path, unc_path=None, lib=None, **kw):
    """cogaps_run_from_file: the reference's file entry point, parsed natively."""
    L = lib if lib is not None else load()
    params = make_params(L, **kw)
    res = CogapsResultC()
    if L.cogaps_run_from_file(os.fsencode(path), C.byref(params), os.fsencode(unc_path) if unc_path else None, C.byref(res)):
        raise _error(L, "cogaps_run_from_file: ")
    return result_to_dict(L, res)


def run(data, unc=None, lib=None, **kw):
    """cogaps_run: one full equilibration + sampling run."""
    L = lib if lib is not None else load()
    d = np.ascontiguousarray(data, dtype=np.float32)
    u = None if unc is None else np.ascontiguousarray(unc, dtype=np.float32)
    p = make_params(L, **kw)
    r = CogapsResultC()
    rc = L.cogaps_run(_fp(d), d.shape[0], d.shape[1], C.byref(p), None if u is None else _fp(u), C.byref(r))
    if rc:
        raise _error(L, "cogaps_run: ")
    return result_to_dict(L, r)


def debug_math(fn, x, mathMode="portable", on_device=False, lib=None):
    """logf (fn = "log") / expf ("exp") in a math mode of the library, on the device or from the same source on the host"""
    L = lib if lib is not None else load()
    xs = np.ascontiguousarray(x, dtype=np.float32)
    ys = np.zeros_like(xs)
    if L.cogaps_debug_math({"log": 0, "exp": 1}[fn], _MATH[mathMode] if isinstance(mathMode, str) else int(mathMode), _fp(xs), _fp(ys), xs.size, int(on_device)):
        raise _error(L)
    return ys


def device_memory(device=-1, lib=None):
    """(free, total) bytes of HBM on `device` (-1: the calling thread's current one)"""
    L = lib if lib is not None else load()
    f, t = C.c_uint64(0), C.c_uint64(0)
    if L.cogaps_device_memory(int(device), C.byref(f), C.byref(t)):
        raise _error(L)
    return int(f.value), int(t.value)


def current_device(lib=None):
    """hipGetDevice for the calling host thread"""
    L = lib if lib is not None else load()
    d = C.c_int(0)
    if L.cogaps_current_device(C.byref(d)):
        raise _error(L)
    return d.value


class Batch:
    """Batched multi-chain launches (cogaps_batch_* of include/cogaps_hip.h): the given sessions stepped in lock-step by one stream,
    one generator / evaluation launch for all of them.  Every chain gives the bits it gives on its own."""

    def __init__(self, sessions):
        self.sessions = list(sessions)
        self.L = self.sessions[0].L
        arr = (C.c_void_p * len(self.sessions))(*[s.h for s in self.sessions])
        self.h = self.L.cogaps_batch_create(arr, len(self.sessions))
        if not self.h:
            raise _error(self.L, "cogaps_batch_create: ")

    def _ck(self, rc):
        if rc:
            raise _error(self.L)

    def run_iterations(self, phase, first, n):
        upd = (C.c_uint64 * len(self.sessions))()
        self._ck(self.L.cogaps_batch_run_iterations(self.h, phase, first, n, upd))
        return [int(u) for u in upd]

    def set_timing(self, on):
        self._ck(self.L.cogaps_batch_set_timing(self.h, int(on)))

    def perf(self, side):
        g, e, n, l = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
        self._ck(self.L.cogaps_batch_perf(self.h, {"A": 0, "P": 1}[side], C.byref(g), C.byref(e), C.byref(n), C.byref(l)))
        return {"gen_us": g.value, "eval_us": e.value, "sampled": n.value, "launches": l.value}

    def close(self):
        if getattr(self, "h", None):
            self.L.cogaps_batch_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def run_batch(datas, uncs=None, lib=None, kws=None, **common):
    """cogaps_run for several chains at once through the batched launches: datas[i] with the keyword arguments kws[i] (merged over
    `common`) -- or ONE iterable of (data, uncertainty, keyword arguments) triples, consumed one at a time: a chain's host matrices
    are released as soon as its session holds them in HBM.  All chains need the same nIterations.  Returns the result dicts in order."""
    L = lib if lib is not None else load()
    if uncs is None and kws is None and not isinstance(datas, (list, tuple)):
        triples = datas
    else:
        kws = kws or [{} for _ in datas]
        uncs = uncs or [None] * len(datas)
        triples = zip(datas, uncs, kws)
    ss, b = [], None
    try:
        for d, u, k in triples:
            s_ = Session(d, unc=u, lib=L, **dict(common, **k))
            s_.d = s_.u = None                # the library copied them at creation (cogaps_session_create)
            ss.append(s_)
            del d, u
        n_iter = {int(s.p.nIterations) for s in ss}
        if len(n_iter) != 1:
            raise ValueError("the chains of a batch need the same nIterations")
        n_iter = n_iter.pop()
        b = Batch(ss)
        b.run_iterations(1, 0, n_iter)
        b.run_iterations(2, 0, n_iter)
        return [s.finish() for s in ss]
    finally:
        if b is not None:
            b.close()
        for s in ss:
            s.close()
