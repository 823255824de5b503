"""ctypes binding of the CPU oracle (oracle/gaps_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under cogaps_amd/ may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))

MATH_LIBM = 0
MATH_PORTABLE = 1
MATH_GLIBC_FMA = 2      # glibc 2.35 logf / expf restated, -mfma ifunc variant
MATH_GLIBC_SSE2 = 3     # ... generic variant


class GoParams(C.Structure):
    _fields_ = [
        ("nPatterns", C.c_uint32), ("nIterations", C.c_uint32), ("seed", C.c_uint32),
        ("outputFrequency", C.c_uint32), ("maxThreads", C.c_uint32),
        ("alphaA", C.c_float), ("alphaP", C.c_float),
        ("maxGibbsMassA", C.c_float), ("maxGibbsMassP", C.c_float),
        ("transposeData", C.c_int32), ("subsetData", C.c_int32), ("subsetGenes", C.c_int32),
        ("subsetIndices", C.POINTER(C.c_uint32)), ("nSubset", C.c_uint32),
        ("whichMatrixFixed", C.c_char), ("fixedPatterns", C.POINTER(C.c_float)),
        ("fixedRows", C.c_uint32), ("math_mode", C.c_int32),
        ("redW_A", C.c_uint32), ("redW_P", C.c_uint32), ("redG", C.c_uint32),
        ("useSparseOptimization", C.c_int32), ("takePumpSamples", C.c_int32),
        ("snapshotFrequency", C.c_uint32), ("snapshotPhase", C.c_int32),
    ]


class GoResult(C.Structure):
    _fields_ = [
        ("nGenes", C.c_uint32), ("nSamples", C.c_uint32), ("nPatterns", C.c_uint32),
        ("Amean", C.POINTER(C.c_float)), ("Asd", C.POINTER(C.c_float)),
        ("Pmean", C.POINTER(C.c_float)), ("Psd", C.POINTER(C.c_float)),
        ("nHistory", C.c_uint32), ("chisqHistory", C.POINTER(C.c_float)),
        ("atomHistoryA", C.POINTER(C.c_uint32)), ("atomHistoryP", C.POINTER(C.c_uint32)),
        ("totalUpdates", C.c_uint64), ("meanChiSq", C.c_float),
        ("averageQueueLengthA", C.c_float), ("averageQueueLengthP", C.c_float),
        ("samplerSeconds", C.c_double),
        ("pumpMatrix", C.POINTER(C.c_float)), ("meanPatternAssignment", C.POINTER(C.c_float)),
        ("nEquilibrationSnapshots", C.c_uint32), ("nSamplingSnapshots", C.c_uint32),
        ("equilibrationSnapshotsA", C.POINTER(C.c_float)), ("equilibrationSnapshotsP", C.POINTER(C.c_float)),
        ("samplingSnapshotsA", C.POINTER(C.c_float)), ("samplingSnapshotsP", C.POINTER(C.c_float)),
    ]


class GoTraceRec(C.Structure):
    _fields_ = [
        ("pos", C.c_uint64), ("rng_state", C.c_uint64),
        ("atom1", C.c_uint32), ("atom2", C.c_uint32),
        ("r1", C.c_uint32), ("c1", C.c_uint32), ("r2", C.c_uint32), ("c2", C.c_uint32),
        ("type", C.c_uint32), ("batch", C.c_uint32),
    ]


TRACE_DTYPE = np.dtype([
    ("pos", "<u8"), ("rng_state", "<u8"), ("atom1", "<u4"), ("atom2", "<u4"),
    ("r1", "<u4"), ("c1", "<u4"), ("r2", "<u4"), ("c2", "<u4"), ("type", "<u4"), ("batch", "<u4"),
])


class GoTrace(C.Structure):
    _fields_ = [
        ("rec", C.POINTER(GoTraceRec)), ("cap", C.c_uint32), ("n", C.c_uint32),
        ("batch_nproc", C.POINTER(C.c_uint32)), ("batch_qlen", C.POINTER(C.c_uint32)),
        ("batch_cap", C.c_uint32), ("n_batches", C.c_uint32),
    ]


def build(force=False):
    """Compile the oracle shared objects (gcc).  Building the checker is not using it."""
    targets = [os.path.join(_HERE, "liboracle.so"), os.path.join(_HERE, "liboracle_omp.so")]
    src = os.path.join(_HERE, "gaps_oracle.c")
    stale = force or any((not os.path.exists(t)) or os.path.getmtime(t) < os.path.getmtime(src) for t in targets)
    if stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s", "all"])
    return targets


_libs = {}


def lib(omp=False):
    key = bool(omp)
    if key in _libs:
        return _libs[key]
    build()
    L = C.CDLL(os.path.join(_HERE, "liboracle_omp.so" if omp else "liboracle.so"))
    fp = C.POINTER(C.c_float)
    L.go_default_params.argtypes = [C.POINTER(GoParams)]
    L.go_create.restype = C.c_void_p
    L.go_create.argtypes = [fp, C.c_uint32, C.c_uint32, C.POINTER(GoParams), fp]
    L.go_destroy.argtypes = [C.c_void_p]
    L.go_run.argtypes = [fp, C.c_uint32, C.c_uint32, C.POINTER(GoParams), fp, C.POINTER(GoResult)]
    L.go_result_free.argtypes = [C.POINTER(GoResult)]
    L.go_set_annealing.argtypes = [C.c_void_p, C.c_float]
    L.go_natoms.restype = C.c_uint32
    L.go_natoms.argtypes = [C.c_void_p, C.c_char]
    L.go_draw_steps.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.go_update.argtypes = [C.c_void_p, C.c_char, C.c_uint32, C.POINTER(GoTrace)]
    L.go_sync.argtypes = [C.c_void_p, C.c_char]
    L.go_iterate.restype = C.c_uint64
    L.go_iterate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32]
    L.go_stats_update.argtypes = [C.c_void_p]
    L.go_chisq.restype = C.c_float
    L.go_chisq.argtypes = [C.c_void_p, C.c_char]
    L.go_get_matrix.argtypes = [C.c_void_p, C.c_char, fp]
    L.go_get_ap.argtypes = [C.c_void_p, C.c_char, fp]
    L.go_get_rows.argtypes = [C.c_void_p, C.c_char, fp]
    L.go_debug_set_matrices.argtypes = [C.c_void_p, fp, fp]
    L.go_import_state.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), fp, C.c_uint32, fp, C.POINTER(C.c_uint64), fp, C.c_uint32, fp]
    L.go_debug_alpha.argtypes = [C.c_void_p, C.c_char, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, fp]
    L.go_get_atoms.argtypes = [C.c_void_p, C.c_char, C.POINTER(C.c_uint64), fp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.go_get_dims.argtypes = [C.c_void_p, C.c_char] + [C.POINTER(C.c_uint32)] * 3
    for n in ("go_lambda", "go_max_gibbs_mass", "go_avg_queue"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_void_p, C.c_char]
    L.go_get_luts.argtypes = [C.c_void_p, fp, fp, fp]
    L.go_finish.argtypes = [C.c_void_p, C.POINTER(GoResult)]
    L.go_run_iterations.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.go_portable_logf.restype = C.c_float
    L.go_portable_logf.argtypes = [C.c_float]
    L.go_portable_expf.restype = C.c_float
    L.go_portable_expf.argtypes = [C.c_float]
    L.go_build_luts.argtypes = [fp, fp, fp]
    L.go_seeder_stream.restype = C.c_uint64
    L.go_seeder_stream.argtypes = [C.c_uint32, C.POINTER(C.c_uint64), C.c_uint64]
    L.go_pcg_next.restype = C.c_uint32
    L.go_pcg_next.argtypes = [C.POINTER(C.c_uint64)]
    L.go_glibc_mismatches.restype = C.c_uint64
    L.go_glibc_mismatches.argtypes = [C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
    for n in ("go_glibc_logf", "go_glibc_expf"):
        getattr(L, n).restype = C.c_float
        getattr(L, n).argtypes = [C.c_float, C.c_int]
    L.go_strtof.restype = C.c_float
    L.go_strtof.argtypes = [C.c_char_p]
    _libs[key] = L
    return L


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def make_params(nPatterns=3, nIterations=1000, seed=0, outputFrequency=500, maxThreads=1,
                alphaA=0.01, alphaP=0.01, maxGibbsMassA=100.0, maxGibbsMassP=100.0,
                transposeData=False, subsetIndices=None, subsetDim=0,
                whichMatrixFixed="N", fixedPatterns=None, math_mode=MATH_LIBM,
                redW_A=1, redW_P=1, redG=1, sparseOptimization=False, takePumpSamples=False,
                snapshotFrequency=0, snapshotPhase=0):
    p = GoParams()
    lib().go_default_params(C.byref(p))
    p.nPatterns, p.nIterations, p.seed = nPatterns, nIterations, seed
    p.outputFrequency, p.maxThreads = outputFrequency, maxThreads
    p.alphaA, p.alphaP, p.maxGibbsMassA, p.maxGibbsMassP = alphaA, alphaP, maxGibbsMassA, maxGibbsMassP
    p.transposeData = int(bool(transposeData))
    keep = []
    if subsetIndices is not None and subsetDim > 0:
        idx = np.ascontiguousarray(subsetIndices, dtype=np.uint32)
        keep.append(idx)
        p.subsetData = 1
        p.subsetGenes = 1 if subsetDim == 1 else 0
        p.subsetIndices = idx.ctypes.data_as(C.POINTER(C.c_uint32))
        p.nSubset = idx.size
    p.whichMatrixFixed = whichMatrixFixed.encode()
    if fixedPatterns is not None:
        fx = np.ascontiguousarray(fixedPatterns, dtype=np.float32)
        keep.append(fx)
        p.fixedPatterns = _fp(fx)
        p.fixedRows = fx.shape[0]
    p.math_mode = math_mode
    p.redW_A, p.redW_P, p.redG = redW_A, redW_P, redG
    p.useSparseOptimization = int(bool(sparseOptimization))
    p.takePumpSamples = int(bool(takePumpSamples))
    p.snapshotFrequency, p.snapshotPhase = int(snapshotFrequency), int(snapshotPhase)
    p._keep = keep
    return p


def _result_to_dict(r):
    g, s, k, h = r.nGenes, r.nSamples, r.nPatterns, r.nHistory
    out = {
        "Amean": np.ctypeslib.as_array(r.Amean, shape=(g, k)).copy(),
        "Asd": np.ctypeslib.as_array(r.Asd, shape=(g, k)).copy(),
        "Pmean": np.ctypeslib.as_array(r.Pmean, shape=(s, k)).copy(),
        "Psd": np.ctypeslib.as_array(r.Psd, shape=(s, k)).copy(),
        "chisq": np.ctypeslib.as_array(r.chisqHistory, shape=(max(h, 1),))[:h].copy(),
        "atomsA": np.ctypeslib.as_array(r.atomHistoryA, shape=(max(h, 1),))[:h].copy(),
        "atomsP": np.ctypeslib.as_array(r.atomHistoryP, shape=(max(h, 1),))[:h].copy(),
        "totalUpdates": int(r.totalUpdates), "meanChiSq": float(r.meanChiSq),
        "averageQueueLengthA": float(r.averageQueueLengthA),
        "averageQueueLengthP": float(r.averageQueueLengthP),
        "samplerSeconds": float(r.samplerSeconds),
    }
    def _arr(ptr, shape):
        n = int(np.prod(shape))
        return np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].reshape(shape).copy() if ptr else np.zeros(shape, dtype=np.float32)
    if r.pumpMatrix:
        out["pumpMatrix"] = _arr(r.pumpMatrix, (g, k)); out["meanPatternAssignment"] = _arr(r.meanPatternAssignment, (g, k))
    ne, ns = int(r.nEquilibrationSnapshots), int(r.nSamplingSnapshots)
    out["equilibrationSnapshotsA"] = _arr(r.equilibrationSnapshotsA, (ne, g, k)); out["equilibrationSnapshotsP"] = _arr(r.equilibrationSnapshotsP, (ne, s, k))
    out["samplingSnapshotsA"] = _arr(r.samplingSnapshotsA, (ns, g, k)); out["samplingSnapshotsP"] = _arr(r.samplingSnapshotsP, (ns, s, k))
    return out


def run(data, unc=None, omp=False, **kw):
    """Full run (gaps::run).  data: 2-D array (genes x samples unless transposeData)."""
    L = lib(omp)
    d = np.ascontiguousarray(data, dtype=np.float32)
    u = None if unc is None else np.ascontiguousarray(unc, dtype=np.float32)
    p = make_params(**kw)
    r = GoResult()
    L.go_run(_fp(d), d.shape[0], d.shape[1], C.byref(p), None if u is None else _fp(u), C.byref(r))
    out = _result_to_dict(r)
    L.go_result_free(C.byref(r))
    return out


class Session:
    """Step-wise oracle session for batch-level parity tests."""

    def __init__(self, data, unc=None, omp=False, **kw):
        self.L = lib(omp)
        self.d = np.ascontiguousarray(data, dtype=np.float32)
        self.u = None if unc is None else np.ascontiguousarray(unc, dtype=np.float32)
        self.p = make_params(**kw)
        self.h = self.L.go_create(_fp(self.d), self.d.shape[0], self.d.shape[1], C.byref(self.p),
                                  None if self.u is None else _fp(self.u))

    def close(self):
        if self.h:
            self.L.go_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_annealing(self, t):
        self.L.go_set_annealing(self.h, t)

    def natoms(self, which):
        return self.L.go_natoms(self.h, which.encode())

    def draw_steps(self):
        a, b = C.c_uint32(), C.c_uint32()
        self.L.go_draw_steps(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def update(self, which, nsteps, trace_cap=0):
        if not trace_cap:
            self.L.go_update(self.h, which.encode(), nsteps, None)
            return None
        rec = np.zeros(trace_cap, dtype=TRACE_DTYPE)
        bn = np.zeros(trace_cap, dtype=np.uint32)
        bq = np.zeros(trace_cap, dtype=np.uint32)
        t = GoTrace()
        t.rec = rec.ctypes.data_as(C.POINTER(GoTraceRec))
        t.cap = trace_cap
        t.batch_nproc = bn.ctypes.data_as(C.POINTER(C.c_uint32))
        t.batch_qlen = bq.ctypes.data_as(C.POINTER(C.c_uint32))
        t.batch_cap = trace_cap
        self.L.go_update(self.h, which.encode(), nsteps, C.byref(t))
        assert t.n <= trace_cap and t.n_batches <= trace_cap, "trace overflow"
        return {"rec": rec[:t.n], "nproc": bn[:t.n_batches], "qlen": bq[:t.n_batches]}

    def sync(self, which):
        self.L.go_sync(self.h, which.encode())

    def iterate(self, nA, nP):
        return self.L.go_iterate(self.h, nA, nP)

    def run_iterations(self, phase, first, n):
        """iterations [first, first + n) of a phase as go_run runs them; returns the Poisson step counts (nA[], nP[])"""
        a = np.zeros(n, dtype=np.uint32)
        b = np.zeros(n, dtype=np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self.L.go_run_iterations(self.h, phase, first, n, a.ctypes.data_as(u32p), b.ctypes.data_as(u32p))
        return a, b

    def stats_update(self):
        self.L.go_stats_update(self.h)

    def chisq(self, which):
        return self.L.go_chisq(self.h, which.encode())

    def dims(self, which):
        m, n, k = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self.L.go_get_dims(self.h, which.encode(), C.byref(m), C.byref(n), C.byref(k))
        return m.value, n.value, k.value

    def matrix(self, which):
        m, n, k = self.dims(which)
        out = np.zeros((m, k), dtype=np.float32)
        self.L.go_get_matrix(self.h, which.encode(), _fp(out))
        return out

    def debug_set_matrices(self, A, P):
        a = np.ascontiguousarray(A, dtype=np.float32); b = np.ascontiguousarray(P, dtype=np.float32)
        self.L.go_debug_set_matrices(self.h, _fp(a), _fp(b))

    def import_state(self, atomsA, A, atomsP, P):
        """a fresh session takes over a chain state: atoms dicts (pos, mass in vector order) and the factor matrices [rows][K]"""
        u64p = C.POINTER(C.c_uint64)
        pa, ma = np.ascontiguousarray(atomsA["pos"], np.uint64), np.ascontiguousarray(atomsA["mass"], np.float32)
        pp, mp = np.ascontiguousarray(atomsP["pos"], np.uint64), np.ascontiguousarray(atomsP["mass"], np.float32)
        a, b = np.ascontiguousarray(A, np.float32), np.ascontiguousarray(P, np.float32)
        if self.L.go_import_state(self.h, pa.ctypes.data_as(u64p), _fp(ma), pa.size, _fp(a), pp.ctypes.data_as(u64p), _fp(mp), pp.size, _fp(b)):
            raise RuntimeError("go_import_state needs a fresh session")

    def debug_alpha(self, which, mode, r1, c1, r2=0, c2=0, ch=0.0):
        out = np.zeros(2, dtype=np.float32)
        self.L.go_debug_alpha(self.h, which.encode(), mode, r1, c1, r2, c2, ch, _fp(out))
        return float(out[0]), float(out[1])

    def rows(self, which):
        """HybridMatrix row copy (sparse model); the matrix itself for the dense model"""
        m, n, k = self.dims(which)
        out = np.zeros((m, k), dtype=np.float32)
        self.L.go_get_rows(self.h, which.encode(), _fp(out))
        return out

    def ap(self, which):
        m, n, k = self.dims(which)
        out = np.zeros((m, n), dtype=np.float32)
        self.L.go_get_ap(self.h, which.encode(), _fp(out))
        return out

    def atoms(self, which):
        n = self.natoms(which)
        pos = np.zeros(n, dtype=np.uint64)
        mass = np.zeros(n, dtype=np.float32)
        left = np.zeros(n, dtype=np.uint32)
        right = np.zeros(n, dtype=np.uint32)
        self.L.go_get_atoms(self.h, which.encode(), pos.ctypes.data_as(C.POINTER(C.c_uint64)), _fp(mass),
                            left.ctypes.data_as(C.POINTER(C.c_uint32)), right.ctypes.data_as(C.POINTER(C.c_uint32)))
        return {"pos": pos, "mass": mass, "left": left, "right": right}

    def lam(self, which):
        return self.L.go_lambda(self.h, which.encode())

    def max_gibbs_mass(self, which):
        return self.L.go_max_gibbs_mass(self.h, which.encode())

    def avg_queue(self, which):
        return self.L.go_avg_queue(self.h, which.encode())

    def finish(self):
        r = GoResult()
        self.L.go_finish(self.h, C.byref(r))
        out = _result_to_dict(r)
        self.L.go_result_free(C.byref(r))
        return out


def luts():
    e = np.zeros(3001, np.float32)
    ei = np.zeros(5001, np.float32)
    qg = np.zeros(5001, np.float32)
    lib().go_build_luts(_fp(e), _fp(ei), _fp(qg))
    return e, ei, qg


def read_mtx(path):
    """MatrixMarket coordinate reader with the reference's text -> fp32 rule
    (file_parser/MtxParser.cpp:50-60, MatrixElement.cpp:15-23: one rounding, text -> float)."""
    L = lib()
    with open(path) as f:
        line = f.readline()
        while "%" in line:
            line = f.readline()
        nr, nc = [int(x) for x in line.split()[:2]]
        out = np.zeros((nr, nc), dtype=np.float32)
        for ln in f:
            t = ln.split()
            if len(t) < 3:
                continue
            out[int(t[0]) - 1, int(t[1]) - 1] = L.go_strtof(t[2].encode())
    return out
