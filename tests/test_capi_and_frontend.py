"""The C-ABI library loads without a GPU and exports every symbol include/cogaps_hip.h declares; the
front-end mirrors the reference's parameter names, defaults and validation (R/class-CogapsParams.R,
R/HelperFunctions.R:194-249).  No compute is launched here."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT


def test_library_builds_loads_and_exports_header_symbols():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "cogaps_amd", "csrc")])
    from cogaps_amd import _capi
    lib = _capi.load()
    header = open(os.path.join(ROOT, "include", "cogaps_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)        # declarations only, not the comments
    declared = set(re.findall(r"\b(cogaps_[a-z_0-9]+)\s*\(", header))
    declared -= {"cogaps_session"}
    assert declared, "no declarations found"
    for name in sorted(declared):
        assert hasattr(lib, name), "libcogaps_hip.so does not export " + name
    assert set(_capi.EXPORTS) <= declared
    assert b"gfx950" in lib.cogaps_build_report()
    assert lib.cogaps_checkpoints_enabled() == 0 and lib.cogaps_compiled_with_openmp() == 0
    assert [lib.cogaps_reduction_width(n) for n in (9, 256, 257, 2000, 2049, 20000, 70000)] == [64, 64, 128, 512, 1024, 8192, 16384]


def test_struct_layouts_match_header():
    from cogaps_amd import _capi
    lib = _capi.load()
    p = _capi.CogapsParamsC()
    lib.cogaps_default_params(ctypes.byref(p))
    # GapsParameters.h:79-111 defaults
    assert (p.nPatterns, p.nIterations, p.outputFrequency, p.maxThreads) == (3, 1000, 500, 1)
    assert abs(p.alphaA - 0.01) < 1e-9 and abs(p.maxGibbsMassP - 100.0) < 1e-9
    assert p.whichMatrixFixed == b"N" and p.asynchronousUpdates == 1 and p.device == -1


def test_no_cpu_fallback_when_library_missing(monkeypatch, tmp_path):
    from cogaps_amd import _capi
    monkeypatch.setattr(_capi, "_lib", None)
    monkeypatch.setattr(_capi, "LIB_PATH", str(tmp_path / "missing.so"))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load()


def test_product_never_touches_the_oracle():
    for d, _, files in os.walk(os.path.join(ROOT, "cogaps_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".cpp", ".hip")):
                txt = open(os.path.join(d, f)).read()
                assert "pyoracle" not in txt and "gaps_oracle" not in txt and "liboracle" not in txt, f


def test_params_defaults_and_validation():
    from cogaps_amd import CogapsParams
    p = CogapsParams(nPatterns=5, seed=11)
    # R/class-CogapsParams.R:99-123
    assert (p.nIterations, p.alphaA, p.maxGibbsMassA, p.nSets, p.cut, p.minNS, p.maxNS) == (50000, 0.01, 100.0, 4, 5, 2, 6)
    assert p.whichMatrixFixed == "N" and p.distributed is None and p.sparseOptimization is False
    p.setParam("alpha", 0.05)
    assert p.alphaA == p.alphaP == 0.05
    with pytest.raises(ValueError):
        p.setParam("nSets", 3)
    with pytest.raises(ValueError):
        p.setParam("nPatterns", 0)
    with pytest.raises(ValueError):
        CogapsParams(distributed="everywhere")
    with pytest.raises(ValueError):
        p.setFixedPatterns(np.ones((3, 5)), "Q")
    p.setParam("distributed", "genome-wide")
    p.setDistributedParams(nSets=8)
    assert (p.nSets, p.minNS, p.maxNS) == (8, 4, 12)


def test_check_inputs_rules():
    from cogaps_amd import CogapsParams
    from cogaps_amd.api import check_inputs
    p = CogapsParams(nPatterns=3, seed=1)
    ok = np.ones((10, 8), dtype=np.float32)
    check_inputs(ok, None, p)
    with pytest.raises(ValueError, match="negative"):
        check_inputs(-ok, None, p)
    with pytest.raises(ValueError, match="nPatterns must be less"):
        check_inputs(np.ones((3, 8), np.float32), None, p)
    bad = ok.copy(); bad[0, 0] = np.nan
    with pytest.raises(ValueError, match="NA"):
        check_inputs(bad, None, p)
    with pytest.raises(ValueError, match="checkpoints"):
        check_inputs(ok, None, p, checkpointInFile="x")


def test_readers(tmp_path, gist):
    from cogaps_amd.io import read_matrix
    from conftest import GOLDEN
    m = read_matrix(os.path.join(GOLDEN, "GIST.mtx"))
    assert m.shape == (1363, 9) and np.array_equal(m, gist)
    p = tmp_path / "x.csv"
    p.write_text(",s1,s2\ng1,1.5,2\ng2,3,4e-1\n")
    assert np.array_equal(read_matrix(str(p)), np.array([[1.5, 2], [3, 0.4]], dtype=np.float32))


def test_input_file_formats(gist):
    """the reference's GIST data set in its four input formats gives one fp32 matrix (tests/testthat/test_reading_input_files.R
    checks the same through CoGAPS()); scientific notation follows MatrixElement.cpp:24-46"""
    import os
    from conftest import GOLDEN
    from cogaps_amd.io import read_matrix, parse_value
    for ext in ("mtx", "csv", "tsv", "gct"):
        m, rown, coln = read_matrix(os.path.join(GOLDEN, "GIST." + ext), return_names=True)
        assert m.dtype == np.float32 and np.array_equal(m, gist), ext
        if ext != "mtx":
            assert len(rown) == 1363 and rown[0] == "Hs.101174" and len(coln) == 9 and coln[0] == "IM00"
    assert parse_value("1.5e3") == np.float32(1500.0) and parse_value("-2e-2") == np.float32(np.float32(-2.0) * np.power(np.float32(10.0), np.float32(-2.0)))
    assert parse_value("0.1") == np.float32(0.1)
    with pytest.raises(ValueError):
        parse_value("abc")


def test_native_file_reader_matches_fixtures_and_python_reader(tmp_path, gist):
    """csrc/file_reader.h (behind cogaps_read_matrix_file / cogaps_file_info, host only) against the four GIST fixtures, the
    Python reader and hand-made edge cases of the reference's parsers (CharacterDelimitedParser.cpp, MtxParser.cpp,
    MatrixElement.cpp:10-47)"""
    import ctypes
    from conftest import GOLDEN
    from cogaps_amd import _capi
    from cogaps_amd.io import read_matrix
    L = _capi.bind(ctypes.CDLL(_capi.LIB_PATH))
    for ext in ("mtx", "csv", "tsv", "gct"):
        path = os.path.join(GOLDEN, "GIST." + ext)
        m = _capi.read_matrix_file(path, lib=L)
        assert m.dtype == np.float32 and np.array_equal(m, gist), ext
        nr, nc, rown, coln = _capi.file_info(path, lib=L)
        assert (nr, nc) == (1363, 9)
        if ext != "mtx":
            assert len(rown) == 1363 and rown[0] == "Hs.101174" and rown[-1] == read_matrix(path, return_names=True)[1][-1] and coln[0] == "IM00" and len(coln) == 9
    cases = {
        "a.csv": ',"s1",s2\r\n"g1", 1.5 ,2\r\ng2,3,4e-1\r\n\r\n',                   # quotes, blanks, CRLF, trailing blank line, row names
        "b.csv": "s1,s2,s3\n1,2,3\n-4.25,5e2,6.\n",                                      # no row names
        "c.tsv": "\ts1\ts2\ng1\t1e-3\t-2.5e1\ng2\t0\t.5\n",
        "d.mtx": "%%MatrixMarket matrix coordinate real general\n% comment\n3 2 3\n1 1 1.5\n3 2 2e-2\n2 1 -7\n",
        "e.gct": "#1.2\n2\t3\nName\tDescription\tx\ty\tz\nr1\td1\t1\t2\t3\nr2\td2\t4.5\t5e0\t-6\n",
    }
    for name, text in cases.items():
        f = tmp_path / name
        f.write_bytes(text.encode())
        a, b = _capi.read_matrix_file(str(f), lib=L), read_matrix(str(f))
        assert a.shape == b.shape and np.array_equal(a, b), name
    assert _capi.read_matrix_file(str(tmp_path / "c.tsv"), lib=L)[0, 0] == np.float32(np.float32(1.0) * np.power(np.float32(10.0), np.float32(-3.0)))
    assert _capi.file_info(str(tmp_path / "b.csv"), lib=L) == (2, 3, [], ["s1", "s2", "s3"])
    assert _capi.file_info(str(tmp_path / "e.gct"), lib=L) == (2, 3, ["r1", "r2"], ["x", "y", "z"])
    for name, text in {"bad1.csv": ",a\nr,abc\n", "bad2.csv": ",a,b\nr,1\n", "bad3.gct": "#1.2\n3\t1\nN\tD\tx\nr\td\t1\n", "bad.xyz": "1"}.items():
        f = tmp_path / name
        f.write_text(text)
        with pytest.raises(RuntimeError):
            _capi.read_matrix_file(str(f), lib=L)
    with pytest.raises(RuntimeError):
        _capi.read_matrix_file(str(tmp_path / "missing.csv"), lib=L)


def test_compute_calls_fail_loudly_without_a_gpu():
    """no device, no result: the library reports the HIP error through the C ABI (non-zero code + message), nothing falls
    back to the host"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: this is the no-device behaviour")
    from cogaps_amd import _capi, CoGAPS
    with pytest.raises(RuntimeError, match="device"):
        _capi.run(np.ones((10, 8), np.float32), nPatterns=2, nIterations=5)
    with pytest.raises(RuntimeError, match="device"):
        CoGAPS(np.ones((10, 8), np.float32) * 2, nPatterns=2, nIterations=5, messages=False)


def test_rcpp_glue_meets_a_compiler():
    """bindings/r/CogapsHip.cpp -- the file a maintainer of the R package copies (reference src/Cogaps.cpp:148-254, src/RcppExports.cpp:11-117)
    -- parses and type-checks against include/cogaps_hip.h and a mock of the Rcpp declarations it uses (tests/c/mock_rcpp: test
    infrastructure, pins nothing about R).  It registers the six .Call entry points of R/RcppExports.R with the reference's arities, and a
    typo in it fails this test."""
    import re, subprocess, shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    src = os.path.join(ROOT, "bindings", "r", "CogapsHip.cpp")
    cmd = ["g++", "-std=c++11", "-fsyntax-only", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "tests", "c", "mock_rcpp"), "-I" + os.path.join(ROOT, "include")]
    out = subprocess.run(cmd + [src], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr
    text = open(src).read()
    table = dict(re.findall(r'\{"(_CoGAPS_\w+)",\s*\(DL_FUNC\) &\w+,\s*(\d+)\}', text))
    assert table == {"_CoGAPS_cogaps_cpp": "3", "_CoGAPS_cogaps_from_file_cpp": "3", "_CoGAPS_getBuildReport_cpp": "0", "_CoGAPS_checkpointsEnabled_cpp": "0",
                     "_CoGAPS_compiledWithOpenMPSupport_cpp": "0", "_CoGAPS_getFileInfo_cpp": "1"}
    # every C-ABI symbol the glue calls is declared by the header and exported by the library's symbol list
    called = set(re.findall(r"\b(cogaps_[a-z_]+)\s*\(", text)) - {"cogaps_params", "cogaps_result", "cogaps_cpp", "cogaps_from_file_cpp"}
    header = open(os.path.join(ROOT, "include", "cogaps_hip.h")).read()
    for fn in called:
        assert re.search(r"\b" + fn + r"\s*\(", header), fn
    # the check has teeth: a misspelt call does not pass
    bad = os.path.join(ROOT, "tests", "c", "_typo_CogapsHip.cpp")
    with open(bad, "w") as f:
        f.write(text.replace("cogaps_result_free(&r);", "cogaps_result_fre(&r);", 1))
    try:
        assert subprocess.run(cmd + [bad], capture_output=True, text=True).returncode != 0
    finally:
        os.remove(bad)


def test_plain_c_client_builds_and_fails_loudly_without_a_gpu():
    """tests/c/rcpp_shim_test.c: getGapsParameters (src/Cogaps.cpp:64-139) restated in C99 against include/cogaps_hip.h.  It must
    compile as C, link against the product library and -- on a box without a GPU -- end with the library's error text and a
    non-zero status, never a crash or a silent CPU path (GAPS_ERROR -> Rcpp::stop in the reference, utils/GapsAssert.h:19-25)"""
    import subprocess
    cdir = os.path.join(ROOT, "tests", "c")
    subprocess.check_call(["make", "-s", "-C", cdir])
    exe = os.path.join(cdir, "rcpp_shim_test.bin")
    out = subprocess.run([exe, os.path.join(GOLDEN, "GIST.mtx"), "checkpointInFile=x"], capture_output=True, text=True)
    assert out.returncode == 1 and "checkpoints are disabled" in out.stderr
    out = subprocess.run([exe, os.path.join(GOLDEN, "GIST.mtx"), "bogusKey=1"], capture_output=True, text=True)
    assert out.returncode == 2
    # getFileInfo_cpp (Cogaps.cpp:243-254) through cogaps_file_info, host only: the names of GIST.gct / .csv round-trip through the
    # '\n'-joined buffers; a .mtx carries none
    from cogaps_amd.io import read_matrix
    for ext in ("gct", "csv", "mtx"):
        out = subprocess.run([exe, os.path.join(GOLDEN, "GIST." + ext), "entry=info"], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        lines = out.stdout.splitlines()
        assert lines[0] == "dimensions 1363 9"
        _, rown, coln = read_matrix(os.path.join(GOLDEN, "GIST." + ext), return_names=True)
        assert [ln.split(" ", 2)[2] for ln in lines if ln.startswith("rowName ")] == rown
        assert [ln.split(" ", 2)[2] for ln in lines if ln.startswith("colName ")] == coln
        assert "rowNames %d" % len(rown) in lines and "colNames %d" % len(coln) in lines
    # error returns of the file entry points: a missing file, an unknown extension, a malformed table -- status 1 and the message
    bad = os.path.join(cdir, "_bad_table.csv")
    with open(bad, "w") as f:
        f.write(",a,b\nr1,1\n")
    try:
        for path, what in (("/nonexistent/x.csv", "cannot open"), (os.path.join(GOLDEN, "c3_k50_s42_i100_lane.npz"), "unsupported file extension"), (bad, "Invalid character delimited file")):
            for entry in ("file", "info"):
                out = subprocess.run([exe, path, "entry=" + entry, "nPatterns=3", "nIterations=5", "messages=0"], capture_output=True, text=True)
                assert out.returncode == 1 and out.stderr.startswith("CoGAPS terminated: ") and what in out.stderr and out.stdout == "", (path, entry, out.stderr)
    finally:
        os.remove(bad)
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    if not gpu:
        out = subprocess.run([exe, os.path.join(GOLDEN, "GIST.mtx"), "nPatterns=3", "nIterations=5", "messages=0"], capture_output=True, text=True)
        assert out.returncode == 1 and out.stderr.startswith("CoGAPS terminated: ") and out.stdout == ""


def test_native_subset_reader_reads_only_the_named_rows_or_columns(tmp_path, gist):
    """cogaps_read_matrix_file_subset: what a worker of a distributed run reads of a file (Matrix(path, genesInCols, subsetGenes,
    indices), data_structures/Matrix.cpp:70-134): the indices are sorted, an element lands at its index's lower_bound position, a
    duplicated index fills its first position only; all four formats, rows and columns; file_info parses no values"""
    import ctypes
    from conftest import GOLDEN
    from cogaps_amd import _capi
    L = _capi.bind(ctypes.CDLL(_capi.LIB_PATH))
    rows = np.array([700, 3, 1363, 41, 42, 1], dtype=np.uint32)
    cols = np.array([9, 2, 5], dtype=np.uint32)
    for ext in ("mtx", "csv", "tsv", "gct"):
        path = os.path.join(GOLDEN, "GIST." + ext)
        a = _capi.read_matrix_file(path, lib=L, rows=rows)
        assert a.shape == (6, 9) and np.array_equal(a, gist[np.sort(rows) - 1]), ext
        b = _capi.read_matrix_file(path, lib=L, cols=cols)
        assert b.shape == (1363, 3) and np.array_equal(b, gist[:, np.sort(cols) - 1]), ext
        d = _capi.read_matrix_file(path, lib=L, rows=np.array([5, 5, 2], dtype=np.uint32))       # sorted: 2, 5, 5 -> the second 5 stays empty
        assert np.array_equal(d[0], gist[1]) and np.array_equal(d[1], gist[4]) and not d[2].any(), ext
        for bad in (np.array([0, 1], np.uint32), np.array([1364], np.uint32)):
            with pytest.raises(RuntimeError):
                _capi.read_matrix_file(path, lib=L, rows=bad)
        with pytest.raises(RuntimeError):
            _capi.read_matrix_file(path, lib=L, cols=np.array([10], np.uint32))
    with pytest.raises(ValueError):
        _capi.read_matrix_file(os.path.join(GOLDEN, "GIST.mtx"), lib=L, rows=rows, cols=cols)
    # dimensions and names without values: a file whose values do not parse still answers file_info (getFileInfo_cpp needs no matrix)
    f = tmp_path / "names_only.csv"
    f.write_text(",a,b\nr1,oops,1\nr2,2,3\n")
    assert _capi.file_info(str(f), lib=L) == (2, 2, ["r1", "r2"], ["a", "b"])
    with pytest.raises(RuntimeError):
        _capi.read_matrix_file(str(f), lib=L)


def test_bench_refuses_to_report_fewer_gpus_than_asked_for():
    """`python bench.py --gpus 2` without a launcher starts its own two ranks (torch.distributed.run, 127.0.0.1); where they cannot
    come up -- here: no GPU at all -- the command fails and prints NO bench line: never `n_gpus: 1` for `--gpus 2`"""
    import subprocess
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the launch path is covered by the gpu tests")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu"], cwd=ROOT, env=env,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert "2-rank launch failed" in out.stderr and "needs an MI355X" in out.stderr
    # a launcher-provided world that contradicts --gpus is refused as well
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--no-cpu"], cwd=ROOT, env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0"),
                         capture_output=True, text=True, timeout=600)
    assert out.returncode != 0 and "WORLD_SIZE=2" in out.stderr and not [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
