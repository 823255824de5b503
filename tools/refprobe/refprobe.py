"""Python side of tools/refprobe (TEST TOOLING, build container only): build the probe, write inputs in the reference's four file
formats, run it, parse what it prints.  Nothing under cogaps_amd/ imports this."""
import os
import shutil
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_SRC = os.environ.get("REFERENCE_SRC", "/root/reference/src")


def available():
    return os.path.isdir(REFERENCE_SRC) and shutil.which("g++") is not None


def build(out=None):
    """-> path of the probe binary (compiled into /tmp, never into the repository)"""
    out = out or os.environ.get("REFPROBE_OUT", "/tmp/refprobe")
    r = subprocess.run(["bash", os.path.join(HERE, "build.sh"), out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    if r.returncode != 0:
        raise RuntimeError("refprobe build failed:\n" + r.stderr[-2000:])
    return r.stdout.strip().splitlines()[-1]


def _txt(v):
    """shortest fixed-notation decimal that reads back as the same float32 (the reference parses token -> float directly and treats
    exponent notation with its own fp32 pow, file_parser/MatrixElement.cpp:15-47: keep to plain decimals)"""
    return np.format_float_positional(np.float32(v), unique=True, trim="-")


def write_matrix(path, m):
    """m: float32 [rows][cols]; format from the extension (.mtx / .csv / .tsv / .gct), as file_parser/FileParser.cpp picks it"""
    m = np.asarray(m, dtype=np.float32)
    ext = os.path.splitext(path)[1]
    nr, nc = m.shape
    with open(path, "w") as f:
        if ext == ".mtx":
            nz = [(i, j) for j in range(nc) for i in range(nr) if m[i, j] != 0]
            f.write("%%MatrixMarket matrix coordinate real general\n%d %d %d\n" % (nr, nc, len(nz)))
            for i, j in nz:
                f.write("%d %d %s\n" % (i + 1, j + 1, _txt(m[i, j])))
        elif ext in (".csv", ".tsv"):
            d = "," if ext == ".csv" else "\t"
            f.write(d.join(['""'] + ['"c%d"' % j for j in range(nc)]) + "\n")
            for i in range(nr):
                f.write(d.join(['"r%d"' % i] + [_txt(v) for v in m[i]]) + "\n")
        elif ext == ".gct":
            f.write("#1.2\n%d\t%d\n" % (nr, nc))
            f.write("\t".join(['"NAME"', '"Description"'] + ['"c%d"' % j for j in range(nc)]) + "\n")
            for i in range(nr):
                f.write("\t".join(['"r%d"' % i, '"BLANK"'] + [_txt(v) for v in m[i]]) + "\n")
        else:
            raise ValueError(ext)


def run(binary, data, unc=None, subset=None, subsetDim=0, fixed="N", fixedFile=None, timeout=None, **kw):
    """kw: nPatterns nIterations seed threads outFreq sparse transpose alphaA alphaP maxGibbsA maxGibbsP pump snapshots snapshotPhase.
    -> dict: atomsA, atomsP (uint32), chisq (float32), totalUpdates, meanChiSq, qA, qP (float32), rows[(name, r)] -> float32 vector,
    hashes[name] -> (fnv, count), snapE / snapS lists of (hashA, hashP), samplerSeconds, wallSeconds"""
    args = [binary, "data=" + data]
    if unc:
        args.append("unc=" + unc)
    if subsetDim:
        args += ["subsetDim=%d" % subsetDim, "subset=" + subset]
    if fixed != "N":
        args += ["fixed=" + fixed, "fixedFile=" + fixedFile]
    args += ["%s=%s" % (k, int(v) if isinstance(v, (bool, np.bool_)) else v) for k, v in kw.items()]
    env = dict(os.environ)
    env["OMP_NUM_THREADS"] = str(kw.get("threads", 1))
    p = subprocess.run(args, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)
    if p.returncode != 0:
        raise RuntimeError("refprobe failed (%d): %s\n%s" % (p.returncode, " ".join(args), p.stderr[-2000:]))
    out = {"rows": {}, "hashes": {}, "snapE": [], "snapS": []}
    for ln in p.stdout.splitlines():
        t = ln.split()
        if not t:
            continue
        k = t[0]
        if k == "dims":
            out["dims"] = tuple(int(x) for x in t[1:4])
        elif k in ("atomsA", "atomsP"):
            out[k] = np.array([int(x) for x in t[1:]], dtype=np.uint32)
        elif k == "chisq":
            out[k] = np.array([float(x) for x in t[1:]], dtype=np.float32)
        elif k == "totalUpdates":
            out[k] = int(t[1])
        elif k in ("meanChiSq", "qA", "qP"):
            out[k] = np.float32(float(t[1]))
        elif k == "row":
            out["rows"][(t[1], int(t[2]))] = np.array([float(x) for x in t[3:]], dtype=np.float32)
        elif k == "hash":
            out["hashes"][t[1]] = (int(t[2]), int(t[3]))
        elif k in ("snapE", "snapS"):
            out[k].append((int(t[2]), int(t[3])))
        elif k in ("samplerSeconds", "wallSeconds"):
            out[k] = float(t[1])
    return out


def fnv_matrix(m):
    """the driver's hash: FNV-1a 64 over the float32 bit patterns, row-major, little-endian bytes"""
    h = 1469598103934665603
    for b in np.ascontiguousarray(m, dtype="<f4").tobytes():
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h
