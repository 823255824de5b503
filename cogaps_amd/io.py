"""Matrix readers for the fixture formats (.mtx coordinate, .csv/.tsv with optional header/row names).
Text is parsed straight to fp32 (numpy float32 parser), the reference's rule (file_parser/MatrixElement.cpp:15-23)."""
import numpy as np


def read_matrix(path):
    p = path.lower()
    if p.endswith(".mtx"):
        with open(path) as f:
            line = f.readline()
            while "%" in line:
                line = f.readline()
            nr, nc = [int(x) for x in line.split()[:2]]
            out = np.zeros((nr, nc), dtype=np.float32)
            for ln in f:
                t = ln.split()
                if len(t) >= 3:
                    out[int(t[0]) - 1, int(t[1]) - 1] = np.float32(t[2])
        return out
    delim = "\t" if p.endswith(".tsv") else ","
    rows = [ln.rstrip("\n").split(delim) for ln in open(path) if ln.strip()]

    def isnum(s):
        try:
            float(s)
            return True
        except ValueError:
            return False
    if not all(isnum(x) for x in rows[0]):
        rows = rows[1:]                                          # header line with sample names
    if not isnum(rows[0][0]):
        rows = [r[1:] for r in rows]                             # leading gene-name column
    return np.array([[np.float32(x) for x in r] for r in rows], dtype=np.float32)
