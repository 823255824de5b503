"""Matrix readers for the reference's four input formats (src/file_parser/): .mtx (MtxParser.cpp), .csv / .tsv and
.gct (CharacterDelimitedParser.cpp).  Text becomes fp32 by the reference's rule (MatrixElement.cpp:10-47): a plain
decimal token is converted once, correctly rounded; a token in scientific notation is base * powf(10, exponent) with
base and exponent both read as fp32."""
import numpy as np

_TRIM = " \r\n\""
_PLAIN = set("0123456789.-")


def parse_value(tok):
    """MatrixElement.cpp:15-47"""
    if tok and set(tok) <= _PLAIN:
        return np.float32(tok)
    pos = tok.find("e")
    if pos < 0:
        raise ValueError("Invalid entry found in input data: %s" % tok)
    base, exp = tok[:pos], tok[pos + 1:]
    if not (base and set(base) <= _PLAIN and exp and set(exp) <= _PLAIN):
        raise ValueError("Invalid entry found in input data: %s" % tok)
    return np.float32(np.float32(base) * np.power(np.float32(10.0), np.float32(exp), dtype=np.float32))


def _tokens(line, delim):
    return [t.strip(_TRIM) for t in line.rstrip("\n").split(delim)]


def read_matrix(path, return_names=False):
    """The data matrix as the reference reads it (rows x columns of the file), fp32.  With return_names also the
    row / column names the file carries (empty lists when it has none)."""
    p = path.lower()
    rown, coln = [], []
    if p.endswith(".mtx"):                                          # MtxParser.cpp:8-62: '%' comment lines, then "nrow ncol [nnz]"
        with open(path) as f:
            line = f.readline()
            while "%" in line:
                line = f.readline()
            nr, nc = [int(x) for x in line.split()[:2]]
            out = np.zeros((nr, nc), dtype=np.float32)
            for ln in f:
                t = ln.split()
                if len(t) >= 3:
                    out[int(t[0]) - 1, int(t[1]) - 1] = parse_value(t[2])
    elif p.endswith(".gct"):                                        # CharacterDelimitedParser.cpp:64-74, 105-147: version line,
        with open(path) as f:                                       # dimensions, column names; two leading columns per row
            f.readline()
            nr, nc = [int(x) for x in f.readline().split()[:2]]
            coln = _tokens(f.readline(), "\t")[2:]
            rows = [_tokens(ln, "\t") for ln in f if ln.strip()]
        rown = [r[0] for r in rows]
        out = np.array([[parse_value(x) for x in r[2:]] for r in rows], dtype=np.float32).reshape(len(rows), -1)
        if out.shape != (nr, nc):
            raise ValueError("Invalid character delimited file")
    elif p.endswith(".csv") or p.endswith(".tsv"):                  # :75-101: the first line holds the column names; row names are
        delim = "," if p.endswith(".csv") else "\t"                 # present iff its first field is empty
        with open(path) as f:
            head = _tokens(f.readline(), delim)
            rows = [_tokens(ln, delim) for ln in f if ln.strip()]
        has_row_names = head[0] == ""
        coln = head[1:] if has_row_names else head
        if has_row_names:
            rown = [r[0] for r in rows]
            rows = [r[1:] for r in rows]
        out = np.array([[parse_value(x) for x in r] for r in rows], dtype=np.float32).reshape(len(rows), -1)
    else:
        raise ValueError("unsupported file extension (FileParser.cpp:76-84: .csv, .tsv, .mtx, .gct): " + path)
    return (out, rown, coln) if return_names else out
