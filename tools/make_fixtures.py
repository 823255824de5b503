"""Build the committed data fixtures under tests/golden/ from the reference's own data files.

Run in the build container only (it reads /root/reference); the outputs are data, not source:
  tests/golden/GIST.{mtx,csv,tsv,gct} -- verbatim copies of inst/extdata/GIST.* (the reference's test data files, 1363x9,
                                    one matrix in the four input formats of src/file_parser/)
  tests/golden/modsimdata.csv    -- data/modsimdata.rda (gzip + XDR data.frame, 25x20) printed with
                                    %.17g so that text -> fp32 equals the double -> fp32 cast the
                                    reference applies to R matrices (Cogaps.cpp:21-32)
"""
import gzip
import os
import shutil
import struct

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


class XDR:
    def __init__(self, b):
        self.b, self.i = b, 0

    def i32(self):
        v = struct.unpack(">i", self.b[self.i:self.i + 4])[0]
        self.i += 4
        return v

    def f64(self, n):
        v = np.frombuffer(self.b[self.i:self.i + 8 * n], dtype=">f8").astype(np.float64)
        self.i += 8 * n
        return v

    def raw(self, n):
        v = self.b[self.i:self.i + n]
        self.i += n
        return v


def read_item(x, reals, strings):
    flags = x.i32()
    t = flags & 0xFF
    has_attr = bool(flags & 0x200)
    has_tag = bool(flags & 0x400)
    if t == 0xFE or t == 0xFB or t == 0xFD:   # NILVALUE / MISSINGARG / GLOBALENV
        return None
    if t == 0xFF:                              # REFSXP
        return ("ref", flags >> 8)
    if t == 1:                                 # SYMSXP
        return ("sym", read_item(x, reals, strings))
    if t == 2:                                 # LISTSXP (pairlist)
        if has_attr:
            read_item(x, reals, strings)
        if has_tag:
            read_item(x, reals, strings)
        car = read_item(x, reals, strings)
        cdr = read_item(x, reals, strings)
        return ("pair", car, cdr)
    if t == 9:                                 # CHARSXP
        n = x.i32()
        s = None if n == -1 else x.raw(n).decode()
        strings.append(s)
        return s
    if t == 13:                                # INTSXP
        n = x.i32()
        v = [x.i32() for _ in range(n)]
        if has_attr:
            read_item(x, reals, strings)
        return v
    if t == 14:                                # REALSXP
        n = x.i32()
        v = x.f64(n)
        reals.append(v)
        if has_attr:
            read_item(x, reals, strings)
        return v
    if t == 16 or t == 19:                     # STRSXP / VECSXP
        n = x.i32()
        v = [read_item(x, reals, strings) for _ in range(n)]
        if has_attr:
            read_item(x, reals, strings)
        return v
    raise ValueError("unsupported SEXP type %d at %d" % (t, x.i))


def read_modsim():
    raw = gzip.open(os.path.join(REF, "data", "modsimdata.rda"), "rb").read()
    assert raw[:5] == b"RDX3\n" and raw[5:7] == b"X\n"
    x = XDR(raw[7:])
    x.i32(); x.i32(); x.i32()                  # format version, writer, min reader
    n = x.i32(); x.raw(n)                      # native encoding
    reals, strings = [], []
    read_item(x, reals, strings)
    cols = [r for r in reals if len(r) == 25]
    assert len(cols) == 20, len(cols)
    return np.stack(cols, axis=1)              # 25 x 20, data.frame columns -> matrix columns


def main():
    os.makedirs(OUT, exist_ok=True)
    for ext in ("mtx", "csv", "tsv", "gct"):
        shutil.copyfile(os.path.join(REF, "inst", "extdata", "GIST." + ext), os.path.join(OUT, "GIST." + ext))
        os.chmod(os.path.join(OUT, "GIST." + ext), 0o644)
    m = read_modsim()
    with open(os.path.join(OUT, "modsimdata.csv"), "w") as f:
        for row in m:
            f.write(",".join("%.17g" % v for v in row) + "\n")
    print("modsimdata", m.shape, float(m.min()), float(m.max()))


if __name__ == "__main__":
    main()
