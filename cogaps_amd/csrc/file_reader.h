// file_reader.h -- host-only readers for the four input formats of the reference's file entry point
// (gaps::run(const std::string &data, ...), GapsRunner.h:24-29; Rcpp cogaps_from_file_cpp, Cogaps.cpp:217-227;
// getFileInfo, Cogaps.cpp:229-246): Matrix Market coordinate files (.mtx, file_parser/MtxParser.cpp), comma / tab
// separated tables with a header line (.csv / .tsv) and GenePattern .gct (file_parser/CharacterDelimitedParser.cpp).
// Text becomes fp32 by the reference's rule (file_parser/MatrixElement.cpp:10-47): a token made of "0123456789.-" is
// converted once (stream extraction = strtof, correctly rounded); anything else must be <base>e<exp> with both parts of
// that alphabet and is evaluated as float(base) * powf(10.f, float(exp)) -- NOT the correctly rounded value.
// The file's text is held in memory and scanned once; the result is the dense row-major matrix -- of the whole file, or, with a
// Subset, of the named rows / columns only: the reference's workers read their subset of a file that way
// (Matrix(path, genesInCols, subsetGenes, indices), data_structures/Matrix.cpp:70-134: the indices are SORTED first, an element is
// kept when its row / column index is found by lower_bound and lands at that position -- a duplicated index fills its first
// position only), so that no worker of a distributed run ever holds the whole matrix.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>

namespace cgio {

struct Table {
    uint32_t nrow = 0, ncol = 0;
    std::vector<float> v;                     // row-major [nrow][ncol]
    std::vector<std::string> rowNames, colNames;
    uint32_t fileRows = 0, fileCols = 0;      // dimensions of the file itself (= nrow, ncol without a subset)
};

// which rows (byRows) or columns of the file to keep: 1-based indices, sorted by the constructor (Matrix.cpp:113)
struct Subset {
    bool active = false, byRows = true;
    std::vector<uint32_t> idx;
    Subset() {}
    Subset(bool rows, const uint32_t *p, size_t n) : active(true), byRows(rows), idx(p, p + n) { std::sort(idx.begin(), idx.end()); }
    // position of 1-based index i in the subset, or -1 (lower_bound: the first of equal entries)
    long find(uint32_t i) const { auto it = std::lower_bound(idx.begin(), idx.end(), i); return (it != idx.end() && *it == i) ? (long)(it - idx.begin()) : -1l; }
};
// values = false: dimensions and names only (getFileInfo_cpp needs no matrix)
struct ReadOpts { Subset sub; bool values = true; };

inline bool plain_number(const char *b, const char *e)
{
    if (b == e) return false;
    for (const char *p = b; p != e; ++p) { const char c = *p; if (!((c >= '0' && c <= '9') || c == '.' || c == '-')) return false; }
    return true;
}
inline float to_f32(const char *b, const char *e)
{
    char buf[64]; const size_t n = (size_t)(e - b);
    if (n < sizeof(buf)) { for (size_t i = 0; i < n; ++i) buf[i] = b[i]; buf[n] = 0; return strtof(buf, nullptr); }
    return strtof(std::string(b, e).c_str(), nullptr);
}
inline float parse_value(const char *b, const char *e)
{
    if (plain_number(b, e)) return to_f32(b, e);
    const char *x = b; while (x != e && *x != 'e') ++x;
    if (x == e || !plain_number(b, x) || !plain_number(x + 1, e)) throw std::runtime_error("Invalid entry found in input data: " + std::string(b, e));
    return to_f32(b, x) * powf(10.f, to_f32(x + 1, e));
}
inline bool trim_char(char c) { return c == ' ' || c == '\r' || c == '\n' || c == '"'; }
inline void trim(const char *&b, const char *&e) { while (b != e && trim_char(*b)) ++b; while (e != b && trim_char(e[-1])) --e; }

inline std::string slurp(const std::string &path)
{
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    std::string s; char buf[1 << 16]; size_t n;
    while ((n = fread(buf, 1, sizeof(buf), f)) > 0) s.append(buf, n);
    fclose(f);
    return s;
}
inline bool ends_with(const std::string &s, const char *suf) { const std::string t(suf); return s.size() >= t.size() && s.compare(s.size() - t.size(), t.size(), t) == 0; }
// [b, e) of the next line (without the '\n'); false at the end of the text
inline bool next_line(const std::string &s, size_t &pos, const char *&b, const char *&e)
{
    if (pos >= s.size()) return false;
    size_t nl = s.find('\n', pos); if (nl == std::string::npos) nl = s.size();
    b = s.data() + pos; e = s.data() + nl; pos = nl + 1;
    return true;
}
inline bool blank(const char *b, const char *e) { for (; b != e; ++b) if (!(*b == ' ' || *b == '\t' || *b == '\r')) return false; return true; }
template <class F> inline void for_fields(const char *b, const char *e, char delim, F f)
{
    const char *p = b; uint32_t k = 0;
    for (;;) {
        const char *q = p; while (q != e && *q != delim) ++q;
        const char *tb = p, *te = q; trim(tb, te); f(k++, tb, te);
        if (q == e) break;
        p = q + 1;
        if (p == e) break;            // (a trailing delimiter opens no further field: std::getline semantics)
    }
}
inline void two_uints(const char *b, const char *e, uint32_t &a, uint32_t &c)
{
    std::string t(b, e); char *end = nullptr;
    a = (uint32_t)strtoul(t.c_str(), &end, 10); c = (uint32_t)strtoul(end, &end, 10);
}

inline Table read_mtx(const std::string &text, const ReadOpts &o = ReadOpts())                       // MtxParser.cpp:8-62
{
    Table t; size_t pos = 0; const char *b, *e;
    do { if (!next_line(text, pos, b, e)) throw std::runtime_error("Invalid MTX file"); } while (std::string(b, e).find('%') != std::string::npos);
    two_uints(b, e, t.fileRows, t.fileCols);
    t.nrow = t.fileRows; t.ncol = t.fileCols;
    if (o.sub.active) { if (o.sub.byRows) t.nrow = (uint32_t)o.sub.idx.size(); else t.ncol = (uint32_t)o.sub.idx.size(); }
    if (!o.values) return t;
    t.v.assign((size_t)t.nrow * t.ncol, 0.f);
    while (next_line(text, pos, b, e)) {
        if (blank(b, e)) continue;
        // "row col value", whitespace separated, 1-based
        const char *p = b; const char *tok[3][2]; int n = 0;
        while (p != e && n < 3) {
            while (p != e && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
            if (p == e) break;
            tok[n][0] = p; while (p != e && !(*p == ' ' || *p == '\t' || *p == '\r')) ++p; tok[n][1] = p; ++n;
        }
        if (n < 3) continue;
        uint32_t r = (uint32_t)strtoul(std::string(tok[0][0], tok[0][1]).c_str(), nullptr, 10), c = (uint32_t)strtoul(std::string(tok[1][0], tok[1][1]).c_str(), nullptr, 10);
        if (r < 1 || r > t.fileRows || c < 1 || c > t.fileCols) throw std::runtime_error("MTX entry outside the stated dimensions");
        if (o.sub.active) {
            const long k = o.sub.find(o.sub.byRows ? r : c);
            if (k < 0) continue;
            if (o.sub.byRows) r = (uint32_t)k + 1u; else c = (uint32_t)k + 1u;
        }
        t.v[(size_t)(r - 1) * t.ncol + (c - 1)] = parse_value(tok[2][0], tok[2][1]);
    }
    return t;
}

inline Table read_delimited(const std::string &text, char delim, bool gct, const ReadOpts &o = ReadOpts())          // CharacterDelimitedParser.cpp:56-147
{
    Table t; size_t pos = 0; const char *b, *e;
    uint32_t statedRows = 0, statedCols = 0; bool rowNames = false; uint32_t lead = 0;
    std::vector<std::string> allCols;
    if (gct) {
        if (!next_line(text, pos, b, e) || !next_line(text, pos, b, e)) throw std::runtime_error("Invalid character delimited file");
        two_uints(b, e, statedRows, statedCols);
        if (!next_line(text, pos, b, e)) throw std::runtime_error("Invalid character delimited file");
        for_fields(b, e, delim, [&](uint32_t k, const char *fb, const char *fe) { if (k >= 2) allCols.emplace_back(fb, fe); });
        rowNames = true; lead = 2;
    } else {
        if (!next_line(text, pos, b, e)) throw std::runtime_error("Invalid character delimited file");
        // row names are present iff the first header field is empty (:77-84)
        for_fields(b, e, delim, [&](uint32_t k, const char *fb, const char *fe) { if (k == 0) { rowNames = (fb == fe); if (!rowNames) allCols.emplace_back(fb, fe); } else allCols.emplace_back(fb, fe); });
        lead = rowNames ? 1u : 0u;
    }
    t.fileCols = (uint32_t)allCols.size();
    const bool subRows = o.sub.active && o.sub.byRows, subCols = o.sub.active && !o.sub.byRows;
    // column subset: position of every file column in the kept matrix (-1 = dropped)
    std::vector<long> colPos;
    if (subCols) {
        for (uint32_t i : o.sub.idx) if (i < 1 || i > t.fileCols) throw std::runtime_error("subset index outside the file's columns");
        colPos.resize(t.fileCols); for (uint32_t c = 0; c < t.fileCols; ++c) colPos[c] = o.sub.find(c + 1u);
        t.ncol = (uint32_t)o.sub.idx.size(); t.colNames.assign(t.ncol, std::string());
        for (uint32_t c = 0; c < t.fileCols; ++c) if (colPos[c] >= 0) t.colNames[(size_t)colPos[c]] = allCols[c];
    } else { t.ncol = t.fileCols; t.colNames = allCols; }
    if (subRows) { t.nrow = (uint32_t)o.sub.idx.size(); t.rowNames.assign(rowNames ? t.nrow : 0u, std::string()); if (o.values) t.v.assign((size_t)t.nrow * t.ncol, 0.f); }
    std::vector<float> line(t.ncol);
    while (next_line(text, pos, b, e)) {
        if (blank(b, e)) continue;
        const uint32_t fileRow = t.fileRows++;               // 0-based row of the file
        const long rowPos = subRows ? o.sub.find(fileRow + 1u) : (long)fileRow;
        uint32_t got = 0;
        if (subCols && o.values) std::fill(line.begin(), line.end(), 0.f);
        for_fields(b, e, delim, [&](uint32_t k, const char *fb, const char *fe) {
            if (k == 0 && rowNames && rowPos >= 0) { if (subRows) t.rowNames[(size_t)rowPos] = std::string(fb, fe); else t.rowNames.emplace_back(fb, fe); }
            if (k < lead) return;
            const uint32_t c = k - lead; ++got;
            if (!o.values || rowPos < 0 || c >= t.fileCols) return;
            if (subCols) { if (colPos[c] >= 0) line[(size_t)colPos[c]] = parse_value(fb, fe); }
            else if (subRows) t.v[(size_t)rowPos * t.ncol + c] = parse_value(fb, fe);
            else t.v.push_back(parse_value(fb, fe));
        });
        if (got != t.fileCols) throw std::runtime_error("Invalid character delimited file: a row has " + std::to_string(got) + " values, the header " + std::to_string(t.fileCols));
        if (subCols && o.values) t.v.insert(t.v.end(), line.begin(), line.end());
    }
    if (!subRows) t.nrow = t.fileRows;
    else for (uint32_t i : o.sub.idx) if (i < 1 || i > t.fileRows) throw std::runtime_error("subset index outside the file's rows");
    if (gct && (t.fileRows != statedRows || t.fileCols != statedCols)) throw std::runtime_error("Invalid character delimited file");
    return t;
}

inline Table read_matrix_file(const std::string &path, const ReadOpts &o = ReadOpts())               // FileParser.cpp:76-84: dispatch on the extension
{
    std::string low = path; for (char &c : low) if (c >= 'A' && c <= 'Z') c = (char)(c - 'A' + 'a');
    Table t;
    if (ends_with(low, ".mtx")) {
        t = read_mtx(slurp(path), o);
        if (o.sub.active) for (uint32_t i : o.sub.idx) if (i < 1 || i > (o.sub.byRows ? t.fileRows : t.fileCols)) throw std::runtime_error("subset index outside the file's dimensions");
    }
    else if (ends_with(low, ".csv")) t = read_delimited(slurp(path), ',', false, o);
    else if (ends_with(low, ".tsv")) t = read_delimited(slurp(path), '\t', false, o);
    else if (ends_with(low, ".gct")) t = read_delimited(slurp(path), '\t', true, o);
    else throw std::runtime_error("unsupported file extension (.csv, .tsv, .mtx, .gct): " + path);
    return t;
}

} // namespace cgio
