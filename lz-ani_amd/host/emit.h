// emit.h -- result files of the `lz-ani` host binary: ids TSV, ANI TSV, single-txt.
//
// Behaviour follows CLZMatcher::store_results (/root/reference/src/lz_matcher.cpp:280-579) and the
// number formatting of refresh::real_to_pchar (/root/reference/libs/refresh/conversions/lib/
// numeric_conversions.h:228-300, 341-390): shortest round-trip decimal (here std::to_chars, the same
// digits as the reference's dragonbox), rounded half-up to `prec` significant digits, then the
// reference's fixed/exponent layout.  Byte-identical output is the contract (BASELINE config 1).
#pragma once
#include <algorithm>
#include <atomic>
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/lzani.h"
#include "ingest.h"

namespace host {

enum class Comp { query, reference, qidx, ridx, qlen, rlen, tani, gani, ani, qcov, rcov, len_ratio, nt_match, nt_mismatch, num_alns };

inline const std::map<std::string, Comp>& comp_names()
{
    static const std::map<std::string, Comp> m = {
        {"query", Comp::query}, {"reference", Comp::reference}, {"qidx", Comp::qidx}, {"ridx", Comp::ridx},
        {"qlen", Comp::qlen}, {"rlen", Comp::rlen}, {"tani", Comp::tani}, {"gani", Comp::gani}, {"ani", Comp::ani},
        {"qcov", Comp::qcov}, {"rcov", Comp::rcov}, {"len_ratio", Comp::len_ratio}, {"nt_match", Comp::nt_match},
        {"nt_mismatch", Comp::nt_mismatch}, {"num_alns", Comp::num_alns}};
    return m;
}
inline const std::map<std::string, std::string>& comp_metas()          // params.h:65-69
{
    static const std::map<std::string, std::string> m = {
        {"complete", "qidx,ridx,query,reference,tani,gani,ani,qcov,rcov,num_alns,len_ratio,qlen,rlen,nt_match,nt_mismatch"},
        {"standard", "qidx,ridx,query,reference,tani,gani,ani,qcov,num_alns,len_ratio"},
        {"lite", "qidx,ridx,tani,gani,ani,qcov,num_alns,len_ratio"}};
    return m;
}

inline size_t uint_to_chars(uint64_t v, char* out)
{
    auto r = std::to_chars(out, out + 24, v);
    return (size_t)(r.ptr - out);
}

// real_to_pchar(val, out, prec) for finite non-negative doubles (the only ones the emitter produces)
inline size_t real_to_chars(double val, char* out, int prec)
{
    if (val == 0) { *out = '0'; return 1; }
    char* ptr = out;
    if (val < 0) { *ptr++ = '-'; val = -val; }
    char buf[40];
    auto r = std::to_chars(buf, buf + sizeof buf, val, std::chars_format::scientific);   // d[.ddd]e[+-]xx, shortest
    *r.ptr = 0;
    uint64_t sig = 0;
    int nd = 0, frac = 0, exp10 = 0;
    const char* p = buf;
    bool seen_dot = false;
    for (; *p && *p != 'e'; ++p) {
        if (*p == '.') { seen_dot = true; continue; }
        sig = sig * 10 + (uint64_t)(*p - '0');
        ++nd;
        if (seen_dot) ++frac;
    }
    if (*p == 'e') exp10 = atoi(p + 1);
    exp10 -= frac;                                   // val = sig * 10^exp10, sig has nd digits, no trailing zeros
    while (nd > 1 && sig % 10 == 0) { sig /= 10; ++exp10; --nd; }
    static const uint64_t p10[] = {1ull, 10ull, 100ull, 1000ull, 10000ull, 100000ull, 1000000ull, 10000000ull, 100000000ull,
                                   1000000000ull, 10000000000ull, 100000000000ull, 1000000000000ull, 10000000000000ull,
                                   100000000000000ull, 1000000000000000ull, 10000000000000000ull, 100000000000000000ull};
    if (nd > prec) {
        sig += p10[nd - prec] / 2;
        sig /= p10[nd - prec];
        exp10 += nd - prec;
        nd = prec;
        if (sig >= p10[prec]) { sig /= 10; ++exp10; }
    }
    char dig[24];
    size_t dl = uint_to_chars(sig, dig);
    nd = (int)dl;
    if (exp10 == 0) { memcpy(ptr, dig, dl); ptr += dl; }
    else if (exp10 > 0 || -exp10 >= nd + 4) {
        int e = exp10;
        if (nd == 1) *ptr++ = dig[0];
        else { *ptr++ = dig[0]; *ptr++ = '.'; memcpy(ptr, dig + 1, dl - 1); ptr += dl - 1; e += nd - 1; }
        *ptr++ = 'e';
        if (e < 0) { *ptr++ = '-'; e = -e; } else *ptr++ = '+';
        int n = e < 100 ? 2 : e < 1000 ? 3 : 4;
        for (int k = n - 1, v = e; k >= 0; --k, v /= 10) ptr[k] = (char)('0' + v % 10);
        ptr += n;
    } else if (-exp10 < nd) {
        int k = nd + exp10;
        memcpy(ptr, dig, (size_t)k); ptr += k;
        *ptr++ = '.';
        memcpy(ptr, dig + k, dl - (size_t)k); ptr += dl - (size_t)k;
    } else {
        *ptr++ = '0'; *ptr++ = '.';
        for (int z = 0; z < -exp10 - nd; ++z) *ptr++ = '0';
        memcpy(ptr, dig, dl); ptr += dl;
    }
    return (size_t)(ptr - out);
}

// The matching stage's output as it leaves the engine: CSR rows, row r = reference r against its queries in
// ascending id order (the order the reference sorts every results[ref] into, lz_matcher.cpp:253-255).  Dense
// all2all rows store no query list (row r = every id != r).  12 bytes per directed pair, no per-row vectors.
struct PairTable {
    uint32_t n = 0;
    bool dense = true;
    std::vector<uint64_t> row_off;          // n + 1
    std::vector<uint32_t> query_ids;        // filtered runs only
    std::vector<lzani_result> res;          // CSR-aligned

    void init_dense(uint32_t n_)
    {
        n = n_; dense = true; query_ids.clear();
        row_off.resize((size_t)n + 1);
        for (uint32_t r = 0; r <= n; ++r) row_off[r] = (uint64_t)r * (n ? n - 1 : 0);
        res.assign(row_off[n], lzani_result{0, 0, 0});
    }
    // rows[r] = query ids of reference r (any order; sorted here)
    void init_sparse(std::vector<std::vector<uint32_t>>& rows)
    {
        n = (uint32_t)rows.size(); dense = false;
        row_off.assign((size_t)n + 1, 0);
        for (uint32_t r = 0; r < n; ++r) { std::sort(rows[r].begin(), rows[r].end()); row_off[r + 1] = row_off[r] + rows[r].size(); }
        query_ids.resize(row_off[n]);
        for (uint32_t r = 0; r < n; ++r) std::copy(rows[r].begin(), rows[r].end(), query_ids.begin() + (ptrdiff_t)row_off[r]);
        res.assign(row_off[n], lzani_result{0, 0, 0});
    }
    uint64_t row_size(uint32_t r) const { return row_off[r + 1] - row_off[r]; }
    uint32_t id_at(uint32_t r, uint64_t j) const { return dense ? (uint32_t)j + (j >= r ? 1u : 0u) : query_ids[row_off[r] + j]; }
    const lzani_result& at(uint32_t r, uint64_t j) const { return res[row_off[r] + j]; }
    // result of parse(query = id, ref = r), or nullptr when the pair was not computed
    const lzani_result* find(uint32_t r, uint32_t id) const
    {
        if (dense) return id == r || id >= n ? nullptr : &res[row_off[r] + (id < r ? id : id - 1)];
        auto b = query_ids.begin() + (ptrdiff_t)row_off[r], e = query_ids.begin() + (ptrdiff_t)row_off[r + 1];
        auto it = std::lower_bound(b, e, id);
        return (it != e && *it == id) ? &res[(size_t)(it - query_ids.begin())] : nullptr;
    }
    // first position of row r whose query id exceeds r
    uint64_t first_above(uint32_t r) const
    {
        if (dense) return r;
        auto b = query_ids.begin() + (ptrdiff_t)row_off[r], e = query_ids.begin() + (ptrdiff_t)row_off[r + 1];
        return (uint64_t)(std::upper_bound(b, e, r) - b);
    }
};

struct EmitParams {
    std::string out_name, ids_name;
    bool single_txt = false, in_percent = false;
    std::vector<Comp> comps;
    uint64_t filter_mask = 0;
    double filter_vals[16] = {0};
    uint32_t threads = 1;
    std::string params_dump;                               // CParams::str() text for single-txt
    int mrd = 40;
};

// ---- one TSV line = one DIRECTION of an unordered pair --------------------------------------------------
// For the pair {a < b} the stage computed X = parse(query b, ref a) and Y = parse(query a, ref b).  A line
// describes one of them ("own") with the other ("rev") supplying the reference-side coverage; the two lines
// of a pair share tani.  All ratios are IEEE double divisions of exact integers, as in the reference
// (lz_matcher.cpp:434-447), so equal integers give byte-equal text.
struct Direction {
    uint32_t qidx, ridx, qlen, rlen;
    const std::string *qname, *rname;
    int32_t nt_match, nt_mismatch, num_alns;
    double tani, gani, ani, qcov, rcov;
};

inline Direction make_direction(const std::vector<Genome>& g, const std::vector<uint32_t>& len, uint32_t q, uint32_t r,
                                const lzani_result& own, const lzani_result& rev, double tani)
{
    Direction d;
    d.qidx = q; d.ridx = r; d.qlen = len[q]; d.rlen = len[r];
    d.qname = &g[q].name; d.rname = &g[r].name;
    d.nt_match = own.sym_in_matches; d.nt_mismatch = own.sym_in_literals; d.num_alns = own.no_components;
    const int32_t aligned = own.sym_in_matches + own.sym_in_literals, aligned_rev = rev.sym_in_matches + rev.sym_in_literals;
    d.tani = tani;
    d.gani = (double)own.sym_in_matches / d.qlen;
    d.ani = aligned != 0 ? (double)own.sym_in_matches / aligned : 0;
    d.qcov = (double)aligned / d.qlen;
    d.rcov = (double)aligned_rev / d.rlen;
    return d;
}

struct LineOut {
    std::string& s;
    char num[64];
    void u(uint64_t v) { s.append(num, uint_to_chars(v, num)); }
    void i(int64_t v) { if (v < 0) { s.push_back('-'); v = -v; } u((uint64_t)v); }
    void r(double v, int prec) { s.append(num, real_to_chars(v, num, prec)); }
};

// column writers, indexed by Comp (the enum's order); `m` = 100 with --out-in-percent, else 1
typedef void (*ColumnFn)(LineOut&, const Direction&, double m);
inline const ColumnFn* column_table()
{
    static const ColumnFn t[15] = {
        /* query       */ [](LineOut& o, const Direction& d, double) { o.s.append(*d.qname); },
        /* reference   */ [](LineOut& o, const Direction& d, double) { o.s.append(*d.rname); },
        /* qidx        */ [](LineOut& o, const Direction& d, double) { o.u(d.qidx); },
        /* ridx        */ [](LineOut& o, const Direction& d, double) { o.u(d.ridx); },
        /* qlen        */ [](LineOut& o, const Direction& d, double) { o.u(d.qlen); },
        /* rlen        */ [](LineOut& o, const Direction& d, double) { o.u(d.rlen); },
        /* tani        */ [](LineOut& o, const Direction& d, double m) { o.r(m * d.tani, 6); },
        /* gani        */ [](LineOut& o, const Direction& d, double m) { o.r(m * d.gani, 6); },
        /* ani         */ [](LineOut& o, const Direction& d, double m) { o.r(m * d.ani, 6); },
        /* qcov        */ [](LineOut& o, const Direction& d, double m) { o.r(m * d.qcov, 6); },
        /* rcov        */ [](LineOut& o, const Direction& d, double m) { o.r(m * d.rcov, 6); },
        /* len_ratio   */ [](LineOut& o, const Direction& d, double) {
            if (d.qlen && d.rlen) o.r(d.qlen < d.rlen ? (double)d.qlen / d.rlen : (double)d.rlen / d.qlen, 4);
            else o.s.push_back('0'); },
        /* nt_match    */ [](LineOut& o, const Direction& d, double) { o.i(d.nt_match); },
        /* nt_mismatch */ [](LineOut& o, const Direction& d, double) { o.i(d.nt_mismatch); },
        /* num_alns    */ [](LineOut& o, const Direction& d, double) { o.i(d.num_alns); }};
    return t;
}

// --out-filter: minimum values per ratio column (a line is dropped when any ratio is below its minimum)
struct Cut { double Direction::*field; double min; };
inline std::vector<Cut> make_cuts(const EmitParams& ep)
{
    std::vector<Cut> cuts;
    if (!ep.filter_mask) return cuts;
    static const std::pair<Comp, double Direction::*> ratio[] = {
        {Comp::tani, &Direction::tani}, {Comp::gani, &Direction::gani}, {Comp::ani, &Direction::ani},
        {Comp::qcov, &Direction::qcov}, {Comp::rcov, &Direction::rcov}};
    for (auto& x : ratio) cuts.push_back(Cut{x.second, ep.filter_vals[(int)x.first]});
    return cuts;
}

inline void write_direction(const Direction& d, const EmitParams& ep, const std::vector<Cut>& cuts, std::string& out)
{
    for (const Cut& c : cuts) if (d.*(c.field) < c.min) return;
    LineOut o{out, {0}};
    const ColumnFn* col = column_table();
    const double m = ep.in_percent ? 100 : 1;
    bool first = true;
    for (Comp c : ep.comps) {
        if (!first) out.push_back('\t');
        first = false;
        col[(int)c](o, d, m);
    }
    out.push_back('\n');
}

// Text of the unordered pairs {a < b} led by genome a (one reference row of the output file):
// two TSV lines per pair (b against a, then a against b), or one single-txt line.
inline void format_row(const std::vector<Genome>& g, const std::vector<uint32_t>& len, const PairTable& T, const EmitParams& ep,
                       const std::vector<Cut>& cuts, uint32_t a, std::string& out)
{
    const uint64_t cnt = T.row_size(a);
    for (uint64_t j = T.first_above(a); j < cnt; ++j) {
        const uint32_t b = T.id_at(a, j);
        const lzani_result& X = T.at(a, j);                 // parse(query = b, ref = a)
        const lzani_result* Y = T.find(b, a);               // parse(query = a, ref = b)
        if (!Y) continue;                                   // the reference asserts symmetry here (lz_matcher.cpp:417-418)
        if (ep.single_txt) {
            LineOut o{out, {0}};
            o.u(a); out.push_back(' '); o.u(b);
            for (const lzani_result* r : {Y, &X}) {
                out.push_back(' '); o.i(r->sym_in_matches); out.push_back(' '); o.i(r->sym_in_literals); out.push_back(' '); o.i(r->no_components);
            }
            out.push_back('\n');
            continue;
        }
        const double tani = (double)(X.sym_in_matches + Y->sym_in_matches) / (len[b] + len[a]);
        write_direction(make_direction(g, len, b, a, X, *Y, tani), ep, cuts, out);
        write_direction(make_direction(g, len, a, b, *Y, X, tani), ep, cuts, out);
    }
}

// The result files, written front to back: open() puts down the ids file and the header line, emit_rows(a0, a1)
// the text of the reference rows [a0, a1) -- which needs every pair {a, b > a} of those rows in both directions, no
// more -- and close() finishes.  A host that computes the all2all block by block (lz-ani's tiled matching) emits the
// rows of a block while the GPU works on the next one; store_results below is the same thing in one go.
class ResultWriter {
public:
    bool open(const std::vector<Genome>& g, const EmitParams& ep_)
    {
        ep = ep_;
        std::string fn_ids, fn_anis;
        if (!ep.single_txt) {
            fn_anis = ep.out_name;
            fn_ids = ep.ids_name;
            if (fn_ids.empty()) {                             // <stem>.ids<ext> (lz_matcher.cpp:295-302)
                auto p = fn_anis.rfind('.');
                fn_ids = p == std::string::npos ? fn_anis + ".ids" : fn_anis.substr(0, p) + ".ids" + fn_anis.substr(p);
            }
        } else fn_ids = ep.out_name;
        ofs.open(fn_ids, std::ios::binary);
        if (!ofs.is_open()) { std::cerr << "Cannot open output file: " << fn_ids << std::endl; return false; }
        // reported length: separators between the contigs of a multi-part item are not counted (lz_matcher.cpp:430-431)
        len.resize(g.size());
        for (size_t i = 0; i < g.size(); ++i) len[i] = (uint32_t)g[i].codes.size() - (g[i].no_parts - 1) * (uint32_t)ep.mrd;
        if (ep.single_txt) {
            ofs << ep.params_dump;
            ofs << "[no_input_sequences]\n" << g.size() << "\n[input_sequences]\n";
            for (size_t i = 0; i < g.size(); ++i) ofs << g[i].name << " " << len[i] << " " << g[i].no_parts << "\n";
            ofs << "[lz_similarities]\n";
        } else {
            ofs << "id\tseq_len\tno_parts\n";
            for (size_t i = 0; i < g.size(); ++i) ofs << g[i].name << "\t" << len[i] << "\t" << g[i].no_parts << "\n";
            ofs.close();
            ofs.open(fn_anis, std::ios::binary);
            if (!ofs.is_open()) { std::cerr << "Cannot open output file: " << fn_anis << std::endl; return false; }
            bool first = true;
            for (Comp c : ep.comps) {
                if (!first) ofs << "\t";
                first = false;
                for (auto& kv : comp_names()) if (kv.second == c) ofs << kv.first;
            }
            ofs << "\n";
        }
        cuts = make_cuts(ep);
        return true;
    }

    // Formatter threads fill the row strings of one block while the writer thread puts the previous block on disk,
    // in row order; two blocks of a few rows per thread are all that is ever held (a 10,000-genome row is ~1.8 MB of
    // text, the whole file 9 GB).
    void emit_rows(const std::vector<Genome>& g, const PairTable& T, size_t a0, size_t a1)
    {
        const uint32_t nt = std::max<uint32_t>(1, ep.threads);
        const size_t block = std::max<size_t>(8, 4 * (size_t)nt);
        text[0].resize(block); text[1].resize(block);
        for (size_t base = a0; base < a1; base += block, cur ^= 1) {
            const size_t cnt = std::min(block, a1 - base);
            std::vector<std::string>& buf = text[cur];
            std::atomic<size_t> next{0};
            auto worker = [&]() {
                for (;;) {
                    size_t k = next.fetch_add(1);
                    if (k >= cnt) break;
                    buf[k].clear();
                    format_row(g, len, T, ep, cuts, (uint32_t)(base + k), buf[k]);
                }
            };
            std::vector<std::thread> th;
            for (uint32_t t = 1; t < std::min<uint32_t>(nt, (uint32_t)cnt); ++t) th.emplace_back(worker);
            worker();
            for (auto& t : th) t.join();
            if (writer.joinable()) writer.join();              // the block before this one is on disk: its buffer is free again
            std::ofstream* o = &ofs;
            writer = std::thread([o, &buf, cnt]() { for (size_t k = 0; k < cnt; ++k) o->write(buf[k].data(), (std::streamsize)buf[k].size()); });
        }
    }

    bool close()
    {
        if (writer.joinable()) writer.join();
        ofs.close();
        return !ofs.fail();
    }
    ~ResultWriter() { if (writer.joinable()) writer.join(); }

private:
    EmitParams ep;
    std::ofstream ofs;
    std::vector<uint32_t> len;
    std::vector<Cut> cuts;
    std::vector<std::string> text[2];
    int cur = 0;
    std::thread writer;
};

inline bool store_results(const std::vector<Genome>& g, const PairTable& T, const EmitParams& ep)
{
    ResultWriter w;
    if (!w.open(g, ep)) return false;
    w.emit_rows(g, T, 0, T.n);
    w.close();
    return true;
}

}  // namespace host
