// emit.h -- result files of the `lz-ani` host binary: ids TSV, ANI TSV, single-txt.
//
// Behaviour follows CLZMatcher::store_results (/root/reference/src/lz_matcher.cpp:280-579) and the
// number formatting of refresh::real_to_pchar (/root/reference/libs/refresh/conversions/lib/
// numeric_conversions.h:228-300, 341-390): shortest round-trip decimal (here std::to_chars, the same
// digits as the reference's dragonbox), rounded half-up to `prec` significant digits, then the
// reference's fixed/exponent layout.  Byte-identical output is the contract (BASELINE config 1).
#pragma once
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

struct IdResult { uint32_t id; lzani_result r; };
using ResultRows = std::vector<std::vector<IdResult>>;     // results[ref] sorted by id (lz_matcher.cpp:253-255)

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

inline const IdResult* find_id(const std::vector<IdResult>& row, uint32_t id)
{
    auto it = std::lower_bound(row.begin(), row.end(), id, [](const IdResult& a, uint32_t v) { return a.id < v; });
    return (it != row.end() && it->id == id) ? &*it : nullptr;
}

// One reference row -> text (lz_matcher.cpp:400-562)
inline void format_row(const std::vector<Genome>& g, const ResultRows& results, const EmitParams& ep, uint32_t ref_id,
                       std::string& out)
{
    const double mult = ep.in_percent ? 100 : 1;
    char num[64];
    auto put_u = [&](uint64_t v, char sep) { out.append(num, uint_to_chars(v, num)); out.push_back(sep); };
    auto put_i = [&](int64_t v, char sep) { if (v < 0) { out.push_back('-'); v = -v; } put_u((uint64_t)v, sep); };
    auto put_r = [&](double v, int prec, char sep) { out.append(num, real_to_chars(v, num, prec)); out.push_back(sep); };
    for (const auto& q : results[ref_id]) {
        if (ref_id >= q.id) continue;
        const IdResult* p = find_id(results[q.id], ref_id);
        if (!p) continue;                                   // the reference asserts symmetry here
        if (ep.single_txt) {
            put_u(ref_id, ' '); put_u(q.id, ' ');
            put_i(p->r.sym_in_matches, ' '); put_i(p->r.sym_in_literals, ' '); put_i(p->r.no_components, ' ');
            put_i(q.r.sym_in_matches, ' '); put_i(q.r.sym_in_literals, ' '); put_i(q.r.no_components, '\n');
            continue;
        }
        const std::string* names[2] = {&g[ref_id].name, &g[q.id].name};
        const uint32_t ids[2] = {ref_id, q.id};
        auto seq_len = [&](uint32_t i) { return (uint32_t)g[i].codes.size() - (g[i].no_parts - 1) * (uint32_t)ep.mrd; };
        const uint32_t len[2] = {seq_len(q.id), seq_len(ref_id)};
        const int32_t mat[2] = {q.r.sym_in_matches, p->r.sym_in_matches};
        const int32_t lit[2] = {q.r.sym_in_literals, p->r.sym_in_literals};
        const int32_t reg[2] = {q.r.no_components, p->r.no_components};
        const double tani = (double)(mat[0] + mat[1]) / (len[0] + len[1]);
        const double gani[2] = {(double)mat[0] / len[0], (double)mat[1] / len[1]};
        const double ani[2] = {mat[0] + lit[0] != 0 ? (double)mat[0] / (mat[0] + lit[0]) : 0,
                               mat[1] + lit[1] != 0 ? (double)mat[1] / (mat[1] + lit[1]) : 0};
        const double cov[2] = {(double)(mat[0] + lit[0]) / len[0], (double)(mat[1] + lit[1]) / len[1]};
        for (int i = 0; i < 2; ++i) {
            if (ep.filter_mask != 0) {
                if (gani[i] < ep.filter_vals[(int)Comp::gani]) continue;
                if (ani[i] < ep.filter_vals[(int)Comp::ani]) continue;
                if (tani < ep.filter_vals[(int)Comp::tani]) continue;
                if (cov[i] < ep.filter_vals[(int)Comp::qcov]) continue;
                if (cov[!i] < ep.filter_vals[(int)Comp::rcov]) continue;
            }
            for (Comp oc : ep.comps) {
                switch (oc) {
                case Comp::ridx: put_u(ids[i], '\t'); break;
                case Comp::qidx: put_u(ids[!i], '\t'); break;
                case Comp::reference: out.append(*names[i]); out.push_back('\t'); break;
                case Comp::query: out.append(*names[!i]); out.push_back('\t'); break;
                case Comp::qcov: put_r(mult * cov[i], 6, '\t'); break;
                case Comp::rcov: put_r(mult * cov[!i], 6, '\t'); break;
                case Comp::gani: put_r(mult * gani[i], 6, '\t'); break;
                case Comp::rlen: put_u(len[!i], '\t'); break;
                case Comp::qlen: put_u(len[i], '\t'); break;
                case Comp::len_ratio:
                    if (len[0] && len[1]) {
                        double lr = len[i] < len[!i] ? (double)len[i] / len[!i] : (double)len[!i] / len[i];
                        put_r(lr, 4, '\t');
                    } else { out.push_back('0'); out.push_back('\t'); }
                    break;
                case Comp::ani: put_r(mult * ani[i], 6, '\t'); break;
                case Comp::num_alns: put_i(reg[i], '\t'); break;
                case Comp::nt_mismatch: put_i(lit[i], '\t'); break;
                case Comp::nt_match: put_i(mat[i], '\t'); break;
                case Comp::tani: put_r(mult * tani, 6, '\t'); break;
                }
            }
            if (!ep.comps.empty()) out.pop_back();
            out.push_back('\n');
        }
    }
}

inline bool store_results(const std::vector<Genome>& g, const ResultRows& results, const EmitParams& ep)
{
    std::string fn_ids, fn_anis;
    if (!ep.single_txt) {
        fn_anis = ep.out_name;
        fn_ids = ep.ids_name;
        if (fn_ids.empty()) {                             // <stem>.ids<ext> (lz_matcher.cpp:295-302)
            auto p = fn_anis.rfind('.');
            fn_ids = p == std::string::npos ? fn_anis + ".ids" : fn_anis.substr(0, p) + ".ids" + fn_anis.substr(p);
        }
    } else fn_ids = ep.out_name;

    std::ofstream ofs(fn_ids, std::ios::binary);
    if (!ofs.is_open()) { std::cerr << "Cannot open output file: " << fn_ids << std::endl; return false; }
    auto seq_len = [&](size_t i) { return (uint32_t)g[i].codes.size() - (g[i].no_parts - 1) * (uint32_t)ep.mrd; };
    if (ep.single_txt) {
        ofs << ep.params_dump;
        ofs << "[no_input_sequences]\n" << g.size() << "\n[input_sequences]\n";
        for (size_t i = 0; i < g.size(); ++i) ofs << g[i].name << " " << seq_len(i) << " " << g[i].no_parts << "\n";
        ofs << "[lz_similarities]\n";
    } else {
        ofs << "id\tseq_len\tno_parts\n";
        for (size_t i = 0; i < g.size(); ++i) ofs << g[i].name << "\t" << seq_len(i) << "\t" << g[i].no_parts << "\n";
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

    // formatter threads fill per-row strings in blocks; the main thread writes them in row order
    const size_t n = results.size();
    const size_t block = 256;
    std::vector<std::string> text(std::min(n, block * 64));
    for (size_t base = 0; base < n; base += text.size()) {
        size_t cnt = std::min(text.size(), n - base);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (;;) {
                size_t k = next.fetch_add(1);
                if (k >= cnt) break;
                text[k].clear();
                format_row(g, results, ep, (uint32_t)(base + k), text[k]);
            }
        };
        std::vector<std::thread> th;
        uint32_t nt = std::max<uint32_t>(1, std::min<uint32_t>(ep.threads, (uint32_t)cnt));
        for (uint32_t t = 1; t < nt; ++t) th.emplace_back(worker);
        worker();
        for (auto& t : th) t.join();
        for (size_t k = 0; k < cnt; ++k) ofs.write(text[k].data(), (std::streamsize)text[k].size());
    }
    ofs.close();
    return true;
}

}  // namespace host
