// ref_driver.cpp -- TEST INFRASTRUCTURE ONLY.
//
// A thin C-ABI driver around the *reference's own* CParser (compiled by oracle/Makefile
// straight from /root/reference/src/parser.cpp into oracle/_ref/libref_lzani.so; no
// reference source is copied into this repository).  It exists to
//   (1) pin the CPU restatement (lzani_oracle.c) against the real implementation,
//   (2) generate the golden vectors under tests/golden/ (oracle/make_goldens.py),
//   (3) serve as bench.py's cpu_baseline of kind "reference" (the built .so travels to
//       the GPU box; /root/reference does not).
// The loop in ref_all2all mirrors CLZMatcher::do_matching (lz_matcher.cpp:172-277):
// threads self-schedule over reference ids, one private CParser per thread.
#include <atomic>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "parser.h"   // the reference's header, found through -I/root/reference/src
#include "../libs/refresh/conversions/lib/numeric_conversions.h"   // the reference's number formatting (header only)

namespace {

CParams make_params(const int32_t* p8)
{
    CParams p;
    p.min_anchor_len = p8[0];
    p.min_seed_len = p8[1];
    p.max_dist_in_ref = p8[2];
    p.max_dist_in_query = p8[3];
    p.min_region_len = p8[4];
    p.approx_window = p8[5];
    p.approx_mismatches = p8[6];
    p.approx_run_len = p8[7];
    return p;
}

// Same packing CSeqReservoir::append applies (seq_reservoir.cpp:39-57): base-6 triples.
std::vector<uint8_t> pack3(const uint8_t* codes, uint32_t len)
{
    std::vector<uint8_t> out((len + 2) / 3, 0);
    for (uint32_t i = 0; i < len; ++i) {
        uint8_t c = codes[i] < 4 ? codes[i] : 5;
        uint32_t w = (i % 3 == 0) ? 36 : (i % 3 == 1) ? 6 : 1;
        out[i / 3] = (uint8_t)(out[i / 3] + w * c);
    }
    return out;
}

}  // namespace

extern "C" {

// One directed pair through the reference parser.  res = {sym_in_matches, sym_in_literals, no_components}.
// If regions != nullptr, get_parsing() is copied out as 6 ints per region
// (ref_start, ref_end, seq_start, seq_end, num_matches, num_mismatches).
int ref_pair(const uint8_t* ref, uint32_t ref_len, const uint8_t* qry, uint32_t qry_len,
             const int32_t* p8, int32_t* res, int32_t* regions, uint32_t max_regions, uint32_t* n_regions)
{
    CParams params = make_params(p8);
    CParser parser(params);
    auto pr = pack3(ref, ref_len), pq = pack3(qry, qry_len);
    parser.prepare_reference(seq_view(pr.data(), ref_len, internal_packing_t::three_in_byte), 1);
    parser.prepare_data(seq_view(pq.data(), qry_len, internal_packing_t::three_in_byte), 1);
    parser.parse();
    if (n_regions) {
        auto v = parser.get_parsing();
        *n_regions = (uint32_t)v.size();
        for (uint32_t k = 0; k < v.size() && k < max_regions && regions; ++k) {
            int32_t* o = regions + 6 * k;
            o[0] = v[k].ref_start; o[1] = v[k].ref_end; o[2] = v[k].seq_start;
            o[3] = v[k].seq_end; o[4] = v[k].num_matches; o[5] = v[k].num_mismatches;
        }
    }
    results_t r = parser.calc_stats();
    res[0] = r.sym_in_matches; res[1] = r.sym_in_literals; res[2] = r.no_components;
    return 0;
}

// Rows of directed pairs, CSR.  For row k the reference is ref_ids[k] and the queries are
// query_ids[row_off[k] .. row_off[k+1]); out is CSR-aligned, 3 ints per pair.
int ref_rows(uint32_t n, const uint8_t* const* codes, const uint32_t* len, const int32_t* p8,
             uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off, const uint32_t* query_ids,
             uint32_t n_threads, int32_t* out)
{
    CParams params = make_params(p8);
    std::vector<std::vector<uint8_t>> packed(n);
    for (uint32_t i = 0; i < n; ++i) packed[i] = pack3(codes[i], len[i]);

    std::atomic<uint32_t> next{0};
    auto worker = [&]() {
        CParser parser(params);
        for (;;) {
            uint32_t k = next.fetch_add(1);
            if (k >= n_rows) break;
            uint32_t r = ref_ids[k];
            parser.prepare_reference(seq_view(packed[r].data(), len[r], internal_packing_t::three_in_byte), 1);
            for (uint64_t e = row_off[k]; e < row_off[k + 1]; ++e) {
                uint32_t q = query_ids[e];
                parser.prepare_data(seq_view(packed[q].data(), len[q], internal_packing_t::three_in_byte), 1);
                parser.parse();
                results_t s = parser.calc_stats();
                out[3 * e] = s.sym_in_matches; out[3 * e + 1] = s.sym_in_literals; out[3 * e + 2] = s.no_components;
            }
        }
    };
    if (n_threads <= 1) worker();
    else {
        std::vector<std::thread> th;
        for (uint32_t t = 0; t < n_threads; ++t) th.emplace_back(worker);
        for (auto& t : th) t.join();
    }
    return 0;
}

// refresh::real_to_pchar (numeric_conversions.h:341-390): the TSV number formatting, for pinning the
// host emitter.  Returns the length written (no terminator counted).
int ref_format_real(double v, int prec, char* out)
{
    size_t n = refresh::real_to_pchar(v, out, (size_t)prec, (char)0);
    return (int)n - 1;
}

// CParams::std_comp / parse_output_format (params.h:65-69, 169-198): the expansion of an --out-format
// string into column names, comma separated; "!<token>" if the reference rejects a component.
int ref_expand_output_format(const char* fmt, char* out, int cap)
{
    CParams p;
    std::string bad = p.parse_output_format(fmt);
    std::string s;
    if (!bad.empty()) s = "!" + bad;
    else
        for (auto c : p.output_components) { if (!s.empty()) s += ","; s += p.comp_id_name[c]; }
    if ((int)s.size() + 1 > cap) return -1;
    memcpy(out, s.c_str(), s.size() + 1);
    return (int)s.size();
}

}  // extern "C"
