// ingest.h -- FASTA ingest, length-descending reorder and kmer-db filter of the `lz-ani` host binary.
//
// Behaviour follows the reference's host data services (studied, not copied):
//   CSeqReservoir::load_multifasta / load_fasta / append / reorder_items
//     (/root/reference/src/seq_reservoir.cpp:20-251, seq_reservoir.h:241-248)
//   CFilter::load_filter / reorder_items (/root/reference/src/filter.cpp:20-345)
//   split() (/root/reference/src/utils.cpp:16-37)
// Sequences are kept as one reservoir symbol code per byte (A0 C1 G2 T3, else 5): exactly what
// lzani_set_genomes takes; the 2-bit packing happens on the device.
#pragma once
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <string>
#include <vector>

namespace host {

struct Genome {
    std::string name;
    std::vector<uint8_t> codes;
    uint32_t no_parts = 1;          // always 1 in the reference (seq_reservoir.cpp:86)
};

inline const uint8_t* dna_code_table()
{
    static uint8_t t[256];
    static bool init = false;
    if (!init) {
        memset(t, 5, sizeof t);
        t['A'] = t['a'] = 0; t['C'] = t['c'] = 1; t['G'] = t['g'] = 2; t['T'] = t['t'] = 3;
        init = true;
    }
    return t;
}

// Streaming FASTA parser: the file goes through a 4 MB gzread window (gzread reads plain files transparently, so
// .gz and plain share one path; the reference sniffs the gzip magic, file_wrapper.h:541-550) and every base is
// turned into its symbol code where it lands -- no whole-file buffer, no text copy of a sequence.
// Line rules are those of the reference's stream_decompression::getline users: lines split at '\n', ONE trailing
// '\r' dropped, empty lines ignored, a line is a header iff its first byte is '>'.
struct FastaParser {
    std::vector<Genome>& out;
    const bool multisample;          // one item per '>' record (seq_reservoir.cpp:156-212) or one per file (90-153)
    const uint32_t sep_len;          // single-sample mode: N symbols between the contigs of a file
    const uint8_t* code = dna_code_table();
    Genome cur;
    bool have = false;               // a record (multisample) / the file's item (single) is open
    // line state
    bool at_line_start = true, in_header = false, pending_cr = false;
    size_t line_mark = 0;            // cur.codes.size() when the current sequence line began
    std::string header;

    FastaParser(std::vector<Genome>& o, bool multi, uint32_t sep) : out(o), multisample(multi), sep_len(sep) {}

    void begin_file(const std::string& fn)
    {
        at_line_start = true; in_header = false; pending_cr = false; header.clear();
        if (!multisample) { cur = Genome(); cur.name = std::filesystem::path(fn).filename().string(); have = true; }
        line_mark = cur.codes.size();
    }
    void close_record()
    {
        if (have) out.push_back(std::move(cur));
        cur = Genome(); have = false; line_mark = 0;
    }
    void end_header()
    {
        if (multisample) {
            close_record();
            size_t sp = header.find(' ');
            cur.name = sp == std::string::npos ? header : header.substr(0, sp);     // cut at the first space (77-81)
            have = !header.empty();                                                 // a bare '>' opens nothing: what follows it is dropped
        } else if (!cur.codes.empty()) cur.codes.insert(cur.codes.end(), sep_len, (uint8_t)5);
        header.clear();
        in_header = false;
    }
    void end_line()
    {
        if (in_header) end_header();
        at_line_start = true; pending_cr = false;
        line_mark = cur.codes.size();
    }
    void put(char c)
    {
        if (at_line_start) { at_line_start = false; if (c == '>') { in_header = true; return; } }
        if (in_header) header.push_back(c);
        else if (have) cur.codes.push_back(code[(uint8_t)c]);
    }
    void feed(const char* p, size_t n)
    {
        for (size_t k = 0; k < n; ++k) {
            const char c = p[k];
            if (c == '\n') { end_line(); continue; }
            if (pending_cr) { pending_cr = false; put('\r'); }                      // a '\r' inside a line is a symbol
            if (c == '\r') { pending_cr = true; continue; }
            if (!at_line_start && !in_header && have) {
                // fast path: the rest of a sequence line in one go
                size_t e = k;
                while (e < n && p[e] != '\n' && p[e] != '\r') ++e;
                const size_t at = cur.codes.size();
                cur.codes.resize(at + (e - k));
                for (size_t t = k; t < e; ++t) cur.codes[at + (t - k)] = code[(uint8_t)p[t]];
                k = e - 1;
                continue;
            }
            put(c);
        }
    }
    // end of a file.  Multisample mode never sees a last line without '\n' (the reference's loop gets < 0 from
    // getline for it, seq_reservoir.cpp:177-178): what it holds is dropped -- and a record whose header line is that
    // last line does not exist.  Single-sample mode keeps it.
    void end_file()
    {
        if (multisample) {
            if (!at_line_start) {
                if (in_header) { header.clear(); in_header = false; }
                else cur.codes.resize(line_mark);
            }
            close_record();
        } else {
            if (!at_line_start) { if (pending_cr) { /* trailing '\r' of the last line: dropped like any other */ } end_line(); }
            close_record();
        }
    }
};

inline bool load_sequences(const std::vector<std::string>& files, bool multisample, uint32_t sep_len, std::vector<Genome>& out)
{
    std::vector<char> buf(4u << 20);
    FastaParser fp(out, multisample, sep_len);
    for (const auto& fn : files) {
        gzFile f = gzopen(fn.c_str(), "rb");
        if (!f) { std::cerr << "Cannot open file: " << fn << std::endl; return false; }
        gzbuffer(f, 1 << 20);
        fp.begin_file(fn);
        for (;;) {
            int n = gzread(f, buf.data(), (unsigned)buf.size());
            if (n < 0) { gzclose(f); std::cerr << "Cannot open file: " << fn << std::endl; return false; }
            if (n == 0) break;
            fp.feed(buf.data(), (size_t)n);
        }
        gzclose(f);
        fp.end_file();
    }
    return true;
}
// One item per '>' record (seq_reservoir.cpp:156-212)
inline bool load_multifasta(const std::vector<std::string>& files, std::vector<Genome>& out) { return load_sequences(files, true, 0, out); }
// One item per file, contigs joined by `sep_len` N symbols, named after the file (seq_reservoir.cpp:90-153)
inline bool load_fasta(const std::vector<std::string>& files, uint32_t sep_len, std::vector<Genome>& out) { return load_sequences(files, false, sep_len, out); }

// Whole (small) text file into memory: the kmer-db filter reader below
inline bool slurp(const std::string& fn, std::string& out)
{
    gzFile f = gzopen(fn.c_str(), "rb");
    if (!f) return false;
    gzbuffer(f, 1 << 20);
    out.clear();
    std::vector<char> buf(1 << 22);
    for (;;) {
        int n = gzread(f, buf.data(), (unsigned)buf.size());
        if (n < 0) { gzclose(f); return false; }
        if (n == 0) break;
        out.append(buf.data(), (size_t)n);
    }
    gzclose(f);
    return true;
}

// Lines of a buffer as the reference's stream_decompression::getline yields them: split at '\n',
// one trailing '\r' removed.  `last_unterminated` reports a final piece with no newline.
struct LineReader {
    const std::string& s;
    size_t pos = 0;
    explicit LineReader(const std::string& str) : s(str) {}
    // returns false at end; `terminated` = the line ended with '\n'
    bool next(std::string& line, bool& terminated)
    {
        if (pos >= s.size()) return false;
        size_t q = s.find('\n', pos);
        terminated = q != std::string::npos;
        size_t e = terminated ? q : s.size();
        line.assign(s, pos, e - pos);
        pos = terminated ? q + 1 : s.size();
        if (!line.empty() && line.back() == '\r') line.pop_back();
        return true;
    }
};

// reorder_items (seq_reservoir.cpp:215-251): stable sort by (len - 2*no_parts) as uint32 descending,
// then name ascending; returns old -> new.
inline std::vector<uint32_t> reorder(std::vector<Genome>& g)
{
    std::vector<uint32_t> idx(g.size());
    for (uint32_t i = 0; i < idx.size(); ++i) idx[i] = i;
    std::stable_sort(idx.begin(), idx.end(), [&](uint32_t x, uint32_t y) {
        uint32_t xs = (uint32_t)g[x].codes.size() - g[x].no_parts * 2;
        uint32_t ys = (uint32_t)g[y].codes.size() - g[y].no_parts * 2;
        if (xs != ys) return xs > ys;
        return g[x].name < g[y].name;
    });
    std::vector<uint32_t> map(g.size());
    std::vector<Genome> r;
    r.reserve(g.size());
    for (uint32_t i = 0; i < idx.size(); ++i) { map[idx[i]] = i; r.push_back(std::move(g[idx[i]])); }
    g = std::move(r);
    return map;
}

inline std::vector<std::string> split(const std::string& str, char sep)      // utils.cpp:16-37
{
    std::vector<std::string> parts;
    std::string s;
    for (char c : str) {
        if (c == sep) { parts.push_back(s); s.clear(); }
        else s.push_back(c);
    }
    if (!s.empty()) parts.push_back(s);
    return parts;
}

// kmer-db sparse text (filter.cpp:20-298): header ",name1,...,nameN," after a free-form first
// field; rows "name,idx:val,..." with 1-based idx; keep val >= thr; symmetrise.  Row ids count the
// data lines longer than 2 characters (the multi-threaded reader's rule, filter.cpp:104-111).
struct Filter {
    std::vector<std::string> names;
    std::vector<std::vector<uint32_t>> rows;
    bool empty() const { return rows.empty(); }
    uint64_t size() const { uint64_t s = 0; for (auto& r : rows) s += r.size(); return s; }
};

inline bool load_filter(const std::string& fn, double thr, Filter& f)
{
    std::string data, line;
    if (!slurp(fn, data)) { std::cerr << "Cannot open file: " << fn << std::endl; return false; }
    LineReader lr(data);
    bool term;
    if (!lr.next(line, term)) line.clear();
    f.names = split(line, ',');
    if (f.names.size() <= 2) { std::cerr << "Incorrect kmer-db filter file\n"; return false; }
    f.names.erase(f.names.begin());
    f.rows.assign(f.names.size(), {});
    uint32_t id = 0;
    while (lr.next(line, term)) {
        if (line.length() <= 2) continue;
        if (id >= f.rows.size()) { std::cerr << "Incorrect kmer-db filter file\n"; return false; }
        auto parts = split(line, ',');
        for (size_t j = 1; j < parts.size(); ++j) {
            auto elem = split(parts[j], ':');
            if (elem.size() != 2) continue;
            double val = atof(elem[1].c_str());
            long k = atol(elem[0].c_str()) - 1;
            if (val >= thr) {
                if (k < 0 || (size_t)k >= f.rows.size()) { std::cerr << "Incorrect kmer-db filter file\n"; return false; }
                f.rows[id].push_back((uint32_t)k);
            }
        }
        ++id;
    }
    std::vector<uint32_t> first(f.rows.size());
    for (size_t i = 0; i < f.rows.size(); ++i) first[i] = (uint32_t)f.rows[i].size();
    for (size_t i = 0; i < f.rows.size(); ++i)
        for (uint32_t k = 0; k < first[i]; ++k) f.rows[f.rows[i][k]].push_back((uint32_t)i);
    return true;
}

inline void reorder_filter(Filter& f, const std::vector<uint32_t>& map)
{
    if (f.rows.empty()) return;
    std::vector<std::vector<uint32_t>> r(f.rows.size());
    for (size_t i = 0; i < map.size(); ++i) r[map[i]] = std::move(f.rows[i]);
    for (auto& row : r) for (auto& x : row) x = map[x];
    f.rows = std::move(r);
}

}  // namespace host
