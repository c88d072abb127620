// synth5 -- seeded synthetic genome sets with HEAVY-TAILED family sizes and the matching kmer-db filter file
// (SURVEY 8(d) config 5 / BASELINE configs[4]: "100,000 viral genomes with --flt-kmerdb prefilter at 0.3").
//
//   synth5 <n> <seed> <lmin> <lmax> <out.fna> <out.kmerdb> <out.bin> [max_family=1000] [threshold=0.3] [fixed_family=0] [dmin=0.01] [dmax=0.15]
//   (fixed_family > 0: every family has that many members -- BASELINE configs[3]: 1,000 x 5 Mbp in families of 10,
//   d ~ U(0.005, 0.08): synth5 1000 3 4500000 5500000 ... 1000 0.3 10 0.005 0.08)
//
// Families: 40 % singletons, 40 % of size 2-10, 15 % of size 11-100, 5 % of size 101-max_family.  Member 0 of a
// family is a uniform random ancestor of length U[lmin,lmax]; the others are the ancestor with substitutions at
// rate d ~ U(0.01,0.15), indels at rate d/10 (length 1-10) and, with probability 0.2, one 1-5 kbp inversion.
// The filter file has the text format kmer-db writes and CFilter::load_filter reads
// (/root/reference/src/filter.cpp:34-42, 61-81; sample: /root/reference/example/fltr.txt): a header line
// "kmer-length: 18 fraction: 1 ,name1,...,nameN," and one row "name,idx:val,...," per genome with 1-based
// indices of EARLIER genomes.  Same-family pairs get val = 1 - (d_i + d_j)/2 (always kept at 0.3); every row
// also gets about one random cross-family pair with val in [0.2, 0.4) (kept or not by the threshold).
// out.bin is a sidecar for the tests: u64 n, u64 off[n+1], then the symbol codes (0..3) of every genome.
// Prints one JSON line: genomes, families, kept unordered pairs at the threshold, largest row.
//
// A self-contained generator (splitmix64), written in C++ because the set sizes of this configuration
// (2*10^4 .. 10^5 genomes) take minutes in the numpy generator of tools/synth_genomes.py.
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef uint64_t u64;

struct Rng {
    u64 s;
    explicit Rng(u64 seed) : s(seed) {}
    u64 next()
    {
        u64 z = (s += 0x9E3779B97F4A7C15ULL);
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
        return z ^ (z >> 31);
    }
    double uni() { return (double)(next() >> 11) * (1.0 / 9007199254740992.0); }
    u64 range(u64 lo, u64 hi) { return lo + next() % (hi - lo + 1); }     // inclusive
};

static void mutate(const std::vector<uint8_t>& anc, double d, Rng& r, std::vector<uint8_t>& out)
{
    out.clear();
    out.reserve(anc.size() + anc.size() / 50);
    const size_t n = anc.size();
    size_t p = 0;
    while (p < n) {
        const double u = r.uni();
        if (u < d) out.push_back((uint8_t)((anc[p] + 1 + r.next() % 3) & 3));
        else if (u < d + d / 10.0) {
            const u64 k = r.next();
            const size_t ln = 1 + (k >> 1) % 10;
            if (k & 1) { for (size_t j = 0; j < ln; ++j) out.push_back((uint8_t)(r.next() & 3)); out.push_back(anc[p]); }
            else { p += ln; continue; }
        } else out.push_back(anc[p]);
        ++p;
    }
    if (r.uni() < 0.2 && out.size() > 12000) {
        const size_t ln = (size_t)r.range(1000, 5000), s = (size_t)r.range(0, out.size() - ln);
        std::reverse(out.begin() + s, out.begin() + s + ln);
        for (size_t j = s; j < s + ln; ++j) out[j] = (uint8_t)(3 - out[j]);
    }
}

int main(int argc, char** argv)
{
    if (argc < 8) { fprintf(stderr, "usage: synth5 n seed lmin lmax out.fna out.kmerdb out.bin [max_family] [threshold]\n"); return 2; }
    const u64 n = strtoull(argv[1], 0, 10), seed = strtoull(argv[2], 0, 10);
    const u64 lmin = strtoull(argv[3], 0, 10), lmax = strtoull(argv[4], 0, 10);
    const u64 maxfam = argc > 8 ? strtoull(argv[8], 0, 10) : 1000;
    const double thr = argc > 9 ? atof(argv[9]) : 0.3;
    const u64 fixed_fam = argc > 10 ? strtoull(argv[10], 0, 10) : 0;
    const double dmin = argc > 11 ? atof(argv[11]) : 0.01, dmax = argc > 12 ? atof(argv[12]) : 0.15;
    FILE* ffa = fopen(argv[5], "wb");
    FILE* fdb = fopen(argv[6], "wb");
    FILE* fbin = fopen(argv[7], "wb");
    if (!ffa || !fdb || !fbin) { fprintf(stderr, "cannot open outputs\n"); return 1; }
    static char iobuf[3][1 << 22];
    setvbuf(ffa, iobuf[0], _IOFBF, sizeof iobuf[0]);
    setvbuf(fdb, iobuf[1], _IOFBF, sizeof iobuf[1]);
    setvbuf(fbin, iobuf[2], _IOFBF, sizeof iobuf[2]);

    Rng r(seed * 0x2545F4914F6CDD1DULL + 77);
    // family plan
    std::vector<u64> fam_first, fam_size;
    for (u64 at = 0; at < n;) {
        const double u = r.uni();
        u64 sz = u < 0.4 ? 1 : u < 0.8 ? r.range(2, 10) : u < 0.95 ? r.range(11, 100) : r.range(101, std::max<u64>(101, maxfam));
        if (fixed_fam) sz = fixed_fam;
        sz = std::min(sz, n - at);
        fam_first.push_back(at); fam_size.push_back(sz);
        at += sz;
    }
    std::vector<std::string> names(n);
    std::vector<double> div(n, 0.0);
    std::vector<u64> fam_of(n);
    for (size_t f = 0; f < fam_first.size(); ++f)
        for (u64 m = 0; m < fam_size[f]; ++m) {
            char nm[64];
            snprintf(nm, sizeof nm, "g%06llu_f%zu_m%llu", (unsigned long long)(fam_first[f] + m), f, (unsigned long long)m);
            names[fam_first[f] + m] = nm;
            fam_of[fam_first[f] + m] = f;
        }
    // header of the filter file
    fputs("kmer-length: 18 fraction: 1 ,", fdb);
    for (u64 i = 0; i < n; ++i) { fputs(names[i].c_str(), fdb); fputc(',', fdb); }
    fputc('\n', fdb);

    std::vector<u64> off(n + 1, 0);
    fwrite(&n, 8, 1, fbin);
    fwrite(off.data(), 8, n + 1, fbin);                       // placeholder, rewritten at the end
    std::vector<uint8_t> anc, g;
    std::string line;
    u64 kept = 0, biggest_row = 0;
    std::vector<u64> row_len(n, 0);
    for (size_t f = 0; f < fam_first.size(); ++f) {
        const u64 L = r.range(lmin, lmax);
        anc.resize(L);
        for (u64 j = 0; j < L; j += 32) {
            u64 w = r.next();
            for (u64 k = j; k < std::min(L, j + 32); ++k, w >>= 2) anc[k] = (uint8_t)(w & 3);
        }
        for (u64 m = 0; m < fam_size[f]; ++m) {
            const u64 i = fam_first[f] + m;
            const std::vector<uint8_t>* s = &anc;
            if (m) { div[i] = dmin + (dmax - dmin) * r.uni(); mutate(anc, div[i], r, g); s = &g; }
            off[i + 1] = off[i] + s->size();
            fwrite(s->data(), 1, s->size(), fbin);
            fputc('>', ffa); fputs(names[i].c_str(), ffa); fputc('\n', ffa);
            line.clear();
            for (size_t k = 0; k < s->size(); ++k) {
                line.push_back("ACGT"[(*s)[k]]);
                if (line.size() == 70) { line.push_back('\n'); fwrite(line.data(), 1, line.size(), ffa); line.clear(); }
            }
            if (!line.empty()) { line.push_back('\n'); fwrite(line.data(), 1, line.size(), ffa); }
            // filter row: earlier members of the family, plus ~one random earlier genome of another family
            fputs(names[i].c_str(), fdb); fputc(',', fdb);
            u64 cross = n;                                          // none
            if (fam_first[f] > 0 && r.uni() < 0.9) cross = r.range(0, fam_first[f] - 1);
            char item[64];
            auto put = [&](u64 j, double v) {
                snprintf(item, sizeof item, "%llu:%.6f,", (unsigned long long)(j + 1), v);
                fputs(item, fdb);
                const char* colon = item;
                while (*colon != ':') ++colon;
                if (atof(colon + 1) >= thr) { ++kept; ++row_len[i]; ++row_len[j]; }       // the value as a reader parses it
            };
            if (cross < n) put(cross, 0.2 + 0.2 * r.uni());
            for (u64 j = fam_first[f]; j < i; ++j) put(j, 1.0 - 0.5 * (div[i] + div[j]));
            fputc('\n', fdb);
        }
    }
    for (u64 i = 0; i < n; ++i) biggest_row = std::max(biggest_row, row_len[i]);
    fseek(fbin, 8, SEEK_SET);
    fwrite(off.data(), 8, n + 1, fbin);
    fclose(ffa); fclose(fdb); fclose(fbin);
    printf("{\"genomes\": %llu, \"families\": %zu, \"pairs_kept\": %llu, \"largest_row\": %llu, \"bases\": %llu}\n",
           (unsigned long long)n, fam_first.size(), (unsigned long long)kept, (unsigned long long)biggest_row, (unsigned long long)off[n]);
    return 0;
}
