// lz-ani (MI355X build) -- the reference's command line on top of the HIP pair engine.
//
// Drop-in surface: /root/reference/src/lz-ani.cpp:39-355 (modes, flags, defaults, exit codes) and the
// stage sequence of CLZMatcher::run_all2all (/root/reference/src/lz_matcher.cpp:582-617):
//   load sequences -> load filter -> check names -> reorder -> LZ matching -> store results.
// The matching stage is one call per GPU into the C-ABI of include/lzani.h (liblzani_hip.so, loaded
// with dlopen so that this binary builds and its ingest/emit code is testable without ROCm present).
// Extras over the reference: --gpus <n> (rows dealt cyclically over n GPUs), --device <id>,
// and the test seams --results-out / --results-in (raw int triples of the matching stage).
#include <dlfcn.h>

#include <chrono>
#include <condition_variable>
#include <deque>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iterator>
#include <sstream>
#include <thread>

#include "emit.h"
#include "ingest.h"

using namespace host;
using namespace std;

static const char* VER = "1.2.3";
static const char* INFO = "lz-ani 1.2.3 (2024-11-02) by Sebastian Deorowicz, Adam Gudys -- MI355X (gfx950) HIP engine";

struct Params {
    uint32_t verbosity = 1, threads = 0;
    lzani_params lz;
    bool multisample = true, in_percent = false, single_txt = false;
    double filter_thr = 0;
    vector<string> inputs;
    string out, out_ids, out_aln, filter_fn, out_format = "standard";
    vector<Comp> comps;
    uint64_t flt_mask = 0;
    double flt_vals[16] = {0};
    int gpus = 1, device = 0;
    string results_out, results_in;
};

static Params P;

static string parse_output_format(const string& of)          // params.h:169-198
{
    P.comps.clear();
    vector<string> names;
    for (const auto& x : split(of, ',')) {
        auto m = comp_metas().find(x);
        if (m == comp_metas().end()) names.push_back(x);
        else for (const auto& y : split(m->second, ',')) names.push_back(y);
    }
    for (const auto& x : names) {
        auto p = comp_names().find(x);
        if (p == comp_names().end()) return x;
        P.comps.push_back(p->second);
    }
    return "";
}

static void usage()
{
    cerr << INFO << "\n"
         << "Tool for rapid determination of similarities among sets of DNA sequences\n"
         << "Usage:\nlz-ani <mode> [options]\nModes:\n  all2all                        - all to all\n"
         << "Options - input specification:\n"
         << "      --in-fasta <file_name>     - FASTA file (for multisample-fasta mode)\n"
         << "      --in-txt <file_name>       - text file with FASTA file names\n"
         << "      --in-dir <path>            - directory with FASTA files\n"
         << "      --multisample-fasta <bool> - multi sample FASTA input (default: true)\n"
         << "      --flt-kmerdb <fn> <float>  - filtering file (kmer-db output) and threshold\n"
         << "Options - output specification:\n"
         << "  -o, --out <file_name>          - output file name\n"
         << "      --out-ids <file_name>      - output file name for ids file (optional)\n"
         << "      --out-alignment <file_name>- output file name for ids file (optional)\n"
         << "      --out-in-percent <bool>    - output in percent (default: false)\n"
         << "      --out-type <type>          - one of:\n"
         << "                                   tsv - two tsv files with: results defined by --out-format and sequence ids (default)\n"
         << "                                   single-txt - combined results in single txt file\n"
         << "      --out-format <type>        - comma-separated list of values: \n"
         << "                                   query,reference,qidx,ridx,qlen,rlen,tani,gani,ani,qcov,rcov,len_ratio,nt_match,nt_mismatch,num_alns\n"
         << "                                   you can include also meta-names:\n";
    for (auto& kv : comp_metas()) cerr << "                                   " << kv.first << "=" << kv.second << "\n";
    cerr << "                                   (default: standard)\n"
         << "      --out-filter <par> <float> - store only results with <par> (can be: tani, gani, ani, cov) at least <float>; can be used multiple times\n"
         << "Options - LZ-parsing-related:\n"
         << "  -a, --mal <int>                - min. anchor length (default: 11)\n"
         << "  -s, --msl <int>                - min. seed length (default: 7)\n"
         << "  -r, --mrd <int>                - max. dist. between approx. matches in reference (default: 40)\n"
         << "  -q, --mqd <int>                - max. dist. between approx. matches in query (default: 40)\n"
         << "  -g, --reg <int>                - min. considered region length (default: 35)\n"
         << "      --aw <int>                 - approx. window length (default: 15)\n"
         << "      --am <int>                 - max. no. of mismatches in approx. window (default: 7)\n"
         << "      --ar <int>                 - min. length of run ending approx. extension (default: 3)\n"
         << "Options - other:\n"
         << "  -t, --threads <int>            - no of host threads; 0 means auto-detect (default: 0)\n"
         << "  -V, --verbose <int>            - verbosity level (default: 1)\n"
         << "      --gpus <int>               - number of GPUs to shard the reference rows over (default: 1)\n"
         << "      --device <int>             - first HIP device ordinal (default: 0)\n";
}

static bool parse_bool(const char* v, bool& out) { if (v == "true"s) out = true; else if (v == "false"s) out = false; else return false; return true; }

// Mirrors parse_params (lz-ani.cpp:105-336), including its return/exit conventions.
static bool parse_params(int argc, char** argv)
{
    if (argc == 2 && argv[1] == "--version"s) { cerr << VER << endl; return true; }
    if (argc < 3) { usage(); return false; }
    if (argv[1] != "all2all"s) { cerr << "Unknown mode: " << argv[1] << endl; usage(); return false; }
    lzani_params& z = P.lz;
    for (int i = 2; i < argc;) {
        string par = argv[i];
        auto has = [&](int k) { return i + k < argc; };
        if (par == "--in-txt" && has(1)) {
            ifstream ifs(argv[i + 1]);
            if (!ifs.is_open()) { cerr << "Cannot open file: " << argv[i + 1] << endl; return false; }
            P.inputs.assign(istream_iterator<string>(ifs), istream_iterator<string>());
            if (P.inputs.empty()) return false;
            i += 2;
        } else if (par == "--in-dir" && has(1)) {
            try {
                P.inputs.clear();
                for (const auto& fs : filesystem::directory_iterator(filesystem::path(argv[i + 1]))) P.inputs.push_back(fs.path().string());
            } catch (...) { cerr << "Non-existing directory: " << argv[i + 1] << endl; return false; }
            if (P.inputs.empty()) return false;
            i += 2;
        } else if (par == "--in-fasta" && has(1)) { P.inputs.assign(1, argv[i + 1]); i += 2; }
        else if ((par == "-o" || par == "--out") && has(1)) { P.out = argv[i + 1]; i += 2; }
        else if (par == "--out-ids" && has(1)) { P.out_ids = argv[i + 1]; i += 2; }
        else if (par == "--out-alignment" && has(1)) { P.out_aln = argv[i + 1]; i += 2; }
        else if ((par == "-t" || par == "--threads") && has(1)) { P.threads = (uint32_t)atoi(argv[i + 1]); i += 2; }
        else if ((par == "-s" || par == "--msl") && has(1)) { z.min_seed_len = atoi(argv[i + 1]); i += 2; }
        else if ((par == "-a" || par == "--mal") && has(1)) { z.min_anchor_len = atoi(argv[i + 1]); i += 2; }
        else if ((par == "-r" || par == "--mrd") && has(1)) { z.max_dist_in_ref = atoi(argv[i + 1]); i += 2; }
        else if ((par == "-q" || par == "--mqd") && has(1)) { z.max_dist_in_query = atoi(argv[i + 1]); i += 2; }
        else if ((par == "-g" || par == "--reg") && has(1)) { z.min_region_len = atoi(argv[i + 1]); i += 2; }
        else if (par == "--aw" && has(1)) { z.approx_window = atoi(argv[i + 1]); i += 2; }
        else if (par == "--am" && has(1)) { z.approx_mismatches = atoi(argv[i + 1]); i += 2; }
        else if (par == "--ar" && has(1)) { z.approx_run_len = atoi(argv[i + 1]); i += 2; }
        else if (par == "--flt-kmerdb" && has(2)) { P.filter_fn = argv[i + 1]; P.filter_thr = atof(argv[i + 2]); i += 3; }
        else if ((par == "-V" || par == "--verbose") && has(1)) { P.verbosity = (uint32_t)atoi(argv[i + 1]); i += 2; }
        else if (par == "--out-type" && has(1)) {
            string t = argv[i + 1];
            if (t == "single-txt") P.single_txt = true;
            else if (t == "tsv") P.single_txt = false;
            else { cerr << "Unknown output-type: " << t << endl; usage(); exit(0); }
            i += 2;
        } else if (par == "--out-format" && has(1)) {
            string bad = parse_output_format(argv[i + 1]);
            if (!bad.empty()) { cerr << "Unknown output-format component: " << bad; return false; }
            P.out_format = argv[i + 1];
            i += 2;
        } else if (par == "--out-filter" && has(2)) {
            static const map<string, Comp> flt = {{"tani", Comp::tani}, {"gani", Comp::gani}, {"ani", Comp::ani}, {"qcov", Comp::qcov}, {"rcov", Comp::rcov}};
            auto p = flt.find(argv[i + 1]);
            if (p == flt.end()) { cerr << "Unknown output-filter component: " << argv[i + 1] << " " << argv[i + 2] << endl; return false; }
            P.flt_mask |= 1ull << (uint32_t)p->second;
            P.flt_vals[(int)p->second] = atof(argv[i + 2]);
            i += 3;
        } else if (par == "--multisample-fasta" && has(1)) {
            if (!parse_bool(argv[i + 1], P.multisample)) { cerr << "Unknown value for --multisample-fasta: " << argv[1] << endl; return false; }
            i += 2;
        } else if (par == "--out-in-percent" && has(1)) {
            if (!parse_bool(argv[i + 1], P.in_percent)) { cerr << "Unknown value for --out-in-percent: " << argv[1] << endl; return false; }
            i += 2;
        } else if (par == "--gpus" && has(1)) { P.gpus = max(1, atoi(argv[i + 1])); i += 2; }
        else if (par == "--device" && has(1)) { P.device = atoi(argv[i + 1]); i += 2; }
        else if (par == "--results-out" && has(1)) { P.results_out = argv[i + 1]; i += 2; }
        else if (par == "--results-in" && has(1)) { P.results_in = argv[i + 1]; i += 2; }
        else { cerr << "Unknown parameter: " << argv[i] << endl; usage(); exit(1); }
    }
    if (P.inputs.empty()) { cerr << "Input file names not provided\n"; return false; }
    // The engine's parameter envelope (lz-ani_amd/csrc/lzani_layout.h: params_supported).  The reference accepts any
    // ints here (lz-ani.cpp:205-260); values outside the envelope are refused before any file is read.
    struct { const char* flag; int v, lo, hi; } env[] = {
        {"--msl", z.min_seed_len, 1, 32}, {"--mal", z.min_anchor_len, 1, 32}, {"--mrd", z.max_dist_in_ref, 0, 1 << 20},
        {"--mqd", z.max_dist_in_query, 0, 64}, {"--aw", z.approx_window, 1, 64}, {"--ar", z.approx_run_len, -(1 << 30), 64},
        {"--am", z.approx_mismatches, 0, 1 << 30}};
    for (auto& e : env)
        if (e.v < e.lo || e.v > e.hi) {
            cerr << "Unsupported value: " << e.flag << " " << e.v << " (this engine supports " << e.lo << " .. " << e.hi << ")" << endl;
            exit(1);
        }
    if (P.verbosity >= 2 && (z.min_anchor_len > 15 || z.min_seed_len > 15))
        cerr << "Note: --mal / --msl above 15: the engine runs without per-genome k-mer words (generic path, several times slower)" << endl;
    return true;
}

static string params_dump()                                   // CParams::str(), params.h:120-157
{
    stringstream ss;
    const lzani_params& z = P.lz;
    ss << "[params]" << endl
       << "min_anchor_len             : " << z.min_anchor_len << endl
       << "min_seed_len               : " << z.min_seed_len << endl
       << "max_dist_in_ref            : " << z.max_dist_in_ref << endl
       << "max_dist_in_query          : " << z.max_dist_in_query << endl
       << "min_region_len             : " << z.min_region_len << endl
       << "approx_window              : " << z.approx_window << endl
       << "approx_mismatches          : " << z.approx_mismatches << endl
       << "approx_run_len             : " << z.approx_run_len << endl
       << "multisample_fasta          : " << boolalpha << P.multisample << noboolalpha << endl
       << "filter_thr                 : " << P.filter_thr << endl
       << "output_format              : " << P.out_format << endl
       << "output_in_percent          : " << boolalpha << P.in_percent << noboolalpha << endl
       << "no_threads                 : " << P.threads << endl
       << "output_file_name           : " << P.out << endl
       << "output_ids_file_name       : " << P.out_ids << endl
       << "output_alignment_file_name : " << P.out_ids << endl
       << "filter_file_name           : " << P.filter_fn << endl
       << "input_file_names           : ";
    for (size_t i = 0; i + 1 < P.inputs.size(); ++i) ss << P.inputs[i] << ", ";
    ss << P.inputs.back() << endl;
    return ss.str();
}

// ---- the engine, bound at run time -----------------------------------------------------------
struct Engine {
    void* so = nullptr;
    int (*create)(const lzani_params*, int, lzani_ctx**) = nullptr;
    void (*destroy)(lzani_ctx*) = nullptr;
    const char* (*last_error)(const lzani_ctx*) = nullptr;
    int (*set_genomes)(lzani_ctx*, uint32_t, const uint8_t* const*, const uint32_t*) = nullptr;
    int (*get_timing)(const lzani_ctx*, lzani_timing*) = nullptr;
    int (*run_rows_regions)(lzani_ctx*, uint32_t, const uint32_t*, const uint64_t*, const uint32_t*, lzani_result*, lzani_region*,
                            uint64_t, uint64_t*) = nullptr;
    int (*row_costs)(uint32_t, const uint32_t*, const uint64_t*, const uint32_t*, uint32_t, const uint32_t*, uint64_t*) = nullptr;
    int (*partition_rows)(uint32_t, const uint64_t*, uint32_t, uint32_t*) = nullptr;
    int (*group_create)(const lzani_params*, uint32_t, const int*, lzani_group**) = nullptr;
    void (*group_destroy)(lzani_group*) = nullptr;
    const char* (*group_last_error)(const lzani_group*) = nullptr;
    int (*group_set_genomes)(lzani_group*, uint32_t, const uint8_t* const*, const uint32_t*) = nullptr;
    int (*group_run_rows)(lzani_group*, uint32_t, const uint32_t*, const uint64_t*, const uint32_t*, lzani_result*) = nullptr;
    int (*group_get_timing)(const lzani_group*, uint32_t, lzani_timing*, double*) = nullptr;
    bool load(const char* argv0)
    {
        vector<string> cand;
        if (const char* e = getenv("LZANI_LIB")) cand.push_back(e);
        error_code ec;
        auto self = filesystem::canonical("/proc/self/exe", ec);
        if (!ec) { cand.push_back((self.parent_path() / "liblzani_hip.so").string()); cand.push_back((self.parent_path().parent_path() / "liblzani_hip.so").string()); }
        (void)argv0;
        cand.push_back("liblzani_hip.so");
        for (auto& c : cand) if ((so = dlopen(c.c_str(), RTLD_NOW | RTLD_LOCAL))) break;
        if (!so) { cerr << "Cannot load liblzani_hip.so (the HIP engine; there is no CPU fallback): " << dlerror() << endl; return false; }
#define BIND(f) f = reinterpret_cast<decltype(f)>(dlsym(so, "lzani_" #f)); if (!f) { cerr << "Missing symbol lzani_" #f << endl; return false; }
        BIND(create) BIND(destroy) BIND(last_error) BIND(set_genomes) BIND(get_timing) BIND(run_rows_regions)
        BIND(row_costs) BIND(partition_rows)
        BIND(group_create) BIND(group_destroy) BIND(group_last_error) BIND(group_set_genomes) BIND(group_run_rows) BIND(group_get_timing)
#undef BIND
        return true;
    }
};

struct AlnRegion { uint32_t ref, qry; lzani_region r; };

// store_alignment (lz_matcher.cpp:102-169): one BLAST-tab-like row per region; rows of one pair in
// calc_regions order (length desc, seq_start asc), pairs in (reference, query) order.
static bool store_alignment(const vector<Genome>& g, vector<AlnRegion>& regs)
{
    ofstream o(P.out_aln, ios::binary);
    if (!o.is_open()) { cerr << "Cannot open output file: " << P.out_aln << endl; return false; }
    o << "query\treference\tpident\talnlen\tqstart\tqend\trstart\trend\tnt_match\tnt_mismatch\n";
    sort(regs.begin(), regs.end(), [](const AlnRegion& a, const AlnRegion& b) {
        if (a.ref != b.ref) return a.ref < b.ref;
        if (a.qry != b.qry) return a.qry < b.qry;
        int la = a.r.seq_end - a.r.seq_start, lb = b.r.seq_end - b.r.seq_start;
        if (la != lb) return la > lb;
        return a.r.seq_start < b.r.seq_start;
    });
    string line;
    char num[64];
    for (size_t k = 0; k < regs.size();) {
        size_t k1 = k;
        long mat = 0, lit = 0;
        while (k1 < regs.size() && regs[k1].ref == regs[k].ref && regs[k1].qry == regs[k].qry) { mat += regs[k1].r.num_matches; lit += regs[k1].r.num_mismatches; ++k1; }
        const uint32_t r = regs[k].ref, q = regs[k].qry;
        const int len1 = (int)g[r].codes.size(), len2 = (int)g[q].codes.size();
        bool keep = true;
        if (P.flt_mask != 0) {                                  // the part of --out-filter that applies here (124-137)
            double gani = (double)mat / len2, ani = mat + lit != 0 ? (double)mat / (mat + lit) : 0, qcov = (double)(mat + lit) / len2;
            keep = !(gani < P.flt_vals[(int)Comp::gani]) && !(ani < P.flt_vals[(int)Comp::ani]) && !(qcov < P.flt_vals[(int)Comp::qcov]);
        }
        const int rc_corr = 2 * len1 + 2 * P.lz.max_dist_in_ref + 1;
        for (; k < k1; ++k) {
            if (!keep) continue;
            const lzani_region& x = regs[k].r;
            const int length = x.seq_end - x.seq_start;
            line.clear();
            line += g[q].name; line += '\t'; line += g[r].name; line += '\t';
            line.append(num, real_to_chars(100.0 * x.num_matches / length, num, 6)); line += '\t';
            auto put = [&](long v, char sep) { line += to_string(v); line += sep; };
            put(length, '\t'); put(1 + x.seq_start, '\t'); put(x.seq_end, '\t');
            if (x.ref_start < len1) { put(1 + x.ref_start, '\t'); put(x.ref_end, '\t'); }
            else { put(rc_corr - (1 + x.ref_start), '\t'); put(rc_corr - x.ref_end, '\t'); }
            put(x.num_matches, '\t'); put(x.num_mismatches, '\n');
            o.write(line.data(), (streamsize)line.size());
        }
    }
    return true;
}

// do_matching (lz_matcher.cpp:172-277).  The reference's workers pull reference rows off an atomic counter and
// run one CParser each; here the whole row table goes to the engine in one call: lzani_group_run_rows deals
// the rows over the GPUs of the group (cyclic for dense rows, LPT for filtered ones), runs them, gathers the
// shards over RCCL on the first GPU and copies the results out once, CSR-aligned with T.
static bool do_matching(const Engine& E, const vector<Genome>& g, Filter& flt, PairTable& T, vector<AlnRegion>* aln)
{
    const uint32_t n = (uint32_t)g.size();
    if (P.verbosity >= 1) cerr << "All2all sparse" << endl;
    if (flt.empty()) T.init_dense(n);
    else { T.init_sparse(flt.rows); vector<vector<uint32_t>>().swap(flt.rows); }   // the reference clears filter rows as it goes (266-267)
    vector<const uint8_t*> ptr(n);
    vector<uint32_t> len(n), ref_ids(n);
    for (uint32_t i = 0; i < n; ++i) { ptr[i] = g[i].codes.data(); len[i] = (uint32_t)g[i].codes.size(); ref_ids[i] = i; }
    const uint32_t* qids = T.dense ? nullptr : T.query_ids.data();
    const int ng = max(1, min<int>(P.gpus, (int)max<uint32_t>(n, 1)));

    if (!aln) {
        vector<int> devs(ng);
        for (int d = 0; d < ng; ++d) devs[d] = P.device + d;
        if (const char* e = getenv("LZANI_DEVICE_LIST")) {       // rehearsals: e.g. "0,0,0" runs three shards on one GPU
            devs.clear();
            for (const auto& x : split(e, ',')) devs.push_back(atoi(x.c_str()));
            if (devs.empty()) devs.push_back(P.device);
        }
        lzani_group* grp = nullptr;
        const auto t_a = chrono::steady_clock::now();
        int rc = E.group_create(&P.lz, (uint32_t)devs.size(), devs.data(), &grp);
        if (rc != LZANI_OK) {
            cerr << "LZ matching failed: lzani_group_create failed with code " << rc << ": " << E.group_last_error(nullptr) << endl;
            return false;
        }
        const auto t_b = chrono::steady_clock::now();
        rc = E.group_set_genomes(grp, n, ptr.data(), len.data());
        const auto t_c = chrono::steady_clock::now();
        if (rc == LZANI_OK) rc = E.group_run_rows(grp, n, ref_ids.data(), T.row_off.data(), qids, T.res.data());
        if (P.verbosity >= 2)
            cerr << "engine: create " << chrono::duration<double>(t_b - t_a).count() << " s, genomes to the device " << chrono::duration<double>(t_c - t_b).count()
                 << " s, matching " << chrono::duration<double>(chrono::steady_clock::now() - t_c).count() << " s\n";
        if (rc != LZANI_OK) { cerr << "LZ matching failed: " << E.group_last_error(grp) << endl; E.group_destroy(grp); return false; }
        if (P.verbosity >= 2)
            for (int d = 0; d < (int)devs.size(); ++d) {
                lzani_timing t; double gather = 0;
                if (E.group_get_timing(grp, (uint32_t)d, &t, &gather) == LZANI_OK)
                    cerr << "GPU " << devs[d] << ": " << t.pairs << " pairs, index " << t.index_ms << " ms, k-mer words " << t.kmers_ms
                         << " ms, candidate stage " << t.cand_ms << " ms, pair kernel " << t.pairs_ms << " ms"
                         << (d == 0 && devs.size() > 1 ? ", gather " + to_string(gather) + " ms" : string()) << "\n";
            }
        E.group_destroy(grp);
        return true;
    }

    // --out-alignment: the per-pair region lists are variable-length, so every GPU's shard comes back through host
    // memory (lzani_run_rows_regions per context); the rows are dealt by the same partition as above.
    vector<uint32_t> part(n);
    {
        vector<uint64_t> cost;
        if (!T.dense) { cost.resize(n); E.row_costs(n, ref_ids.data(), T.row_off.data(), qids, n, len.data(), cost.data()); }
        E.partition_rows(n, T.dense ? nullptr : cost.data(), (uint32_t)ng, part.data());
    }
    vector<string> errs(ng);
    mutex mtx;
    auto shard = [&](int d) {
        lzani_ctx* ctx = nullptr;
        int rc = E.create(&P.lz, P.device + d, &ctx);
        if (rc != LZANI_OK) { errs[d] = "lzani_create failed with code " + to_string(rc) + (rc == LZANI_ERR_PARAMS ? " (LZ parameters outside the supported envelope)" : ""); return; }
        rc = E.set_genomes(ctx, n, ptr.data(), len.data());
        vector<uint32_t> rows, query_ids;
        vector<uint64_t> row_off(1, 0);
        for (uint32_t r = 0; r < n; ++r) {
            if (part[r] != (uint32_t)d) continue;
            rows.push_back(r);
            if (!T.dense) query_ids.insert(query_ids.end(), T.query_ids.begin() + (ptrdiff_t)T.row_off[r], T.query_ids.begin() + (ptrdiff_t)T.row_off[r + 1]);
            row_off.push_back(row_off.back() + T.row_size(r));
        }
        vector<lzani_result> out(row_off.back());
        vector<lzani_region> regs;
        uint64_t cap = max<uint64_t>(1024, row_off.back() / 4), cnt = 0;
        while (rc == LZANI_OK) {
            regs.resize(cap);
            rc = E.run_rows_regions(ctx, (uint32_t)rows.size(), rows.data(), row_off.data(), T.dense ? nullptr : query_ids.data(),
                                    out.data(), regs.data(), cap, &cnt);
            if (rc != LZANI_OK || cnt <= cap) break;
            cap = cnt;
        }
        if (rc != LZANI_OK) errs[d] = E.last_error(ctx);
        else {
            regs.resize(cnt);
            for (size_t k = 0; k < rows.size(); ++k)
                copy(out.begin() + (ptrdiff_t)row_off[k], out.begin() + (ptrdiff_t)row_off[k + 1], T.res.begin() + (ptrdiff_t)T.row_off[rows[k]]);
            lock_guard<mutex> lck(mtx);
            for (const auto& x : regs) {
                size_t k = upper_bound(row_off.begin(), row_off.end(), x.pair) - row_off.begin() - 1;
                aln->push_back(AlnRegion{rows[k], T.id_at(rows[k], x.pair - row_off[k]), x});
            }
            if (P.verbosity >= 2) {
                lzani_timing t;
                if (E.get_timing(ctx, &t) == LZANI_OK)
                    cerr << "GPU " << P.device + d << ": " << t.pairs << " pairs, index " << t.index_ms << " ms, k-mer words " << t.kmers_ms
                         << " ms, candidate stage " << t.cand_ms << " ms, pair kernel " << t.pairs_ms << " ms\n";
            }
        }
        E.destroy(ctx);
    };
    vector<thread> th;
    for (int d = 1; d < ng; ++d) th.emplace_back(shard, d);
    shard(0);
    for (auto& t : th) t.join();
    for (auto& e : errs) if (!e.empty()) { cerr << "LZ matching failed: " << e << endl; return false; }
    return true;
}

// Dense all2all, tiled: the matrix goes to the engine in row x column blocks -- block A = the rows of A against the
// queries from A on, plus the rows behind A against the queries of A -- so that after block A every pair {a in A, b > a}
// is there in both directions, which is all the text of the reference rows A needs (store_results, lz_matcher.cpp:
// 371-567, walks the pairs the same way, after ALL of do_matching): an emitter thread formats and writes the rows of A
// while the GPUs work on the next block.  Every directed pair is still computed exactly once; the file is byte for
// byte the one the untiled run writes.
static bool do_matching_tiled(const Engine& E, const vector<Genome>& g, PairTable& T, const EmitParams& ep, uint32_t tile)
{
    const uint32_t n = (uint32_t)g.size();
    if (P.verbosity >= 1) cerr << "All2all sparse" << endl;
    T.init_dense(n);
    vector<const uint8_t*> ptr(n);
    vector<uint32_t> len(n);
    for (uint32_t i = 0; i < n; ++i) { ptr[i] = g[i].codes.data(); len[i] = (uint32_t)g[i].codes.size(); }
    const int ng = max(1, min<int>(P.gpus, (int)max<uint32_t>(n, 1)));
    vector<int> devs(ng);
    for (int d = 0; d < ng; ++d) devs[d] = P.device + d;
    if (const char* e = getenv("LZANI_DEVICE_LIST")) {           // rehearsals: e.g. "0,0,0" runs three shards on one GPU
        devs.clear();
        for (const auto& x : split(e, ',')) devs.push_back(atoi(x.c_str()));
        if (devs.empty()) devs.push_back(P.device);
    }
    lzani_group* grp = nullptr;
    int rc = E.group_create(&P.lz, (uint32_t)devs.size(), devs.data(), &grp);
    if (rc != LZANI_OK) {
        cerr << "LZ matching failed: lzani_group_create failed with code " << rc << ": " << E.group_last_error(nullptr) << endl;
        return false;
    }
    rc = E.group_set_genomes(grp, n, ptr.data(), len.data());
    if (rc != LZANI_OK) { cerr << "LZ matching failed: " << E.group_last_error(grp) << endl; E.group_destroy(grp); return false; }

    ResultWriter W;
    if (!W.open(g, ep)) { E.group_destroy(grp); return false; }
    // emitter: the row blocks in order, as they complete
    mutex mtx;
    condition_variable cv;
    deque<pair<uint32_t, uint32_t>> ready;
    bool done = false;
    thread emitter([&]() {
        for (;;) {
            pair<uint32_t, uint32_t> blk;
            {
                unique_lock<mutex> lk(mtx);
                cv.wait(lk, [&]() { return done || !ready.empty(); });
                if (ready.empty()) return;
                blk = ready.front(); ready.pop_front();
            }
            W.emit_rows(g, T, blk.first, blk.second);
        }
    });
    vector<double> t_index(devs.size(), 0), t_pairs(devs.size(), 0), t_cand(devs.size(), 0);
    vector<uint64_t> n_pairs(devs.size(), 0);
    double t_gather = 0;
    // Three sets of buffers: while the GPUs run block k, one host thread lays out the row lists of block k + 1 and another
    // moves the results of block k - 1 into the table and hands its rows to the emitter.
    struct Block { uint32_t a0 = 0, a1 = 0; vector<uint32_t> rows, q; vector<uint64_t> off; vector<lzani_result> out; };
    Block blk[3];
    const uint32_t n_blocks = (n + tile - 1) / tile;
    auto build = [&](uint32_t k) {
        Block& B = blk[k % 3];
        B.a0 = k * tile; B.a1 = min(n, B.a0 + tile);
        B.rows.clear(); B.q.clear(); B.off.assign(1, 0);
        for (uint32_t r = B.a0; r < n; ++r) {
            const size_t before = B.q.size();
            if (r < B.a1) { for (uint32_t x = B.a0; x < n; ++x) if (x != r) B.q.push_back(x); }
            else for (uint32_t x = B.a0; x < B.a1; ++x) B.q.push_back(x);
            if (B.q.size() == before) continue;
            B.rows.push_back(r);
            B.off.push_back(B.q.size());
        }
        B.out.resize(B.q.size());
    };
    auto post = [&](uint32_t k) {
        Block& B = blk[k % 3];
        for (size_t i = 0; i < B.rows.size(); ++i) {              // into the dense table: row r, query x at x - (x > r)
            const uint32_t r = B.rows[i];
            lzani_result* dst = T.res.data() + T.row_off[r];
            for (uint64_t e = B.off[i]; e < B.off[i + 1]; ++e) { const uint32_t x = B.q[e]; dst[x < r ? x : x - 1] = B.out[e]; }
        }
        { lock_guard<mutex> lk(mtx); ready.emplace_back(B.a0, B.a1); }
        cv.notify_one();
    };
    bool ok = true;
    future<void> f_build, f_post;
    build(0);
    for (uint32_t k = 0; k < n_blocks && ok; ++k) {
        Block& B = blk[k % 3];
        if (k + 1 < n_blocks) f_build = async(launch::async, build, k + 1);
        if (!B.rows.empty()) {
            rc = E.group_run_rows(grp, (uint32_t)B.rows.size(), B.rows.data(), B.off.data(), B.q.data(), B.out.data());
            if (rc != LZANI_OK) { cerr << "LZ matching failed: " << E.group_last_error(grp) << endl; ok = false; }
            for (size_t d = 0; d < devs.size() && ok; ++d) {
                lzani_timing t; double gather = 0;
                if (E.group_get_timing(grp, (uint32_t)d, &t, &gather) == LZANI_OK) {
                    t_index[d] += t.index_ms; t_pairs[d] += t.pairs_ms; t_cand[d] += t.cand_ms + t.kmers_ms; n_pairs[d] += t.pairs;
                    if (d == 0) t_gather += gather;
                }
            }
        }
        if (f_build.valid()) f_build.get();
        if (f_post.valid()) f_post.get();                         // (block k - 1 is out of its buffers before block k + 2 is laid out in them)
        if (ok) f_post = async(launch::async, post, k);
    }
    if (f_post.valid()) f_post.get();
    { lock_guard<mutex> lk(mtx); done = true; }
    cv.notify_one();
    if (P.verbosity >= 2 && ok) {
        cerr << "LZ matching done (" << (n + tile - 1) / tile << " blocks of " << tile << " rows); the last rows are being written" << endl;
        for (size_t d = 0; d < devs.size(); ++d)
            cerr << "GPU " << devs[d] << ": " << n_pairs[d] << " pairs, index " << t_index[d] << " ms, k-mer words + candidate stage " << t_cand[d]
                 << " ms, pair kernel " << t_pairs[d] << " ms" << (d == 0 && devs.size() > 1 ? ", gather " + to_string(t_gather) + " ms" : string()) << "\n";
    }
    emitter.join();
    E.group_destroy(grp);
    return W.close() && ok;
}

// test seams: the matching stage's integers as text, "ref query mat lit comp" per directed pair
static bool write_raw(const string& fn, const PairTable& T)
{
    ofstream o(fn);
    if (!o.is_open()) return false;
    for (uint32_t r = 0; r < T.n; ++r)
        for (uint64_t j = 0; j < T.row_size(r); ++j) {
            const lzani_result& x = T.at(r, j);
            o << r << ' ' << T.id_at(r, j) << ' ' << x.sym_in_matches << ' ' << x.sym_in_literals << ' ' << x.no_components << '\n';
        }
    return true;
}
static bool read_raw(const string& fn, size_t n, PairTable& T)
{
    ifstream in(fn);
    if (!in.is_open()) { cerr << "Cannot open file: " << fn << endl; return false; }
    struct Rec { uint32_t q; lzani_result r; };
    vector<vector<Rec>> rows(n);
    size_t r; Rec x;
    while (in >> r >> x.q >> x.r.sym_in_matches >> x.r.sym_in_literals >> x.r.no_components) { if (r >= n || x.q >= n) return false; rows[r].push_back(x); }
    vector<vector<uint32_t>> ids(n);
    for (size_t k = 0; k < n; ++k) {
        stable_sort(rows[k].begin(), rows[k].end(), [](const Rec& a, const Rec& b) { return a.q < b.q; });
        for (auto& e : rows[k]) ids[k].push_back(e.q);
    }
    T.init_sparse(ids);
    for (size_t k = 0; k < n; ++k) for (size_t j = 0; j < rows[k].size(); ++j) T.res[T.row_off[k] + j] = rows[k][j].r;
    return true;
}

static bool run_all2all(const char* argv0)
{
    using clk = chrono::high_resolution_clock;
    vector<pair<clk::time_point, string>> times{{clk::now(), ""}};
    auto stamp = [&](const char* s) { times.emplace_back(clk::now(), s); };

    if (P.verbosity >= 1) cerr << "Loading sequences\n";
    vector<Genome> g;
    if (P.multisample ? !load_multifasta(P.inputs, g) : !load_fasta(P.inputs, (uint32_t)P.lz.max_dist_in_ref, g)) return false;
    if (P.verbosity >= 2) cerr << g.size() << endl;
    stamp("Loading sequences");

    Filter flt;
    if (!P.filter_fn.empty()) {
        if (P.verbosity >= 1) cerr << "Loading filter data" << endl;
        if (!load_filter(P.filter_fn, P.filter_thr, flt)) return false;
        if (P.verbosity >= 1) cerr << "Filter size: " << flt.size() << endl;
    }
    stamp("Loading filter");

    if (!flt.empty()) {                                     // compare_sequences (lz_matcher.cpp:43-75)
        bool same = flt.names.size() == g.size();
        for (size_t i = 0; same && i < g.size(); ++i) same = flt.names[i] == g[i].name;
        if (!same) {
            cerr << "seq_sn.size(): " << g.size() << endl << "flt_sn.size(): " << flt.names.size() << endl;
            cerr << (flt.names.size() != g.size() ? "Input sequences and filter sequences sets are of different size!"
                                                  : "Input sequences and filter sequences are different!") << endl;
            return false;
        }
    }
    stamp("Comparing sequence and filter compatibility");

    if (P.verbosity >= 1) cerr << "Reordering sequences" << endl;
    auto map = reorder(g);
    if (!flt.empty()) { if (P.verbosity >= 1) cerr << "Reordering filter" << endl; reorder_filter(flt, map); }
    stamp("Reordering sequences");

    EmitParams ep;
    ep.out_name = P.out; ep.ids_name = P.out_ids; ep.single_txt = P.single_txt; ep.in_percent = P.in_percent;
    ep.comps = P.comps; ep.filter_mask = P.flt_mask; memcpy(ep.filter_vals, P.flt_vals, sizeof ep.filter_vals);
    ep.threads = P.threads; ep.mrd = P.lz.max_dist_in_ref;
    if (P.single_txt) ep.params_dump = params_dump();

    // A large dense all2all runs tiled, the result rows of a block written while the next block is matched
    // (LZANI_TILE_ROWS: rows per block, 0 = never; LZANI_TILE_MIN: genomes from which on)
    uint32_t tile = 1024, tile_min = 4096;
    if (const char* e = getenv("LZANI_TILE_ROWS")) tile = (uint32_t)max(0, atoi(e));
    if (const char* e = getenv("LZANI_TILE_MIN")) tile_min = (uint32_t)max(0, atoi(e));
    const bool tiled = tile > 0 && P.results_in.empty() && P.out_aln.empty() && flt.empty() && g.size() >= tile_min && g.size() > tile;

    PairTable results;
    if (!P.results_in.empty()) { if (!read_raw(P.results_in, g.size(), results)) return false; }
    else {
        Engine E;
        if (!E.load(argv0)) return false;
        if (tiled) {
            if (P.verbosity >= 1) cerr << "Storing results (row blocks, while the matching goes on)" << endl;
            if (!do_matching_tiled(E, g, results, ep, tile)) return false;
        } else {
            vector<AlnRegion> aln;
            if (!do_matching(E, g, flt, results, P.out_aln.empty() ? nullptr : &aln)) return false;
            if (!P.out_aln.empty() && !store_alignment(g, aln)) return false;
        }
    }
    stamp(tiled ? "LZ matching + storing results" : "LZ matching");
    if (!P.results_out.empty() && !write_raw(P.results_out, results)) return false;

    if (!tiled) {
        if (P.verbosity >= 1) cerr << "Storing results" << endl;
        if (!store_results(g, results, ep)) return false;
        stamp("Storing results");
    }

    if (P.verbosity > 1) {
        ifstream st("/proc/self/status");
        for (string ln; getline(st, ln);) if (ln.rfind("VmHWM:", 0) == 0) cerr << "Peak RSS " << ln.substr(6) << "\n";
        cerr << "Timings\n";
        for (size_t i = 1; i < times.size(); ++i) cerr << times[i].second << " : " << chrono::duration<double>(times[i].first - times[i - 1].first).count() << "s\n";
        cerr << "Total time: " << chrono::duration<double>(times.back().first - times.front().first).count() << "s\n";
    }
    return true;
}

int main(int argc, char** argv)
{
    P.lz = lzani_params{11, 7, 40, 40, 35, 15, 7, 3};
    parse_output_format("standard");
    if (!parse_params(argc, argv)) return 0;                  // the reference returns 0 here too (lz-ani.cpp:341-342)
    if (argc == 2) return 0;                                  // --version
    if (P.threads == 0) { P.threads = thread::hardware_concurrency(); if (!P.threads) P.threads = 1; }
    if (!run_all2all(argv[0])) { cerr << "Run failed" << endl; exit(1); }
    return 0;
}
