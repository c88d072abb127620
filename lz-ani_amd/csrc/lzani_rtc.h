// lzani_rtc.h -- pair kernels compiled at run time with the context's eight LZ parameters folded in as constants.
// Included by lzani_hip.hip only (host side).
//
// Why: the reference reads its eight ints at run time and has one speed for all of them (lz-ani.cpp:205-260,
// parser.h:31); the hand-written null chain and the constant folding of the pair kernel exist ahead of time for two
// tuples only (the defaults and --mal 15 --msl 9 --reg 60).  Passing the parameters as scalar operands was tried and
// lost to register pressure (docs/history.md, item ag).  So every other tuple gets its own code object: the kernel
// headers are embedded in the library as text (.incbin), compiled by hipRTC with -DLZANI_P_MAL=.. etc. the first time a
// context needs the kernel (2-3 s), and kept on disk keyed by a hash of the source, the options and the gfx target.
// Nothing here is on the path of the two ahead-of-time tuples; if hipRTC is unavailable or the compile fails, the run
// uses the generic ahead-of-time kernel (DEFP = 0) and lzani_get_rtc_info says so.
#pragma once
#include <hip/hiprtc.h>

#include <chrono>
#include <fstream>
#include <sstream>

#include <sys/stat.h>
#include <unistd.h>

#if !defined(__HIP_DEVICE_COMPILE__)
// the kernel headers as text (the build passes -I for csrc/ and include/)
#define LZ_EMBED(sym, file)                                                                        \
    asm(".pushsection .rodata\n.global " #sym "\n" #sym ":\n.incbin \"" file "\"\n.byte 0\n.popsection\n")
LZ_EMBED(lzani_src_api, "lzani.h");
LZ_EMBED(lzani_src_core, "lzani_core.h");
LZ_EMBED(lzani_src_layout, "lzani_layout.h");
LZ_EMBED(lzani_src_tables, "lzani_tables.h");
LZ_EMBED(lzani_src_pairs, "lzani_kernels_pairs.h");
#undef LZ_EMBED
#endif
extern "C" const char lzani_src_api[], lzani_src_core[], lzani_src_layout[], lzani_src_tables[], lzani_src_pairs[];

namespace lzani_rtc {

struct Kernel {
    hipModule_t mod = nullptr;
    hipFunction_t fn = nullptr;
    hipDeviceptr_t guard = nullptr;      // the module's own g_guard_trip
    bool tried = false;
};

struct State {
    Kernel k[2][3];                      // [genomes without N][CAND: 0 probe, 1 join, 2 candidate bitmaps]
    int built = 0, from_cache = 0, failed = 0;
    double compile_ms = 0;
    std::string log;
};

inline bool enabled()
{
    const char* e = getenv("LZANI_RTC");
    return !(e && *e == '0');
}

// local includes and the HIP runtime header go (hipRTC brings its own runtime declarations); <stdint.h> / <type_traits> stay
inline std::string strip_includes(const char* text)
{
    std::istringstream in(text);
    std::string out, line;
    while (std::getline(in, line)) {
        const size_t a = line.find_first_not_of(" \t");
        if (a != std::string::npos && line.compare(a, 8, "#include") == 0 &&
            (line.find('"', a) != std::string::npos || line.find("<hip/") != std::string::npos)) continue;
        if (a != std::string::npos && line.compare(a, 12, "#pragma once") == 0) continue;
        out += line;
        out += '\n';
    }
    return out;
}

inline std::string source_of(const lzani::Params& P, bool nfree, int cand)
{
    char head[512];
    snprintf(head, sizeof head,
             "#define LZANI_RTC 1\n#define LZANI_P_MAL %d\n#define LZANI_P_MSL %d\n#define LZANI_P_MRD %d\n#define LZANI_P_MQD %d\n"
             "#define LZANI_P_REG %d\n#define LZANI_P_AW %d\n#define LZANI_P_AM %d\n#define LZANI_P_AR %d\n"
             "#define LZANI_RTC_NFREE %d\n#define LZANI_RTC_CAND %d\n",
             P.mal, P.msl, P.mrd, P.mqd, P.reg, P.aw, P.am, P.ar, nfree ? 1 : 0, cand);
    std::string s = head;
    s += strip_includes(lzani_src_api);
    s += "__device__ int g_guard_trip = 0;\n";
    s += strip_includes(lzani_src_core);
    s += strip_includes(lzani_src_layout);
    s += strip_includes(lzani_src_tables);
    s += strip_includes(lzani_src_pairs);
    return s;
}

inline unsigned long long fnv1a(const std::string& s, unsigned long long h = 1469598103934665603ULL)
{
    for (unsigned char ch : s) { h ^= ch; h *= 1099511628211ULL; }
    return h;
}

inline std::string cache_dir()
{
    if (const char* e = getenv("LZANI_RTC_CACHE")) return *e ? std::string(e) : std::string();     // empty = no disk cache
    std::string base;
    if (const char* x = getenv("XDG_CACHE_HOME")) base = x;
    else if (const char* h = getenv("HOME")) base = std::string(h) + "/.cache";
    else base = "/tmp/lzani_cache_" + std::to_string((unsigned)getuid());
    return base + "/lzani_rtc";
}

inline bool read_file(const std::string& path, std::vector<char>& out)
{
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) return false;
    const std::streamsize n = f.tellg();
    if (n <= 0) return false;
    out.resize((size_t)n);
    f.seekg(0);
    return (bool)f.read(out.data(), n);
}

inline void write_file_atomic(const std::string& dir, const std::string& path, const std::vector<char>& code)
{
    std::string d;
    for (size_t k = 1; k <= dir.size(); ++k)
        if (k == dir.size() || dir[k] == '/') { d = dir.substr(0, k); (void)mkdir(d.c_str(), 0700); }
    const std::string tmp = path + "." + std::to_string((unsigned)getpid()) + ".tmp";
    {
        std::ofstream f(tmp, std::ios::binary);
        if (!f || !f.write(code.data(), (std::streamsize)code.size())) { (void)unlink(tmp.c_str()); return; }
    }
    if (rename(tmp.c_str(), path.c_str()) != 0) (void)unlink(tmp.c_str());
}

// The kernel of (nfree, cand) for the parameters P on the current device, built or loaded on first use; nullptr if it
// cannot be had (the caller then launches the generic ahead-of-time kernel).
// may_compile = false: only a code object already on disk is taken (a compile costs 2-3 s: it pays from some millions of
// pairs on, lzani_hip.hip decides); the question can be asked again later.
inline Kernel* get(State& st, const lzani::Params& P, bool nfree, int cand, const char* arch, bool may_compile = true)
{
    Kernel& k = st.k[nfree ? 1 : 0][cand];
    if (k.tried) return k.fn ? &k : nullptr;
    const auto t0 = std::chrono::steady_clock::now();
    const std::string src = source_of(P, nfree, cand);
    const std::string archopt = std::string("--offload-arch=") + arch;
    std::vector<const char*> opts = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"};
    int ver_major = 0, ver_minor = 0;
    (void)hiprtcVersion(&ver_major, &ver_minor);
    std::string keytext = src;
    for (const char* o : opts) { keytext += o; keytext += ' '; }
    keytext += std::to_string(ver_major) + "." + std::to_string(ver_minor);
    char name[64];
    snprintf(name, sizeof name, "pairs_%016llx.co", fnv1a(keytext));
    const std::string dir = cache_dir(), path = dir.empty() ? std::string() : dir + "/" + name;
    std::vector<char> code;
    bool cached = !path.empty() && read_file(path, code);
    if (!cached && !may_compile) return nullptr;
    k.tried = true;
    if (!cached) {
        hiprtcProgram prog = nullptr;
        hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "lzani_rtc_pairs.hip", 0, nullptr, nullptr);
        if (r == HIPRTC_SUCCESS) r = hiprtcCompileProgram(prog, (int)opts.size(), opts.data());
        if (r != HIPRTC_SUCCESS) {
            size_t ls = 0;
            st.log = std::string("hipRTC: ") + hiprtcGetErrorString(r);
            if (prog && hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
                std::string lg(ls, '\0');
                if (hiprtcGetProgramLog(prog, &lg[0]) == HIPRTC_SUCCESS) st.log += ": " + lg.substr(0, 2000);
            }
            if (prog) (void)hiprtcDestroyProgram(&prog);
            ++st.failed;
            return nullptr;
        }
        size_t cs = 0;
        if (hiprtcGetCodeSize(prog, &cs) == HIPRTC_SUCCESS && cs) { code.resize(cs); if (hiprtcGetCode(prog, code.data()) != HIPRTC_SUCCESS) code.clear(); }
        (void)hiprtcDestroyProgram(&prog);
        if (code.empty()) { st.log = "hipRTC: no code object"; ++st.failed; return nullptr; }
        if (!path.empty()) write_file_atomic(dir, path, code);
    }
    size_t gbytes = 0;
    if (hipModuleLoadData(&k.mod, code.data()) != hipSuccess || hipModuleGetFunction(&k.fn, k.mod, "lzani_rtc_pairs") != hipSuccess ||
        hipModuleGetGlobal(&k.guard, &gbytes, k.mod, "g_guard_trip") != hipSuccess || gbytes != sizeof(int)) {
        (void)hipGetLastError();
        if (k.mod) (void)hipModuleUnload(k.mod);
        k.mod = nullptr; k.fn = nullptr;
        if (cached && !path.empty()) (void)unlink(path.c_str());       // a stale or damaged entry: built anew next time
        st.log = "hipModuleLoadData: the run-time compiled pair kernel does not load";
        ++st.failed;
        return nullptr;
    }
    ++st.built;
    st.from_cache += cached ? 1 : 0;
    st.compile_ms += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return &k;
}

// compile only (no device needed): the size of the code object, or 0 with the compiler's log in `log`
inline size_t compile_only(const lzani::Params& P, bool nfree, int cand, const char* arch, std::string& log)
{
    const std::string src = source_of(P, nfree, cand);
    const std::string archopt = std::string("--offload-arch=") + arch;
    const char* opts[] = {archopt.c_str(), "-O3", "-std=c++17", "-ffp-contract=off", "-Wno-unused-value"};
    hiprtcProgram prog = nullptr;
    hiprtcResult r = hiprtcCreateProgram(&prog, src.c_str(), "lzani_rtc_pairs.hip", 0, nullptr, nullptr);
    if (r == HIPRTC_SUCCESS) r = hiprtcCompileProgram(prog, 5, opts);
    size_t ls = 0, cs = 0;
    log = r == HIPRTC_SUCCESS ? "" : std::string("hipRTC: ") + hiprtcGetErrorString(r);
    if (prog && hiprtcGetProgramLogSize(prog, &ls) == HIPRTC_SUCCESS && ls > 1) {
        std::string lg(ls, '\0');
        if (hiprtcGetProgramLog(prog, &lg[0]) == HIPRTC_SUCCESS) log += lg;
    }
    if (r == HIPRTC_SUCCESS) (void)hiprtcGetCodeSize(prog, &cs);
    if (prog) (void)hiprtcDestroyProgram(&prog);
    return r == HIPRTC_SUCCESS ? cs : 0;
}

inline void release(State& st)
{
    for (auto& row : st.k)
        for (auto& k : row) {
            if (k.mod) (void)hipModuleUnload(k.mod);
            k = Kernel();
        }
}

}  // namespace lzani_rtc
