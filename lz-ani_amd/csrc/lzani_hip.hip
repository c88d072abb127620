// lzani_hip.hip -- the C-ABI of include/lzani.h: context, device memory, launches.  gfx950 (MI355X) only.
//
// The kernels (all integer / bit work; no MFMA by design) live in two headers included below:
//   lzani_kernels_index.h   k_pack, k_kmers, k_idx_*   genomes -> packed texts, k-mer words, anchor indexes
//   lzani_kernels_cand.h    k_pm_build, k_pm_cand   dense rows: presence matrix of a group of references -> per-pair candidate bitmaps
//   lzani_kernels_pairs.h   DevWave, k_pairs   the pair kernel
// The algorithm itself (PairMachine and its building blocks, shared with the host model of the tests) is
// lzani_core.h; sizes and the parameter envelope are lzani_layout.h.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/lzani.h"

__device__ int g_guard_trip = 0;   // see LZ_GUARD_TRIP in lzani_core.h
#ifdef LZANI_STAMPS
__device__ unsigned long long g_stamp_acc[8];
#endif
#ifdef LZANI_CHAIN_STATS
__device__ unsigned long long g_chain_stats[24];
#endif
#ifdef LZANI_PHASE_TIME
__device__ unsigned long long g_phase_time[4];
#endif
#ifdef LZANI_PATH_STATS
__device__ unsigned long long g_path_stats[36];
#endif

int lzani_sort_segments(const unsigned long long* in, unsigned long long* out, size_t seg_len, size_t n_seg, int begin_bit, int end_bit,
                        void* tmp, size_t* tmp_bytes, hipStream_t stream);
int lzani_sort_keys(const unsigned long long* in, unsigned long long* out, size_t n, int begin_bit, int end_bit,
                    void* tmp, size_t* tmp_bytes, hipStream_t stream);      // lzani_sort.hip (the engine's own radix sort)

#include "lzani_core.h"
#include "lzani_layout.h"
#include "lzani_tables.h"
#include "lzani_kernels_index.h"
#include "lzani_kernels_cand.h"
#include "lzani_kernels_pairs.h"
#include "lzani_kernels_split.h"
#include "lzani_rtc.h"

// ============================================================================================
// Host side of the C-ABI
// ============================================================================================
using namespace lzani;

struct lzani_ctx {
    Params P;
    int dev = 0;
    hipStream_t stream = nullptr;
    std::vector<hipEvent_t> events;    // four per batch of a run (index begin/end, pairs begin/end)
    int n_cus = 256;
    std::string err;

    u32 n = 0, n_pending = 0;     // n_pending: genome count while lzani_set_genomes is still at work
    std::vector<int> L;
    std::vector<u64> nmoff;
    int Tmax = 0;
    IndexGeom geo{};
    u64* d_t2 = nullptr;
    u64* d_nm = nullptr;
    u64* d_nmoff = nullptr;
    int* d_L = nullptr;
    int* d_hasN = nullptr;
    u32* d_kmL = nullptr;     // k-mer arrays (fast path: mal, msl <= 15), 64 entries per nm word
    u32* d_kmS = nullptr;
    u64 total_nm = 0;
    bool kmers_ready = false;
    bool km_timed = false;        // the last run made the k-mer words (ev_km holds their stamps)
    double join_ms_pending = 0;   // ... and / or the join lists: their time, added to that run's kmers_ms
    hipEvent_t ev_km[2] = {nullptr, nullptr};
    bool all_nfree = false;       // no genome holds an N: the NFREE kernel instantiation applies

    u32* d_dirz = nullptr;
    u32* d_ent = nullptr;
    u32* d_bk = nullptr;          // bucket tables
    u64 bk_stride = 0;
    u32* d_status = nullptr;      // per slot: the LDS index build left this slot to the global-atomics kernels
    bool build_attr_set = false;
    u32* d_tw = nullptr;          // tag words of the bucket tables (tag bits <= 7)
    u64 tw_stride = 0;
    u32* d_fl = nullptr;          // presence filters (probe form with tag words), or one all-ones word
    u64 fl_stride = 0;            // words per slot; 0 = no filter (d_fl = the all-ones word)
    u32 fmask = 31;
    u32 slots = 0;
    u32 batches_last_run = 0;
    // join form of candidate detection (long genomes): per-genome k-mer lists sorted by bucket
    bool join_mode = false, join_ready = false;
    unsigned long long* d_jkeys_in = nullptr;     // unsorted keys, koff[g] + p
    unsigned long long* d_jkeys = nullptr;        // sorted
    u64* d_jkoff = nullptr;                       // per genome: offset of its forward positions (n + 1)
    u64* d_jsoff = nullptr;                       // per genome: offset of its sorted valid keys (n + 1)
    u32* d_jcnt = nullptr;
    void* d_jtmp = nullptr;       // radix-sort scratch (join lists and the sort-based index build)
    size_t jtmp_bytes = 0;
    // sort-based index build (large directories): keys of the batch's references, unsorted / sorted, per-slot counts and starts
    bool sort_build = false;
    unsigned long long* d_ikeys_in = nullptr;
    unsigned long long* d_ikeys = nullptr;
    u32* d_icnt = nullptr;
    u64* d_ibase = nullptr;
    std::vector<u64> jkoff;
    u32 max_slots = 65535;        // gridDim.y limit; LZANI_MAX_SLOTS lowers it (tests force the multi-batch path)
    u64 dir_stride = 0, ent_stride = 0;
    unsigned long long* d_cursor = nullptr;
    u32* d_blkctr = nullptr;      // k_pairs_blk: one pair counter per block
    int blk_launches = 0;         // launches of k_pairs_blk in the last run
    int blk_fold = -1;            // k_pairs_blk: LDS filter = global filter folded 2^blk_fold times (-1: not decided yet, -2: does not fit)
    // dense rows: candidates from the presence matrix of a group of references (lzani_kernels_cand.h)
    u32* d_pm = nullptr;          // the matrix of one group: 2^pm_bits rows of PM_GROUP bits
    size_t pm_bytes = 0;
    u32* d_pm_cbits = nullptr;    // candidate bitmaps of a batch's pairs
    size_t pm_cbits_bytes = 0;
    u32* d_pm_pidx = nullptr;     // rows with query lists: pair of (query, slot of the group), query flags + list + count behind it
    size_t pm_pidx_bytes = 0;
    int pm_launches = 0;          // pair-kernel launches of the last run fed by candidate bitmaps
    u32* d_lpt_cnt = nullptr;     // batches of few, long pairs: candidates per pair, then the ticket keys unsorted / sorted
    unsigned long long* d_lpt_keys = nullptr;
    size_t lpt_pairs = 0;
    int lpt_launches = 0;         // pair-kernel launches of the last run that took their tickets longest pair first
    int pmfi_launches = 0;        // presence matrices of the last run made by k_pm_from_index
    int split_launches = 0;       // batches of the last run whose pairs were scanned by several waves each
    u64 split_items = 0;          // ... segments run in all (with the ones run again)
    bool pm_attr_set = false;
    u32 pmfi_attr_set = 0;                // k_pm_from_index<RW>: bit RW = its LDS limit is raised

    lzani_timing tm{};
    // pair kernels compiled at run time for this context's parameters (lzani_rtc.h); none for the two ahead-of-time tuples
    lzani_rtc::State rtc;
    std::string arch;             // the device's gfx target, as hipRTC wants it
    int rtc_launches = 0;         // pair-kernel launches of the last run by a run-time compiled kernel
    u64 pairs_seen = 0;           // directed pairs this context has been asked for so far (a run-time compile must pay)

    void* comm = nullptr;         // ncclComm_t of lzani_comm_init (one process per GPU), lzani_multi.h
    u32 n_ranks = 1, rank = 0;
};

namespace {

bool trace_on()
{
    static int on = -1;
    if (on < 0) { const char* e = getenv("LZANI_TRACE"); on = (e && *e && *e != '0') ? 1 : 0; }
    return on == 1;
}
#define TRACE(...) do { if (trace_on()) { fprintf(stderr, "[lzani] " __VA_ARGS__); fputc('\n', stderr); fflush(stderr); } } while (0)

// Temporary device buffer, released on every exit path of the call that owns it.
template <class T>
struct DevBuf {
    T* p = nullptr;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t count) { return hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T)); }
    operator T*() const { return p; }
};

int fail(lzani_ctx* c, int code, const std::string& msg)
{
    if (c) c->err = msg;
    return code;
}

#define HIPCHK(c, call)                                                                               \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail(c, e_ == hipErrorOutOfMemory ? LZANI_ERR_NOMEM : LZANI_ERR_DEVICE,            \
                        std::string(#call) + ": " + hipGetErrorString(e_));                           \
    } while (0)

void free_genomes(lzani_ctx* c)
{
    hipFree(c->d_t2); hipFree(c->d_nm); hipFree(c->d_nmoff); hipFree(c->d_L); hipFree(c->d_kmL); hipFree(c->d_kmS); hipFree(c->d_hasN);
    hipFree(c->d_jkeys); hipFree(c->d_jkoff); hipFree(c->d_jsoff); hipFree(c->d_jcnt); hipFree(c->d_jtmp);
    c->d_jkeys_in = c->d_jkeys = nullptr; c->d_jkoff = c->d_jsoff = nullptr; c->d_jcnt = nullptr; c->d_jtmp = nullptr; c->jtmp_bytes = 0;
    c->join_mode = c->join_ready = false;
    c->d_hasN = nullptr;
    c->d_t2 = c->d_nm = c->d_nmoff = nullptr; c->d_L = nullptr; c->d_kmL = c->d_kmS = nullptr; c->kmers_ready = false;
    c->n = 0;
}
void free_pm(lzani_ctx* c)
{
    hipFree(c->d_pm); hipFree(c->d_pm_cbits); hipFree(c->d_pm_pidx); hipFree(c->d_lpt_cnt); hipFree(c->d_lpt_keys);
    c->d_pm = c->d_pm_cbits = c->d_pm_pidx = nullptr; c->pm_bytes = c->pm_cbits_bytes = c->pm_pidx_bytes = 0;
    c->d_lpt_cnt = nullptr; c->d_lpt_keys = nullptr; c->lpt_pairs = 0;
}
void free_slabs(lzani_ctx* c)
{
    hipFree(c->d_dirz); hipFree(c->d_ent); hipFree(c->d_bk); hipFree(c->d_tw); hipFree(c->d_fl); hipFree(c->d_status);
    c->d_fl = nullptr;
    hipFree(c->d_ikeys_in); hipFree(c->d_ikeys); hipFree(c->d_icnt); hipFree(c->d_ibase);
    c->d_ikeys_in = c->d_ikeys = nullptr; c->d_icnt = nullptr; c->d_ibase = nullptr;
    c->d_dirz = c->d_ent = c->d_bk = c->d_tw = c->d_status = nullptr; c->slots = 0;
}

// The form of the anchor index (bucket table, tag words) is a property of the genome set and the parameters:
// decided once per lzani_set_genomes, so the strides of the slabs never change under an allocation.
void choose_index_form(lzani_ctx* c)
{
    int tagbits = 0;
    while (tagbits < 32 && ((c->geo.tagmask >> tagbits) & 1u)) ++tagbits;
    const bool exact = c->geo.tagmask == (u32)lowmask(c->geo.kb - c->geo.dirbits);
    const char* e = getenv("LZANI_NO_BUCKETS");                           // experiments / test_index_forms
    const char* mx = getenv("LZANI_BK_MAX_DIRBITS");
    const int max_dirbits = mx ? atoi(mx) : 26;
    // bucket table (+ tag words): wherever the sentinels cannot be real entries; 20 B per bucket more per slot
    c->bk_stride = (c->d_kmL && exact && c->geo.dirbits <= max_dirbits && tagbits + c->geo.posbits <= 30 && !(e && *e == '1'))
                       ? ((u64)4 << c->geo.dirbits) : 0;
    const char* t = getenv("LZANI_NO_TAGWORDS");
    c->tw_stride = (c->bk_stride && tagbits <= 7 && !(t && *t == '1')) ? ((u64)1 << c->geo.dirbits) : 0;
    // Join form of candidate detection: where the tag words of one reference exceed what an L2 holds by far, a random
    // probe per query position costs one HBM line each; the query's k-mer list sorted by bucket turns the probes into
    // two streams (DevWave::join).  Needs the anchor queue (tag words, seed window <= 128) and keys of 64 bits.
    {
        const char* nj = getenv("LZANI_NO_JOIN");
        const char* jm = getenv("LZANI_JOIN_MIN_BYTES");
        const u64 min_bytes = jm ? strtoull(jm, nullptr, 10) : (8ull << 20);
        const int gbits = ceil_log2((u64)c->n_pending + 1);          // the all-ones genome number is the invalid key's
        c->join_mode = c->tw_stride && c->tw_stride * 4 >= min_bytes && c->P.mqd + c->P.mrd <= 128 &&
                       gbits + c->geo.kb + c->geo.posbits <= 64 && !(nj && *nj == '1');
    }
    // Sort-based index build where the directory is beyond the LDS-staged build (2^19 buckets): keys of 64 bits with up
    // to 16 bits of slot number
    {
        const char* sm = getenv("LZANI_SORT_INDEX_MIN_DIRBITS");      // tests: 0 forces it at every size
        const char* ns = getenv("LZANI_NO_SORT_INDEX");
        c->sort_build = c->d_kmL && c->geo.dirbits >= (sm ? atoi(sm) : 20) && c->geo.kb + c->geo.posbits <= 60 && !(ns && *ns == '1');
    }
    // Presence filter in front of the tag-word probes (probe form only; k_pairs_blk keeps the reference's in LDS): ~3 bits
    // per text position, at most 2^18 bits (genomes up to ~128 kbp); beyond, one all-ones word passes everything
    {
        const char* nf = getenv("LZANI_NO_FILTER");
        const char* fx = getenv("LZANI_FILTER_MAX_BITS");
        const int fmax = fx ? atoi(fx) : 18;                          // 2^18 bits = 32 KB of LDS per block of 16 waves
        const int fbits = std::min(ceil_log2((u64)std::max(c->Tmax, 1024)) + 1, fmax);
        const bool on = c->tw_stride && !c->join_mode && ceil_log2((u64)std::max(c->Tmax, 1024)) <= fmax && !(nf && *nf == '1');
        c->fl_stride = on ? ((u64)1 << fbits) / 32 : 0;
        c->blk_fold = -1;
        c->fmask = on ? (u32)((1u << fbits) - 1u) : 31u;
    }
    const char* ms = getenv("LZANI_MAX_SLOTS");
    c->max_slots = ms && atoi(ms) > 0 ? (u32)std::min(65535, atoi(ms)) : 65535u;
    if (c->sort_build)                                  // the slot number shares the 64-bit key with hash and position
        c->max_slots = (u32)std::min<u64>(c->max_slots, (1ull << std::min(16, 64 - c->geo.kb - c->geo.posbits)) - 1);
}

int ensure_slabs(lzani_ctx* c, u32 want_rows)
{
    size_t per_slot = (size_t)4 * (c->dir_stride + c->ent_stride + c->bk_stride + c->tw_stride + c->fl_stride) + (c->sort_build ? (size_t)16 * c->Tmax + 16 : 0);
    size_t free_b = 0, total_b = 0;
    HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
    size_t have = c->slots * per_slot;
    size_t budget = (size_t)((free_b + have) * 0.6);
    u32 slots = (u32)std::min<size_t>(std::min<u32>(want_rows, c->max_slots), std::max<size_t>(1, budget / per_slot));
    if (slots <= c->slots) return LZANI_OK;
    free_slabs(c);
    HIPCHK(c, hipMalloc(&c->d_dirz, (size_t)slots * c->dir_stride * 4));
    HIPCHK(c, hipMalloc(&c->d_ent, (size_t)slots * c->ent_stride * 4));
    if (c->bk_stride) HIPCHK(c, hipMalloc(&c->d_bk, (size_t)slots * c->bk_stride * 4));
    if (c->tw_stride) HIPCHK(c, hipMalloc(&c->d_tw, (size_t)slots * c->tw_stride * 4));
    if (c->tw_stride) {
        HIPCHK(c, hipMalloc(&c->d_fl, std::max<size_t>((size_t)slots * c->fl_stride * 4, 4)));
        if (!c->fl_stride) HIPCHK(c, hipMemset(c->d_fl, 0xFF, 4));
    }
    HIPCHK(c, hipMalloc(&c->d_status, (size_t)slots * 4));
    if (c->sort_build) {
        HIPCHK(c, hipMalloc(&c->d_ikeys_in, (size_t)slots * c->Tmax * 8));
        HIPCHK(c, hipMalloc(&c->d_ikeys, (size_t)slots * c->Tmax * 8));
        HIPCHK(c, hipMalloc(&c->d_icnt, (size_t)slots * 4));
        HIPCHK(c, hipMalloc(&c->d_ibase, (size_t)slots * 8));
    }
    c->slots = slots;
    return LZANI_OK;
}

GenomeTab gtab(const lzani_ctx* c) { return GenomeTab{c->d_t2, c->d_nm, c->d_nmoff, c->d_L, c->d_kmL, c->d_kmS, c->d_hasN}; }

// Join form: the k-mer list of every genome as a query, sorted by (genome, bucket) -- k_join_keys + the radix sort of lzani_sort.hip,
// once per run, behind k_kmers (it is part of the path's work like the k-mer words it is made from).
// the resident part of the join lists (the sorted keys: 8 B per forward position), allocated before the index slabs are
// sized so that those see what is really left
int alloc_join_lists(lzani_ctx* c)
{
    const u32 n = c->n;
    if (c->d_jkeys) return LZANI_OK;                 // (set last: a partial allocation is released below and redone)
    c->jkoff.assign((size_t)n + 1, 0);
    for (u32 g = 0; g < n; ++g) c->jkoff[g + 1] = c->jkoff[g] + (u64)c->L[g];
    hipError_t e = hipMalloc(&c->d_jkoff, ((size_t)n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&c->d_jsoff, ((size_t)n + 1) * 8);
    if (e == hipSuccess) e = hipMalloc(&c->d_jcnt, (size_t)n * 4);
    if (e == hipSuccess) e = hipMalloc(&c->d_jkeys, std::max<u64>(c->jkoff[n], 1) * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(c->d_jkoff, c->jkoff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream);
    if (e != hipSuccess) {
        hipFree(c->d_jkoff); hipFree(c->d_jsoff); hipFree(c->d_jcnt); hipFree(c->d_jkeys);
        c->d_jkoff = c->d_jsoff = nullptr; c->d_jcnt = nullptr; c->d_jkeys = nullptr;
        return fail(c, e == hipErrorOutOfMemory ? LZANI_ERR_NOMEM : LZANI_ERR_DEVICE, std::string("join lists: ") + hipGetErrorString(e));
    }
    return LZANI_OK;
}

int build_join_lists(lzani_ctx* c)
{
    const u32 n = c->n;
    int rc0 = alloc_join_lists(c);
    if (rc0) return rc0;
    // the unsorted keys live for the duration of the sort only (as much again as the lists themselves)
    DevBuf<unsigned long long> keys_in;
    HIPCHK(c, keys_in.alloc(std::max<u64>(c->jkoff[n], 1)));
    c->d_jkeys_in = keys_in.p;
    int Lmax = 0;
    for (u32 g = 0; g < n; ++g) Lmax = std::max(Lmax, c->L[g]);
    // An invalid key is all ones; the sort looks at the bits [posbits, shift_g + gbits) only, so no real genome number may
    // be all ones in gbits bits, or its keys with the all-ones hash would be indistinguishable from the invalid keys of
    // the genomes before it (found by the fuzz at n = 4: genome 3 lost the k-mers of its last bucket)
    const int shift_g = c->geo.kb + c->geo.posbits, gbits = ceil_log2((u64)n + 1);
    HIPCHK(c, hipMemsetAsync(c->d_jcnt, 0, (size_t)n * 4, c->stream));
    for (u32 g0 = 0; g0 < n && Lmax > 0; g0 += 32768) {
        const u32 cnt = std::min<u32>(32768, n - g0);
        GenomeTab G = gtab(c);
        G.nmoff += g0; G.L += g0;
        // (the genome number of the key is global: the kernel adds g0 through the offset tables it is given)
        hipLaunchKernelGGL(k_join_keys, dim3((Lmax + 4095) / 4096, cnt), dim3(256), 0, c->stream, G, c->d_jkoff + g0, c->d_jkeys_in,
                           c->d_jcnt + g0, shift_g, c->geo.posbits, Lmax, g0);
    }
    HIPCHK(c, hipGetLastError());
    std::vector<u32> valid(n);
    HIPCHK(c, hipMemcpyAsync(valid.data(), c->d_jcnt, (size_t)n * 4, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // sort in groups of whole genomes below 2^30 keys; invalid keys (all ones) end up behind the group's valid ones
    std::vector<u64> soff((size_t)n + 1, 0);
    for (u32 g0 = 0; g0 < n;) {
        u32 g1 = g0;
        u64 keys = 0;
        while (g1 < n && (g1 == g0 || keys + (u64)c->L[g1] <= (1ull << 30))) keys += (u64)c->L[g1++];
        if (keys > 0x7FFFFFF0ull) return fail(c, LZANI_ERR_ARG, "join lists: a genome of more than 2^31 positions");
        u64 at = c->jkoff[g0];
        for (u32 g = g0; g < g1; ++g) { soff[g] = at; at += valid[g]; }
        if (g1 == n) soff[n] = at;
        if (keys) {
            size_t need = 0;
            int e = lzani_sort_keys(c->d_jkeys_in + c->jkoff[g0], c->d_jkeys + c->jkoff[g0], keys, c->geo.posbits, shift_g + gbits, nullptr, &need, c->stream);
            if (e != 0) return fail(c, LZANI_ERR_DEVICE, "join lists: radix sort (size query) failed");
            if (need > c->jtmp_bytes) {
                hipFree(c->d_jtmp); c->d_jtmp = nullptr; c->jtmp_bytes = 0;
                HIPCHK(c, hipMalloc(&c->d_jtmp, need));
                c->jtmp_bytes = need;
            }
            need = c->jtmp_bytes;
            e = lzani_sort_keys(c->d_jkeys_in + c->jkoff[g0], c->d_jkeys + c->jkoff[g0], keys, c->geo.posbits, shift_g + gbits, c->d_jtmp, &need, c->stream);
            if (e != 0) return fail(c, LZANI_ERR_DEVICE, "join lists: radix sort failed");
        }
        g0 = g1;
    }
    // (a genome's list ends after its valid keys -- d_jcnt -- not where the next list begins: between two groups sit the
    // invalid keys of the first)
    HIPCHK(c, hipMemcpyAsync(c->d_jsoff, soff.data(), ((size_t)n + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));        // (also: keys_in is released below)
    c->d_jkeys_in = nullptr;
    c->tm.index_launches += 2;
    return LZANI_OK;
}

// Per-genome k-mer words (and, for long genomes, the sorted join lists made from them): once per genome set, by the
// first run after lzani_set_genomes -- before its index slabs are sized, so that the slabs see what the lists and the
// sort's temporaries have left -- and kept for the runs that follow (they depend on the genomes and the parameters
// only).  Timed on their own (lzani_timing.kmers_ms).
int ensure_kmers(lzani_ctx* c)
{
    if (!c->d_kmL || c->kmers_ready) return LZANI_OK;
    HIPCHK(c, hipEventRecord(c->ev_km[0], c->stream));
    for (u32 g0 = 0; g0 < c->n; g0 += 32768) {
        u32 cnt = std::min<u32>(32768, c->n - g0);
        GenomeTab G = gtab(c);
        G.nmoff += g0; G.L += g0;
        hipLaunchKernelGGL(k_kmers, dim3((c->Tmax + 255) / 256, cnt), dim3(256), 0, c->stream,
                           G, c->d_kmL, c->d_kmS, c->P.mal, c->P.msl, c->P.mrd, c->Tmax);
    }
    HIPCHK(c, hipGetLastError());
    c->tm.index_launches += 1;
    HIPCHK(c, hipEventRecord(c->ev_km[1], c->stream));
    c->kmers_ready = true;
    c->km_timed = true;
    return LZANI_OK;
}

// The sorted join lists of a long-genome set (join form of candidate detection): made by the first run that needs them
// -- dense rows take their candidates from the presence matrix instead -- and kept like the k-mer words they are made
// from; their time is part of that run's kmers_ms.
int ensure_join(lzani_ctx* c)
{
    if (!c->join_mode || c->join_ready) return LZANI_OK;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIPCHK(c, hipEventCreate(&e0));
    hipError_t e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, c->stream);
    int rc = e == hipSuccess ? build_join_lists(c) : fail(c, LZANI_ERR_DEVICE, std::string("join lists: ") + hipGetErrorString(e));
    if (rc == LZANI_OK) {
        float ms = 0;
        if (hipEventRecord(e1, c->stream) == hipSuccess && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess)
            c->join_ms_pending = ms;
        c->join_ready = true;
    }
    hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    return rc;
}

// Index build of `rows` references (device list d_ref_ids) into slots 0..rows-1.
// with_tw = false: the sort-based build leaves the tag words out (a batch whose pairs read candidate bitmaps never probes them:
// 8.6 GB less to write per 128 x 5 Mbp references)
int build_indexes(lzani_ctx* c, const u32* d_ref_ids, u32 rows, bool with_filter = true, bool with_tw = true)
{
    IdxArgs ia;
    ia.G = gtab(c);
    ia.ref_ids = d_ref_ids;
    ia.dirz = c->d_dirz; ia.ent = c->d_ent;
    ia.dir_stride = c->dir_stride; ia.ent_stride = c->ent_stride;
    ia.mal = c->P.mal; ia.mrd = c->P.mrd; ia.geo = c->geo; ia.todo = nullptr;
    const u32 nb = 1u << c->geo.dirbits;
    { int rc = ensure_kmers(c); if (rc) return rc; }
    if (c->fl_stride && with_filter) {              // (only the block kernel reads it)
        HIPCHK(c, hipMemsetAsync(c->d_fl, 0, (size_t)rows * c->fl_stride * 4, c->stream));
        hipLaunchKernelGGL(k_idx_filter, dim3((u32)std::min<u64>(((u64)c->Tmax + 255) / 256, 64), rows), dim3(256), 0, c->stream,
                           ia, c->d_fl, c->fl_stride, c->fmask, c->Tmax);
        c->tm.index_launches += 1;
    }
    if (c->sort_build) {
        // keys -> radix sort, every slot a segment of its own (lzani_sort.hip) -> the tables in one streaming pass.  A key is
        // hash || position; a position without a k-mer is all ones and sorts behind the slot's keys by the one bit above the hash.
        const int shift_slot = c->geo.kb + c->geo.posbits;
        const u64 Tm = (u64)c->Tmax;
        const u32 group = 1;
        HIPCHK(c, hipMemsetAsync(c->d_icnt, 0, (size_t)rows * 4, c->stream));
        hipLaunchKernelGGL(k_idx_keys, dim3((u32)((Tm + 4095) / 4096), rows), dim3(256), 0, c->stream, ia, c->d_ikeys_in, c->d_icnt, c->Tmax, shift_slot);
        {
            size_t need = 0;
            int e = lzani_sort_segments(c->d_ikeys_in, c->d_ikeys, Tm, rows, c->geo.posbits, shift_slot + 1, nullptr, &need, c->stream);
            if (e != 0) return fail(c, LZANI_ERR_DEVICE, "index build: radix sort (size query) failed");
            if (need > c->jtmp_bytes) {
                HIPCHK(c, hipStreamSynchronize(c->stream));
                hipFree(c->d_jtmp); c->d_jtmp = nullptr; c->jtmp_bytes = 0;
                HIPCHK(c, hipMalloc(&c->d_jtmp, need));
                c->jtmp_bytes = need;
            }
            need = c->jtmp_bytes;
            e = lzani_sort_segments(c->d_ikeys_in, c->d_ikeys, Tm, rows, c->geo.posbits, shift_slot + 1, c->d_jtmp, &need, c->stream);
            if (e != 0) return fail(c, LZANI_ERR_DEVICE, "index build: radix sort failed");
        }
        hipLaunchKernelGGL(k_idx_base, dim3((rows + 255) / 256), dim3(256), 0, c->stream, c->d_icnt, c->d_ibase, rows, group, Tm);
        hipLaunchKernelGGL(k_idx_from_sorted, dim3((u32)std::min<u64>((Tm + 255) / 256, 8192), rows), dim3(256), 0, c->stream,
                           ia, c->d_ikeys, c->d_icnt, c->d_ibase, c->d_bk, with_tw ? c->d_tw : nullptr, c->bk_stride, c->tw_stride);
        HIPCHK(c, hipGetLastError());
        c->tm.index_launches += 4;
        return LZANI_OK;
    }
    const char* nolds = getenv("LZANI_NO_LDS_INDEX");
    const char* ldsmax = getenv("LZANI_LDS_INDEX_MAX_DIRBITS");
    const bool lds_build = c->d_kmL && c->geo.dirbits <= (ldsmax ? atoi(ldsmax) : 19) && !(nolds && *nolds == '1');
    // blocks per slot of the global-atomics kernels: the whole range when they build every slot, a handful when
    // they only pick up what k_idx_build left (usually nothing)
    const u32 gx_pos = lds_build ? 16u : (u32)((c->Tmax + 255) / 256), gx_bkt = lds_build ? 16u : (nb + 255) / 256;
    dim3 gp(gx_pos, rows);
    if (lds_build) {
        // one block per reference, everything through LDS; a slot that does not fit (status != 0) falls through
        // to the global-atomics kernels below, which skip every other slot
        const size_t lds = (size_t)(IDX_RANGE / 2 + IDX_STAGE) * 4;
        if (!c->build_attr_set) {
            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_idx_build), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            c->build_attr_set = true;
        }
        HIPCHK(c, hipMemsetAsync(c->d_status, 0, (size_t)rows * 4, c->stream));
        hipLaunchKernelGGL(k_idx_build, dim3(rows), dim3(1024), lds, c->stream, ia, c->d_bk, c->d_tw, c->bk_stride, c->tw_stride, c->d_status);
        ia.todo = c->d_status;
        hipLaunchKernelGGL(k_idx_zero, dim3(gx_bkt, rows), dim3(256), 0, c->stream, c->d_dirz, c->dir_stride, nb, ia.todo);
    } else HIPCHK(c, hipMemsetAsync(c->d_dirz, 0, (size_t)rows * c->dir_stride * 4, c->stream));
    hipLaunchKernelGGL(k_idx_count, gp, dim3(256), 0, c->stream, ia, c->Tmax);
    hipLaunchKernelGGL(k_idx_scan, dim3(rows), dim3(1024), 0, c->stream, c->d_dirz, c->dir_stride, nb, ia.todo);
    hipLaunchKernelGGL(k_idx_fill, gp, dim3(256), 0, c->stream, ia, c->Tmax);
    hipLaunchKernelGGL(k_idx_sort, dim3(gx_bkt, rows), dim3(256), 0, c->stream,
                       c->d_dirz, c->d_ent, c->dir_stride, c->ent_stride, nb, ia.todo, 0);
    if (c->d_bk)
        hipLaunchKernelGGL(k_idx_buckets, dim3(gx_bkt, rows), dim3(256), 0, c->stream,
                           c->d_dirz, c->d_ent, c->d_bk, c->d_tw, c->dir_stride, c->ent_stride, c->bk_stride, c->tw_stride,
                           nb, c->geo.posbits, ia.todo);
    HIPCHK(c, hipGetLastError());
    c->tm.index_launches += 4;
    return LZANI_OK;
}

struct RegionSink { lzani_region* d_regions; unsigned long long* d_count; unsigned long long capacity; };

int run_rows_impl(lzani_ctx* c, u32 n_rows, const u32* ref_ids, const u64* row_off, const u32* query_ids,
                  int* d_out, const RegionSink* rs = nullptr)
{
    if (!c->n) return fail(c, LZANI_ERR_STATE, "lzani_run_rows: no genomes set");
    c->tm = lzani_timing{};
    // (the k-mer words and the join lists are made by the first run after lzani_set_genomes -- inside its timed index
    // stage, reported as kmers_ms -- and kept: they depend on the genome set and the parameters only)
    c->batches_last_run = 0;
    c->blk_launches = 0;
    c->pm_launches = 0;
    c->lpt_launches = 0;
    c->pmfi_launches = 0;
    c->split_launches = 0;
    c->split_items = 0;
    c->rtc_launches = 0;
    if (n_rows == 0) return LZANI_OK;
    const u64 n_pairs = row_off[n_rows];
    for (u32 k = 0; k < n_rows; ++k) {
        if (ref_ids[k] >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: reference id out of range");
        if (row_off[k + 1] < row_off[k]) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: row_off not monotone");
        if (!query_ids && row_off[k + 1] - row_off[k] != (u64)c->n - 1)
            return fail(c, LZANI_ERR_ARG, "lzani_run_rows: dense row must have n-1 queries");
    }
    if (row_off[0] != 0) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: row_off[0] must be 0");
    // (query lists: the same pass tells whether a row names a query twice and how many queries a group of PM_GROUP
    // consecutive rows involves -- what the candidate stage below goes by)
    bool lists_dup = false;
    u64 lists_involved = 0;
    if (query_ids) {
        std::vector<u32> in_row(c->n, 0xFFFFFFFFu), in_group(c->n, 0xFFFFFFFFu);
        for (u32 k = 0; k < n_rows; ++k)
            for (u64 e = row_off[k]; e < row_off[k + 1]; ++e) {
                const u32 q = query_ids[e];
                if (q >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: query id out of range");
                lists_dup |= in_row[q] == k;
                in_row[q] = k;
                if (in_group[q] != k / PM_GROUP) { in_group[q] = k / PM_GROUP; ++lists_involved; }
            }
    }
    if (n_pairs == 0) return LZANI_OK;

    HIPCHK(c, hipSetDevice(c->dev));
    c->km_timed = false;
    int rc = ensure_kmers(c);
    if (rc) return rc;
    // Dense rows: the candidates of every pair of a batch come from the presence matrix of its references
    // (lzani_kernels_cand.h) instead of a probe per query position (viral sizes) or a join of sorted k-mer lists per pair
    // (long genomes); a batch is then also bounded by what the candidate bitmaps of its pairs take, and the index slabs
    // are sized for such a batch.
    bool pm = false;
    u64 cb_words = 0;                                        // 32-bit words of one pair's candidate bitmap
    u32 pm_tiles = 0, pm_group = PM_GROUP;
    // rows of the presence matrix: one per k-mer (exact: the mixer is a bijection on the key bits) where the genomes fill a fair
    // part of the key space, else the hash's top bits -- 2^9 rows per text position keep the false candidates below 0.2 % of the
    // query positions, and a group's matrix is cleared and built in proportion to the genomes, not to 4^mal
    const int pm_bits = std::min(std::min(c->geo.kb, 30), ceil_log2((u64)std::max(c->Tmax, 1)) + 9);
    bool use_join = false;
    std::vector<u32> bstart;
    // Two attempts: the candidate-bitmap form first where it applies; if its buffers (matrix, pair table, bitmaps) cannot be
    // had after all -- the sizing below is an estimate, and hipMalloc may fail on a fragmented heap -- they are released
    // and the run falls back to the probe / join form, which needs none of them.
    for (int attempt = 0; attempt < 2; ++attempt) {
    pm = false;
    u32 want_rows = n_rows;
    u64 pm_cap_pairs = 0;                                    // pairs whose bitmaps a batch may hold
    bool pm_nomem = false;
    if (attempt == 0) {
        const char* e = getenv("LZANI_PM");
        const char* mn = getenv("LZANI_PM_MIN_ROWS");
        const char* mb = getenv("LZANI_PM_MAX_BYTES");
        // (from 32 rows on where the probe form with tag words is the alternative; from 8 rows where it is the rounds of the
        // first kernel: genomes whose tags do not fit a tag byte -- 260 kbp to 2 Mbp at mal 15, viral sizes at mal 13+.
        // Below, the matrix -- 16 GB to clear at 30 key bits -- costs more than it saves.)
        // Long genomes (the join is the alternative: 210 ms for the 56 pairs of 8 x 5 Mbp against 149 by bitmaps, 92 with the
        // pairs cut into segments): from two rows on.
        const u32 min_rows = mn ? (u32)std::max(1, atoi(mn)) : c->join_mode ? 2u : c->tw_stride ? 32u : 8u;
        // Query lists qualify when they are dense where they are: a query that occurs in a group of rows should meet a
        // good part of it (one matrix row read serves all its pairs of the group) -- the row x column blocks of a tiled
        // all2all do, the few relatives a kmer-db filter leaves per row do not.  No query twice in a row (one bitmap each).
        // (Measured in round 4 on the related workload, families of 50 in length order -- 16 pairs per query and group:
        // the pair kernel gains 18 % from the bitmaps, the candidate stage costs more than that; against the ROUNDS of the
        // first kernel -- no tag words: long k-mers on mid-size genomes -- the bitmaps win from two pairs per query on.)
        const char* sh = getenv("LZANI_PM_MIN_SHARE");
        const u64 min_share = sh ? strtoull(sh, nullptr, 10) : c->tw_stride ? 48 : 2;
        const bool lists_ok = !query_ids || (!lists_dup && n_pairs >= min_share * lists_involved);
        pm = !rs && lists_ok && c->d_kmL && c->bk_stride && c->P.mqd + c->P.mrd <= 128 && c->geo.kb <= 30 &&
             c->n >= 2 && n_rows >= min_rows && !(e && *e == '0');
        if (pm) {
            int Lmax = 0;
            u64 max_row = 0;
            for (u32 g = 0; g < c->n; ++g) Lmax = std::max(Lmax, c->L[g]);
            for (u32 k = 0; k < n_rows; ++k) max_row = std::max<u64>(max_row, row_off[k + 1] - row_off[k]);
            pm_tiles = (u32)(((u64)Lmax + c->P.mrd + 320 + PM_TILE - 1) / PM_TILE);
            cb_words = (u64)pm_tiles * PM_TILE_WORDS;
            pm_group = pm_bits <= 27 ? (u32)PM_GROUP : 128u;                      // 64-byte rows up to 2^27 of them (8 GB), 16-byte rows beyond (16 GB at 2^30)
            const size_t m_bytes = ((size_t)1 << pm_bits) * (pm_group / 8);
            const size_t x_bytes = query_ids ? ((size_t)c->n * pm_group + 2 * (size_t)c->n + 64) * 4 : 0;   // pair table, query flags, list, count
            const size_t per_pair = (size_t)cb_words * 4;
            const double avg_row = (double)n_pairs / n_rows;
            const size_t per_slot = (size_t)4 * (c->dir_stride + c->ent_stride + c->bk_stride + c->tw_stride + c->fl_stride) + (c->sort_build ? (size_t)16 * c->Tmax + 16 : 0);
            size_t free_b = 0, total_b = 0;
            HIPCHK(c, hipMemGetInfo(&free_b, &total_b));
            // what this run may lay out anew: the free memory and what the context holds from earlier runs -- ITS slabs
            // included, which is why slabs larger than this run wants are released below (ensure_slabs never shrinks them:
            // an earlier run with sparse rows may have grown them to 60 % of the memory)
            const size_t pool = free_b + (size_t)c->slots * per_slot + c->pm_cbits_bytes + c->pm_bytes + c->pm_pidx_bytes;
            const size_t cap = mb ? (size_t)strtoull(mb, nullptr, 10) : std::min((size_t)64 << 30, total_b / 4);
            const double room = pool * 0.85 - (double)m_bytes - (double)x_bytes;
            u64 fit = room > 0 ? (u64)(room / ((double)per_slot + avg_row * (double)per_pair)) : 0;     // rows: a slab + its pairs' bitmaps each
            fit = std::min<u64>(fit, std::min<u32>(n_rows, c->max_slots));
            pm_cap_pairs = std::min<u64>((u64)(cap / per_pair), 0xFFFFFFF0ull);                          // pair indexes of a batch are 32 bits
            pm_cap_pairs = std::min<u64>(pm_cap_pairs, (u64)((double)fit * avg_row) + max_row);
            if (fit < std::min<u32>(8, n_rows) || pm_cap_pairs < max_row) pm = false;    // (a genome set this large: the probe / join form, batch by batch)
            else {
                want_rows = (u32)fit;
                if (c->slots > want_rows) free_slabs(c);                         // (counted as available above)
                if (c->pm_bytes < m_bytes) {
                    hipFree(c->d_pm); c->d_pm = nullptr; c->pm_bytes = 0;
                    if (hipMalloc(&c->d_pm, m_bytes) != hipSuccess) { (void)hipGetLastError(); c->d_pm = nullptr; pm_nomem = true; }
                    else c->pm_bytes = m_bytes;
                }
                if (!pm_nomem && c->pm_pidx_bytes < x_bytes) {
                    hipFree(c->d_pm_pidx); c->d_pm_pidx = nullptr; c->pm_pidx_bytes = 0;
                    if (hipMalloc(&c->d_pm_pidx, x_bytes) != hipSuccess) { (void)hipGetLastError(); c->d_pm_pidx = nullptr; pm_nomem = true; }
                    else c->pm_pidx_bytes = x_bytes;
                }
            }
        }
    }
    if (pm_nomem) { TRACE("candidate bitmaps: no memory for the matrix / pair table, falling back"); free_pm(c); continue; }
    use_join = c->join_mode && !pm;
    if (use_join) { rc = ensure_join(c); if (rc) return rc; }          // (before the slabs are sized: they take 60 % of what is left)
    rc = ensure_slabs(c, want_rows);
    if (rc) {
        if (pm && rc == LZANI_ERR_NOMEM) { free_pm(c); continue; }
        return rc;
    }
    // Batches: as many consecutive rows as there are index slabs -- and, with candidate bitmaps, as their pairs' bitmaps
    // may take.
    bstart.assign(1, 0);
    {
        u32 rows = 0, rows_cap = c->slots;
        u64 pairs = 0, most = 0;
        if (pm && !query_ids) {                              // dense rows: whole groups of references, if there are several batches
            u64 r = std::min<u64>(rows_cap, pm_cap_pairs / (u64)(c->n - 1));
            if (r < n_rows && r > pm_group) r -= r % pm_group;
            rows_cap = (u32)std::max<u64>(r, 1);
        }
        for (u32 k = 0; k < n_rows; ++k) {
            const u64 len = row_off[k + 1] - row_off[k];
            if (rows && (rows == rows_cap || (pm && pairs + len > pm_cap_pairs))) { bstart.push_back(k); most = std::max(most, pairs); rows = 0; pairs = 0; }
            ++rows; pairs += len;
        }
        bstart.push_back(n_rows);
        most = std::max(most, pairs);
        if (pm) {
            size_t need = (size_t)most * cb_words * 4;
            if (const char* fe = getenv("LZANI_PM_FAIL_CBITS")) if (*fe == '1') need = (size_t)1 << 60;     // tests: the fallback below
            if (c->pm_cbits_bytes < need) {
                hipFree(c->d_pm_cbits); c->d_pm_cbits = nullptr; c->pm_cbits_bytes = 0;
                if (hipMalloc(&c->d_pm_cbits, need) != hipSuccess) {
                    (void)hipGetLastError();
                    c->d_pm_cbits = nullptr;
                    TRACE("candidate bitmaps: no memory for %zu bytes of bitmaps, falling back", need);
                    free_pm(c);
                    continue;
                }
                c->pm_cbits_bytes = need;
            }
        }
    }
    break;
    }
    const u32 bs = c->slots;

    // Batches of `bs` rows (one index slab per row).  Everything the batches need from the host -- row tables
    // and the per-XCD work queues of every batch -- is prepared and uploaded before the first launch, so the
    // batches follow each other on the stream without a host round trip in between.
    const u32 n_batches = (u32)bstart.size() - 1;
    c->batches_last_run = n_batches;
    std::vector<u32> qorder(n_rows);
    std::vector<u64> qcum((size_t)n_rows + n_batches);
    std::vector<u32> qb((size_t)n_batches * (NQUEUES + 1));
    {
        std::vector<u32> by_size;
        std::vector<u32> queue[NQUEUES];
        for (u32 b = 0; b < n_batches; ++b) {
            const u32 k0 = bstart[b], rows = bstart[b + 1] - k0;
            auto rlen = [&](u32 k) { return row_off[k0 + k + 1] - row_off[k0 + k]; };
            // rows -> queues: longest row first onto the least loaded queue (equal rows: round robin)
            by_size.resize(rows);
            for (u32 k = 0; k < rows; ++k) by_size[k] = k;
            std::stable_sort(by_size.begin(), by_size.end(), [&](u32 x, u32 y) { return rlen(x) > rlen(y); });
            u64 load[NQUEUES] = {0};
            for (auto& q : queue) q.clear();
            for (u32 k : by_size) {
                u32 best = 0;
                for (u32 x = 1; x < NQUEUES; ++x) if (load[x] < load[best]) best = x;
                queue[best].push_back(k);
                load[best] += rlen(k);
            }
            u32 at = 0;
            u64 cum = 0;
            u32* qo = qorder.data() + k0;
            u64* qc = qcum.data() + k0 + b;
            qc[0] = 0;
            for (u32 x = 0; x < NQUEUES; ++x) {
                qb[(size_t)b * (NQUEUES + 1) + x] = at;
                for (u32 k : queue[x]) { qo[at] = k; cum += rlen(k); qc[++at] = cum; }
            }
            qb[(size_t)b * (NQUEUES + 1) + NQUEUES] = at;
        }
    }

    DevBuf<u32> d_ref, d_q, d_qorder;
    DevBuf<u64> d_off, d_qcum;
    HIPCHK(c, d_qorder.alloc(n_rows));
    HIPCHK(c, d_qcum.alloc(qcum.size()));
    HIPCHK(c, d_ref.alloc(n_rows));
    HIPCHK(c, d_off.alloc((size_t)n_rows + 1));
    HIPCHK(c, hipMemcpyAsync(d_ref, ref_ids, (size_t)n_rows * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_off, row_off, (size_t)(n_rows + 1) * 8, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_qorder, qorder.data(), qorder.size() * 4, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemcpyAsync(d_qcum, qcum.data(), qcum.size() * 8, hipMemcpyHostToDevice, c->stream));
    if (query_ids) {
        HIPCHK(c, d_q.alloc(n_pairs));
        HIPCHK(c, hipMemcpyAsync(d_q, query_ids, (size_t)n_pairs * 4, hipMemcpyHostToDevice, c->stream));
    }
    enum { EV = 5 };
    while (c->events.size() < (size_t)EV * n_batches) {       // five stamps per batch, read back after the one sync
        hipEvent_t e;
        HIPCHK(c, hipEventCreate(&e));
        c->events.push_back(e);
    }

    u32 blocks_per_cu = 8;                                   // 8 blocks x 4 waves = 8 waves per SIMD
    if (const char* e = getenv("LZANI_BLOCKS_PER_CU")) blocks_per_cu = (u32)std::max(1, std::min(8, atoi(e)));   // occupancy experiments
    const u32 max_blocks = (u32)c->n_cus * blocks_per_cu;
    const Params& q = c->P;
    const bool defp = q.mal == 11 && q.msl == 7 && q.mrd == 40 && q.mqd == 40 && q.reg == 35 && q.aw == 15 && q.am == 7 && q.ar == 3;
    // the long-genome parameters (--mal 15 --msl 9 --reg 60, BASELINE configs[3]): the second set the pair kernel folds into
    // its code, the hand-written null chain included (bitmap and join forms)
    const bool lgp = q.mal == 15 && q.msl == 9 && q.mrd == 40 && q.mqd == 40 && q.reg == 60 && q.aw == 15 && q.am == 7 && q.ar == 3;
    std::vector<char> launched(n_batches, 0);
    std::vector<u32> grp_seen;                               // (query lists + candidate bitmaps) the group a query was last seen in
    u32 grp_stamp = 0;
    const char* const bkenv = getenv("LZANI_BLOCK_KERNEL");
    DevBuf<unsigned long long> d_cbits;                      // join form: one candidate bitmap per resident wave
    u64 cbits_stride = 0;
    if (use_join && !rs) {
        int Lmax = 0;
        for (u32 g = 0; g < c->n; ++g) Lmax = std::max(Lmax, c->L[g]);
        cbits_stride = (u64)((Lmax + c->P.mrd) >> 6) + 8;
        HIPCHK(c, d_cbits.alloc((size_t)max_blocks * 4 * cbits_stride));
    }

    // Any other parameter tuple: the same kernel compiled for it the first time this context needs it (lzani_rtc.h) -- the
    // eight ints folded into the code, the hand-written null chain included where the tuple is inside what the chain is
    // written for (chain_params_ok).  Built (or loaded from the disk cache) here, ahead of the stream's first stamp.
    // A compile takes 2-3 s and the folded kernel saves ~0.13 s per million pairs of 40 kbp: a code object that is not in
    // the disk cache yet is built once the context has been asked for LZANI_RTC_MIN_PAIRS pairs in all (default 2 M: the
    // first such run loses a second or two, every later run and every later process wins).
    lzani_rtc::Kernel* rtc_k = nullptr;
    c->pairs_seen += n_pairs;
    if (!defp && !lgp && !rs && c->d_kmL && c->d_bk && lzani_rtc::enabled()) {
        const int cand = pm ? 2 : (use_join && c->d_tw) ? 1 : c->d_tw ? 0 : -1;
        const char* mp = getenv("LZANI_RTC_MIN_PAIRS");
        const u64 min_pairs = mp ? strtoull(mp, nullptr, 10) : 2000000ull;
        if (cand >= 0) {
            rtc_k = lzani_rtc::get(c->rtc, c->P, c->all_nfree, cand, c->arch.c_str(), c->pairs_seen >= min_pairs);
            if (!rtc_k && c->rtc.failed) TRACE("run-time compile unavailable (%s): the generic kernel runs", c->rtc.log.c_str());
        }
    }

    for (u32 b = 0; b < n_batches; ++b) {
        const u32 k0 = bstart[b], rows = bstart[b + 1] - k0;
        const u64 e0 = row_off[k0], e1 = row_off[k0 + rows];
        hipEvent_t* ev = c->events.data() + (size_t)EV * b;
        TRACE("batch %u rows [%u,%u) pairs [%llu,%llu) slots=%u pm=%d", b, k0, k0 + rows, (unsigned long long)e0, (unsigned long long)e1, bs, (int)pm);
        HIPCHK(c, hipEventRecord(ev[0], c->stream));
        // rows for k_pairs_blk (see below): dense, hundreds of pairs each, probe form with tag words and a filter
        const bool blk_rows = !pm && !rs && c->d_kmL && c->tw_stride && !c->join_mode && c->fl_stride && e1 > e0 && (e1 - e0) / rows >= 128 &&
                              (bkenv ? *bkenv == '1' : query_ids == nullptr);
        rc = build_indexes(c, d_ref + k0, rows, blk_rows, !pm);
        if (rc) return rc;
        HIPCHK(c, hipEventRecord(ev[1], c->stream));
        // Few, long pairs (the batch leaves a wave slot only a few of them): the launch is over when its slowest pair is, so
        // the pairs with the most candidates -- the related ones -- go first (k_pm_cand counts, k_lpt_keys + a sort order)
        // ... and fewer pairs than half the wave slots: every pair by several waves, segment by segment (lzani_kernels_split.h)
        u32 split_S = 0;
        int split_seglen = 0;
        if (pm && !rs && e1 > e0 && c->P.mqd + c->P.mrd <= 128) {
            const char* se = getenv("LZANI_SPLIT");
            const char* sl = getenv("LZANI_SPLIT_SEGLEN");
            const u64 bp = e1 - e0, slots = (u64)max_blocks * 4;
            int Lmax = 0;
            for (u32 g = 0; g < c->n; ++g) Lmax = std::max(Lmax, c->L[g]);
            const int Dmax = Lmax + c->P.mrd;
            // (measured at the end of round 4, 5 Mbp: 56 pairs 6 ms a launch instead of 148, 240 pairs 8 instead of 147, 992 pairs 47
            // instead of 148: from 8 wave slots per pair on)
            // (... measured at 5 Mbp; for shorter queries -- from 256 kbp on -- from 16 wave slots per pair, as the suite has run it)
            const bool on = se ? *se == '1' : (cb_words >= 8192 && (bp * 16 <= slots || (cb_words >= 65536 && bp * 8 <= slots)));
            if (on && bp * 2 <= 0xFFFFFFFFull / 64) {
                const char* sse = getenv("LZANI_SPLIT_S");
                u32 S = (u32)std::min<u64>(sse ? (u64)std::max(2, atoi(sse)) : 64, std::max<u64>(2, slots / bp));      // (8 x 5 Mbp: 67 / 58 / 42 ms a launch with 16 / 32 / 64 a pair)
                int seglen = (Dmax + (int)S - 1) / (int)S;
                if (sl && atoi(sl) > 0) { seglen = atoi(sl); S = (u32)std::min<int>(64, std::max(2, (Dmax + seglen - 1) / seglen)); }
                seglen = std::max(seglen, 512);
                if ((Dmax + seglen - 1) / seglen >= 2) { split_S = std::min<u32>(S, (u32)((Dmax + seglen - 1) / seglen)); split_seglen = seglen; }
            }
        }
        bool lpt = false;
        if (pm && e1 > e0) {
            const char* le = getenv("LZANI_LPT");
            const u64 bp = e1 - e0;
            lpt = !rs && bp >= 2 && bp <= (u64)max_blocks * 4 * 32 && (le ? *le == '1' : cb_words >= 8192);     // (queries from ~256 kbp on)
            if (le && *le == '0') lpt = false;
            if (split_S >= 2) lpt = true;                        // (the split wants the candidate counts: which pairs to cut, which first)
            if (lpt && c->lpt_pairs < bp) {
                hipFree(c->d_lpt_cnt); hipFree(c->d_lpt_keys);
                c->d_lpt_cnt = nullptr; c->d_lpt_keys = nullptr; c->lpt_pairs = 0;
                if (hipMalloc(&c->d_lpt_cnt, (size_t)bp * 4) != hipSuccess || hipMalloc(&c->d_lpt_keys, (size_t)bp * 16) != hipSuccess) {
                    (void)hipGetLastError();
                    hipFree(c->d_lpt_cnt); hipFree(c->d_lpt_keys);
                    c->d_lpt_cnt = nullptr; c->d_lpt_keys = nullptr;
                    lpt = false;                                 // (placement only: the run goes on without it)
                } else c->lpt_pairs = (size_t)bp;
            }
            if (lpt) HIPCHK(c, hipMemsetAsync(c->d_lpt_cnt, 0, (size_t)bp * 4, c->stream));
        }
        if (pm && e1 > e0) {
            // candidate bitmaps of the batch's pairs, group by group of PM_GROUP references
            for (u32 g0 = 0; g0 < rows; g0 += pm_group) {
                PmArgs pg;
                pg.G = gtab(c);
                pg.ref_ids = d_ref + k0; pg.row_off = d_off + k0;
                pg.slot0 = g0; pg.rows = std::min<u32>(pm_group, rows - g0);
                pg.M = c->d_pm; pg.rw = ((pg.rows + 127) / 128) * 4; pg.mmask = (u32)lowmask(pm_bits); pg.rshift = c->geo.kb - pm_bits;
                pg.mal = c->P.mal; pg.mrd = c->P.mrd;
                pg.cbits = c->d_pm_cbits; pg.cb_words = cb_words; pg.e0 = e0; pg.n = c->n; pg.q0 = 0;
                pg.query_ids = d_q.p; pg.pidx = nullptr; pg.qflag = pg.qlist = pg.qcount = nullptr;
                pg.pcount = lpt ? c->d_lpt_cnt : nullptr;
                if (query_ids) {                            // the lists of the group's rows -> pair table + the queries involved
                    const size_t tab = (size_t)c->n * 32 * pg.rw;
                    pg.pidx = c->d_pm_pidx; pg.qflag = c->d_pm_pidx + (size_t)c->n * pm_group; pg.qlist = pg.qflag + c->n; pg.qcount = pg.qlist + c->n;
                    HIPCHK(c, hipMemsetAsync(pg.pidx, 0xFF, tab * 4, c->stream));
                    HIPCHK(c, hipMemsetAsync(pg.qflag, 0, ((size_t)2 * c->n + 1) * 4, c->stream));
                    hipLaunchKernelGGL(k_pm_pairs, dim3(pg.rows), dim3(256), 0, c->stream, pg);
                }
                // the matrix: from the group's indexes, chunk by chunk through LDS (long genomes: no global atomics, no clearing),
                // or by one atomicOr per text position into the cleared matrix
                const int tbits = c->geo.kb - c->geo.dirbits;
                // (chunks of 64 KB: two blocks = 32 waves a CU; with 128 KB chunks, one block a CU, the matrix of 128 x 5 Mbp took 3 ms more)
                const int rcl = std::min(pm_bits, pg.rw <= 4 ? 12 : pg.rw <= 8 ? 11 : 10);
                const char* fie = getenv("LZANI_PM_FROM_INDEX");
                const bool from_index = c->geo.tagmask == (u32)lowmask(tbits) && pm_bits == c->geo.kb && rcl >= tbits &&
                                        (fie ? *fie == '1' : pm_bits > 24);
                if (from_index) {
                    c->pmfi_launches += 1;
                    const size_t fl = ((size_t)pg.rw << rcl) * 4;
                    const dim3 gi(1u << (pm_bits - rcl)), bi(1024);
#define LZ_PM_FI(RW) do { \
                        if (!(c->pmfi_attr_set & (1u << RW))) { \
                            HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pm_from_index<RW>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024)); \
                            c->pmfi_attr_set |= 1u << RW; \
                        } \
                        hipLaunchKernelGGL(k_pm_from_index<RW>, gi, bi, fl, c->stream, pg, c->d_dirz, c->d_ent, c->dir_stride, c->ent_stride, tbits, c->geo.posbits, rcl); \
                    } while (0)
                    switch (pg.rw) {
                    case 4: LZ_PM_FI(4); break;
                    case 8: LZ_PM_FI(8); break;
                    case 12: LZ_PM_FI(12); break;
                    default: LZ_PM_FI(16); break;
                    }
#undef LZ_PM_FI
                } else {
                    HIPCHK(c, hipMemsetAsync(c->d_pm, 0, ((size_t)1 << pm_bits) * pg.rw * 4, c->stream));
                    hipLaunchKernelGGL(k_pm_build, dim3((u32)std::min<u64>(((u64)c->Tmax + 255) / 256, 64), pg.rows), dim3(256), 0, c->stream, pg, c->Tmax);
                }
                const u32 rp = 32 * pg.rw;
                const size_t lds = (size_t)(PM_TILE_WORDS * (rp + 1) + rp) * 4;
                if (!c->pm_attr_set) {
                    const size_t lmax = (size_t)(PM_TILE_WORDS * (PM_GROUP + 1) + PM_GROUP) * 4;
                    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pm_cand<1>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmax));
                    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pm_cand<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmax));
                    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pm_cand<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmax));
                    HIPCHK(c, hipFuncSetAttribute(reinterpret_cast<const void*>(k_pm_cand<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lmax));
                    c->pm_attr_set = true;
                }
                // (query lists: one row of blocks per query that occurs in the group -- counted here, the device list is
                // k_pm_pairs' -- not per genome: 20,000 genomes x 44 tiles of blocks that find nothing to do were most of
                // the candidate stage of a filtered run)
                u32 nq = c->n;
                if (query_ids) {
                    if (grp_seen.size() != c->n) grp_seen.assign(c->n, 0xFFFFFFFFu);
                    const u32 stamp = ++grp_stamp;
                    nq = 0;
                    for (u64 e = row_off[k0 + g0]; e < row_off[k0 + g0 + pg.rows]; ++e)
                        if (grp_seen[query_ids[e]] != stamp) { grp_seen[query_ids[e]] = stamp; ++nq; }
                }
                for (u32 q0 = 0; q0 < nq; q0 += 32768) {       // gridDim.y is limited to 65535
                    pg.q0 = q0;
                    const dim3 gc(pm_tiles, std::min<u32>(32768, nq - q0)), bc(PM_CAND_THREADS);
                    switch (pg.rw / 4) {
                    case 1: hipLaunchKernelGGL(k_pm_cand<1>, gc, bc, lds, c->stream, pg); break;
                    case 2: hipLaunchKernelGGL(k_pm_cand<2>, gc, bc, lds, c->stream, pg); break;
                    case 3: hipLaunchKernelGGL(k_pm_cand<3>, gc, bc, lds, c->stream, pg); break;
                    default: hipLaunchKernelGGL(k_pm_cand<4>, gc, bc, lds, c->stream, pg); break;
                    }
                }
                c->tm.cand_launches += 2;
            }
            if (!lpt || c->d_lpt_cnt == nullptr) split_S = 0;    // (no candidate counts after all -- their buffer could not be had: no split)
            if (lpt && split_S < 2) {                        // the ticket order of the batch's queues
                const u64 bp = e1 - e0;
                QueueBounds qbv;
                for (int x = 0; x <= NQUEUES; ++x) qbv.v[x] = qb[(size_t)b * (NQUEUES + 1) + x];
                hipLaunchKernelGGL(k_lpt_keys, dim3((u32)std::min<u64>((bp + 255) / 256, 4096)), dim3(256), 0, c->stream,
                                   d_qorder + k0, d_qcum + k0 + b, qbv, d_off + k0, e0, c->d_lpt_cnt, c->d_lpt_keys, rows, bp);
                size_t need = 0;
                int e = lzani_sort_keys(c->d_lpt_keys, c->d_lpt_keys + bp, bp, 32, 56, nullptr, &need, c->stream);
                if (e == 0 && need > c->jtmp_bytes) {
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    hipFree(c->d_jtmp); c->d_jtmp = nullptr; c->jtmp_bytes = 0;
                    HIPCHK(c, hipMalloc(&c->d_jtmp, need));
                    c->jtmp_bytes = need;
                }
                need = c->jtmp_bytes;
                if (e == 0) e = lzani_sort_keys(c->d_lpt_keys, c->d_lpt_keys + bp, bp, 32, 56, c->d_jtmp, &need, c->stream);
                if (e != 0) return fail(c, LZANI_ERR_DEVICE, "ticket order: radix sort failed");
            }
            HIPCHK(c, hipGetLastError());
        }
        HIPCHK(c, hipEventRecord(ev[4], c->stream));
        if (e1 > e0) {
            PairArgs pa;
            pa.G = gtab(c);
            pa.P = c->P; pa.geo = c->geo;
            pa.dirz = c->d_dirz; pa.ent = c->d_ent;
            pa.dir_stride = c->dir_stride; pa.ent_stride = c->ent_stride;
            pa.bk = c->d_bk; pa.bk_stride = c->bk_stride;
            pa.tw = c->d_tw; pa.tw_stride = c->tw_stride;
            pa.fl = c->d_fl; pa.fl_stride = c->fl_stride; pa.fmask = c->fmask;
            pa.ref_ids = d_ref + k0; pa.row_off = d_off + k0; pa.query_ids = d_q;
            pa.out = d_out; pa.cursor = c->d_cursor;
            pa.qorder = d_qorder + k0; pa.qcum = d_qcum + k0 + b;
            for (int x = 0; x <= NQUEUES; ++x) pa.qb[x] = qb[(size_t)b * (NQUEUES + 1) + x];
            pa.skeys = cbits_stride ? c->d_jkeys : nullptr; pa.soff = c->d_jsoff; pa.scnt = c->d_jcnt;
            pa.cbits = d_cbits.p; pa.cbits_stride = cbits_stride; pa.cb_e0 = 0;
            if (pm) { pa.cbits = reinterpret_cast<unsigned long long*>(c->d_pm_cbits); pa.cbits_stride = cb_words / 2; pa.cb_e0 = e0; }
            pa.reg_out = rs ? rs->d_regions : nullptr; pa.reg_count = rs ? rs->d_count : nullptr; pa.reg_cap = rs ? rs->capacity : 0;
            pa.torder = (lpt && split_S < 2) ? c->d_lpt_keys + (e1 - e0) : nullptr;
            c->lpt_launches += (lpt && split_S < 2) ? 1 : 0;
            HIPCHK(c, hipMemsetAsync(c->d_cursor, 0, NQUEUES * sizeof(unsigned long long), c->stream));
            const u64 waves = e1 - e0;
            const dim3 gd((u32)std::min<u64>((waves + 3) / 4, max_blocks)), bd(256);
            HIPCHK(c, hipEventRecord(ev[2], c->stream));
#define LZ_PAIRS(F, N, D, A, B) hipLaunchKernelGGL((k_pairs<F, N, D, A, B>), gd, bd, 0, c->stream, pa)
#define LZ_PAIRS_JOIN(N, D) hipLaunchKernelGGL((k_pairs<true, N, D, false, true, 1>), gd, bd, 0, c->stream, pa)
#define LZ_PAIRS_PM(N, D) hipLaunchKernelGGL((k_pairs<true, N, D, false, true, 2>), gd, bd, 0, c->stream, pa)
            const bool fast = c->d_kmL != nullptr, tw = pa.tw != nullptr, nf = c->all_nfree;
            auto rtc_launch = [&]() -> bool {
                if (!rtc_k) return false;
                void* kargs[] = {&pa};
                if (hipModuleLaunchKernel(rtc_k->fn, gd.x, 1, 1, bd.x, 1, 1, 0, c->stream, kargs, nullptr) != hipSuccess) { (void)hipGetLastError(); return false; }
                c->rtc_launches += 1;
                return true;
            };
            // Probe form, dense rows of hundreds of pairs: blocks of 16 waves with the reference's presence filter in LDS
            // (k_pairs_blk).  The rows a kmer-db filter leaves hold related pairs, where most positions pass the filter:
            // BASELINE configs[4] at full size is 6 % slower this way; LZANI_BLOCK_KERNEL=1/0 overrides.
            bool use_blk = blk_rows && fast && tw && !pa.skeys;
            const void* kf = nf ? (defp ? (const void*)k_pairs_blk<true, true> : (const void*)k_pairs_blk<true, false>)
                                : (defp ? (const void*)k_pairs_blk<false, true> : (const void*)k_pairs_blk<false, false>);
            if (use_blk && c->blk_fold == -1) {     // the largest LDS copy of the filter that leaves two blocks per CU
                for (int fold = 0; fold <= 4 && c->blk_fold < 0; ++fold) {
                    const size_t l = (size_t)(BLK_WAVES * SEED_LDS_WORDS + std::max<u64>(c->fl_stride >> fold, 1)) * 4;
                    if (hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l) != hipSuccess) { (void)hipGetLastError(); continue; }
                    int nb = 0;
                    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kf, 64 * BLK_WAVES, l) == hipSuccess && nb >= 2) c->blk_fold = fold;
                }
                (void)hipGetLastError();
                if (c->blk_fold < 0) c->blk_fold = -2;                  // none does: the wave kernel takes these rows too
            }
            if (c->blk_fold < 0) use_blk = false;
            if (use_blk && !c->d_blkctr) HIPCHK(c, hipMalloc(&c->d_blkctr, (size_t)c->n_cus * 2 * 4));
            if (rs) {                                   // alignment output: one generic instantiation per index form
                if (!fast) LZ_PAIRS(false, false, false, true, false);
                else if (tw) LZ_PAIRS(true, false, false, true, true);
                else LZ_PAIRS(true, false, false, true, false);
            } else if (!fast) LZ_PAIRS(false, false, false, false, false);
            else if (pm && split_S >= 2) {               // few, long pairs: several waves a pair (lzani_kernels_split.h)
                c->pm_launches += 1;
                c->split_launches += 1;
                const u32 npb = (u32)(e1 - e0), S = split_S;
                DevBuf<SplitStart> d_cuts;
                DevBuf<SplitOut> d_souts;
                DevBuf<u32> d_work, d_next, d_cnt;
                DevBuf<unsigned char> d_done, d_heavy;
                HIPCHK(c, d_cuts.alloc((size_t)npb * S));
                HIPCHK(c, d_souts.alloc((size_t)npb * S));
                HIPCHK(c, d_work.alloc((size_t)npb * S));
                HIPCHK(c, d_next.alloc((size_t)npb * S));
                HIPCHK(c, d_cnt.alloc(12));
                HIPCHK(c, d_done.alloc(npb));
                HIPCHK(c, hipMemsetAsync(d_cnt.p, 0, 48, c->stream));
                HIPCHK(c, hipMemsetAsync(d_done.p, 0, npb, c->stream));
                HIPCHK(c, hipMemsetAsync(d_cuts.p, 0xFF, (size_t)npb * S * sizeof(SplitStart), c->stream));      // (cut 0 of every pair: no checkpoint)
                SplitArgs sa;
                sa.pa = pa; sa.rows = rows; sa.n_pairs = npb; sa.S = S; sa.seglen = split_seglen;
                sa.cuts = d_cuts.p; sa.outs = d_souts.p; sa.work = d_work.p; sa.work_next = d_next.p; sa.counters = d_cnt.p; sa.done = d_done.p;
                sa.reg = c->P.reg; sa.last_round = 0;
                const int dsel = defp ? 1 : lgp ? 2 : 0;
                auto launch = [&](int mode, u32 items) {
                    const dim3 gs((u32)std::min<u64>(((u64)items + 3) / 4, max_blocks)), bs4(256);
                    sa.n_work = items;
#define LZ_SPLIT(N, D) do { if (mode == 0) hipLaunchKernelGGL((k_split<N, D, 0>), gs, bs4, 0, c->stream, sa); else hipLaunchKernelGGL((k_split<N, D, 1>), gs, bs4, 0, c->stream, sa); } while (0)
                    if (nf) { if (dsel == 1) LZ_SPLIT(true, 1); else if (dsel == 2) LZ_SPLIT(true, 2); else LZ_SPLIT(true, 0); }
                    else { if (dsel == 1) LZ_SPLIT(false, 1); else if (dsel == 2) LZ_SPLIT(false, 2); else LZ_SPLIT(false, 0); }
#undef LZ_SPLIT
                };
                // which pairs to cut: the ones with many anchor candidates (related: a candidate at every other position; a chance
                // pair has one in a hundred and is scanned whole, by its segment 0 with the null chain at work) -- heaviest first
                u32 items = 0;
                {
                    std::vector<u32> cnt(npb);
                    HIPCHK(c, hipMemcpyAsync(cnt.data(), c->d_lpt_cnt, (size_t)npb * 4, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    const char* he = getenv("LZANI_SPLIT_ALL");
                    const char* te = getenv("LZANI_SPLIT_THR");
                    // (every pair by default: at a wave or two per SIMD a chance pair of 5 Mbp takes nearly as long as a related one;
                    // LZANI_SPLIT_ALL=0 cuts the pairs with a candidate at one position in 32 and more only)
                    const u32 thr = (he && *he == '0') ? (u32)(cb_words * 32 / 32) : te ? (u32)strtoul(te, nullptr, 10) : 0u;
                    std::vector<u32> order(npb);
                    for (u32 k = 0; k < npb; ++k) order[k] = k;
                    std::stable_sort(order.begin(), order.end(), [&](u32 x, u32 y) { return cnt[x] > cnt[y]; });
                    std::vector<unsigned char> heavy(npb, 0);
                    std::vector<u32> all;
                    all.reserve((size_t)npb * 2);
                    u32 n_heavy = 0;
                    for (u32 k : order) if (cnt[k] >= thr) { heavy[k] = 1; ++n_heavy; for (u32 sg = 0; sg < S; ++sg) all.push_back(k * S + sg); }
                    for (u32 k : order) if (cnt[k] < thr) all.push_back(k * S);
                    items = (u32)all.size();
                    HIPCHK(c, d_heavy.alloc(npb));
                    HIPCHK(c, hipMemcpyAsync(d_heavy.p, heavy.data(), npb, hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipMemcpyAsync(d_work.p, all.data(), all.size() * 4, hipMemcpyHostToDevice, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));            // (the vectors leave scope)
                    TRACE("split: %u pairs, %u of them cut into %u segments (candidates >= %u)", npb, n_heavy, S, thr);
                }
                sa.heavy = d_heavy.p;
                launch(0, npb * (S - 1));                                  // the checkpoints
                u32* cur = d_work.p; u32* nxt = d_next.p;
                auto t_round = std::chrono::steady_clock::now();
                const int give_up = 6 + (int)S / 4;                      // (a chain of void segments costs a round each: more segments, more rounds allowed)
                for (int round = 0; round < give_up + 4 && items; ++round) {
                    HIPCHK(c, hipMemsetAsync(d_cnt.p, 0, 8, c->stream));     // tickets, next round's items (the finished pairs' count stays)
                    sa.work = cur; sa.work_next = nxt;
                    launch(1, items);
                    sa.last_round = round >= give_up;
                    hipLaunchKernelGGL(k_split_stitch, dim3((npb + 255) / 256), dim3(256), 0, c->stream, sa);
                    u32 cnt[12] = {0};
                    HIPCHK(c, hipMemcpyAsync(cnt, d_cnt.p, 48, hipMemcpyDeviceToHost, c->stream));
                    HIPCHK(c, hipStreamSynchronize(c->stream));
                    c->split_items += items;
                    {
                        const auto t_now = std::chrono::steady_clock::now();
                        TRACE("split: round %d ran %u segments in %.1f ms, %u pairs finished, %u segments to run again (void so far, by cause: look-back cut short %u, kept/dropped %u, dropped/kept %u, floor %u, guess %u, chain %u)",
                              round, items, std::chrono::duration<double, std::milli>(t_now - t_round).count(), cnt[2], cnt[1], cnt[4], cnt[5], cnt[6], cnt[7], cnt[8], cnt[9]);
                        t_round = t_now;
                    }
                    items = cnt[1];
                    std::swap(cur, nxt);
                    if (items == 0 && cnt[2] != npb) return fail(c, LZANI_ERR_DEVICE, "split pairs: the stitch left pairs behind");
                }
                if (items) return fail(c, LZANI_ERR_DEVICE, "split pairs: no end of rounds");
            } else if (pm) {                            // dense rows: candidate bitmaps made ahead (k_pm_cand)
                c->pm_launches += 1;
                if (rtc_launch()) {}
                else if (nf && defp) LZ_PAIRS_PM(true, 1);
                else if (nf && lgp) LZ_PAIRS_PM(true, 2);
                else if (nf) LZ_PAIRS_PM(true, 0);
                else if (defp) LZ_PAIRS_PM(false, 1);
                else if (lgp) LZ_PAIRS_PM(false, 2);
                else LZ_PAIRS_PM(false, 0);
            } else if (tw && pa.skeys) {                // long genomes: candidates by the join
                if (rtc_launch()) {}
                else if (nf && defp) LZ_PAIRS_JOIN(true, 1);
                else if (nf && lgp) LZ_PAIRS_JOIN(true, 2);
                else if (nf) LZ_PAIRS_JOIN(true, 0);
                else if (defp) LZ_PAIRS_JOIN(false, 1);
                else if (lgp) LZ_PAIRS_JOIN(false, 2);
                else LZ_PAIRS_JOIN(false, 0);
            } else if (use_blk) {
                const u32 fw = (u32)std::max<u64>(c->fl_stride >> c->blk_fold, 1);
                const size_t lds = (size_t)(BLK_WAVES * SEED_LDS_WORDS + fw) * 4;
                HIPCHK(c, hipFuncSetAttribute(kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                pa.fmask = c->fmask >> c->blk_fold;
                c->blk_launches += 1;
                const dim3 gb((u32)std::min<u64>((waves + BLK_CHUNK_MIN - 1) / BLK_CHUNK_MIN, (u64)c->n_cus * 2)), bb(64 * BLK_WAVES);
                if (nf && defp) hipLaunchKernelGGL((k_pairs_blk<true, true>), gb, bb, lds, c->stream, pa, fw, (u32)c->blk_fold, c->d_blkctr);
                else if (nf) hipLaunchKernelGGL((k_pairs_blk<true, false>), gb, bb, lds, c->stream, pa, fw, (u32)c->blk_fold, c->d_blkctr);
                else if (defp) hipLaunchKernelGGL((k_pairs_blk<false, true>), gb, bb, lds, c->stream, pa, fw, (u32)c->blk_fold, c->d_blkctr);
                else hipLaunchKernelGGL((k_pairs_blk<false, false>), gb, bb, lds, c->stream, pa, fw, (u32)c->blk_fold, c->d_blkctr);
            } else if (tw) {
                if (rtc_launch()) {}
                else if (nf && defp) LZ_PAIRS(true, true, true, false, true);
                else if (nf) LZ_PAIRS(true, true, false, false, true);
                else if (defp) LZ_PAIRS(true, false, true, false, true);
                else LZ_PAIRS(true, false, false, false, true);
            } else {
                if (nf && defp) LZ_PAIRS(true, true, true, false, false);
                else if (nf) LZ_PAIRS(true, true, false, false, false);
                else if (defp) LZ_PAIRS(true, false, true, false, false);
                else LZ_PAIRS(true, false, false, false, false);
            }
#undef LZ_PAIRS_PM
#undef LZ_PAIRS_JOIN
#undef LZ_PAIRS
            HIPCHK(c, hipGetLastError());
            HIPCHK(c, hipEventRecord(ev[3], c->stream));
            launched[b] = 1;
            c->tm.pair_launches += 1;
        }
        c->tm.pairs += e1 - e0;
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));       // the one host wait of the call
    TRACE("pairs done");
#ifdef LZANI_STAMPS
    {
        unsigned long long acc[8];
        HIPCHK(c, hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_stamp_acc), sizeof acc));
        unsigned long long tot = 0;
        for (int k = 0; k < 8; ++k) tot += acc[k];
        fprintf(stderr, "[lzani stamps] pairs=%llu total_cycles/pair=%.0f shares:", (unsigned long long)n_pairs, (double)tot / (double)n_pairs);
        const char* nm[8] = {"setup", "null_chain", "refill", "event", "find_event", "ext_fwd", "tail", "-"};
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.1f%%", nm[k], 100.0 * (double)acc[k] / (double)tot);
        fprintf(stderr, "\n");
        unsigned long long z[8] = {0};
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_acc), z, sizeof z));
    }
#endif
#ifdef LZANI_PHASE_TIME
    {
        unsigned long long acc[4], z[4] = {0};
        HIPCHK(c, hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_phase_time), sizeof acc));
        if (acc[3])
            fprintf(stderr, "[lzani phase] pairs=%llu, s_memtime ticks per pair: whole pair %.0f, inside the null chain %.0f (%.1f %%), inside refill %.0f (%.1f %%)\n",
                    acc[3], (double)acc[0] / acc[3], (double)acc[1] / acc[3], 100.0 * acc[1] / acc[0], (double)acc[2] / acc[3], 100.0 * acc[2] / acc[0]);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_phase_time), z, sizeof z));
    }
#endif
#ifdef LZANI_PATH_STATS
    {
        unsigned long long acc[36], z[36] = {0};
        HIPCHK(c, hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_path_stats), sizeof acc));
        const char* nm[24] = {"find_event calls", "fe: seed straight from the chain's round", "fe: common call, plain candidate", "fe: light rounds", "fe: light round hit",
                              "fe: light round without a hit", "fe: merge loop hit", "fe: jump to a plain candidate", "stretch calls", "stretch: not applicable",
                              "stretch: no seed step", "stretch: anchor before the seed", "stretch: seed event", "stretch: ... with the masks", "stretch: sum of seed steps",
                              "refills", "chain: nothing", "chain: round done", "chain: event found", "chain: seed event known", "events", "close events", "extension chunks", "stretch chain: events committed"};
        fprintf(stderr, "[lzani paths] pairs=%llu, per pair:", (unsigned long long)n_pairs);
        for (int k = 0; k < 24; ++k) fprintf(stderr, " %s=%.1f;", nm[k], (double)acc[k] / (double)n_pairs);
        const char* wn[12] = {"-", "text end", "no seed step", "several window positions", "seed bounds", "match of 64+", "anchor before the seed", "tag in two slots / overflow",
                              "anchor elsewhere", "extension runs on", "-", "-"};
        fprintf(stderr, "\n[lzani paths] stretch chain exits per pair:");
        for (int k = 1; k < 10; ++k) fprintf(stderr, " %s=%.1f;", wn[k], (double)acc[24 + k] / (double)n_pairs);
        fprintf(stderr, "\n");
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_path_stats), z, sizeof z));
    }
#endif
#ifdef LZANI_CHAIN_STATS
    {
        unsigned long long acc[24], z[24] = {0};
        HIPCHK(c, hipMemcpyFromSymbol(acc, HIP_SYMBOL(g_chain_stats), sizeof acc));
        const char* nm[8] = {"chain_calls", "commits", "exit_nothing", "exit_seed", "exit_not_plain", "exit_event", "events_general", "refills"};
        fprintf(stderr, "[lzani chain] pairs=%llu per pair:", (unsigned long long)n_pairs);
        for (int k = 0; k < 8; ++k) fprintf(stderr, " %s=%.1f", nm[k], (double)acc[k] / (double)n_pairs);
        fprintf(stderr, "\n[lzani chain] wave cycles per pair: after exit_nothing=%.0f after round_done=%.0f after exit_event=%.0f inside the chain=%.0f\n",
                (double)acc[8] / (double)n_pairs, (double)acc[9] / (double)n_pairs, (double)acc[10] / (double)n_pairs, (double)acc[11] / (double)n_pairs);
        fprintf(stderr, "[lzani chain] events found but not null, per pair: close=%.1f region kept or none open=%.1f no forward record=%.1f backward side=%.1f\n",
                (double)acc[12] / (double)n_pairs, (double)acc[13] / (double)n_pairs, (double)acc[14] / (double)n_pairs, (double)acc[15] / (double)n_pairs);
        fprintf(stderr, "[lzani chain] exit_seed by the test that handed the round back, per pair: anchor's own step=%.1f no window position=%.1f several=%.1f text end=%.1f long seed=%.1f other=%.1f; event known, commit left=%.1f\n",
                (double)acc[16] / (double)n_pairs, (double)acc[17] / (double)n_pairs, (double)acc[18] / (double)n_pairs, (double)acc[19] / (double)n_pairs,
                (double)acc[20] / (double)n_pairs, (double)acc[21] / (double)n_pairs, (double)acc[22] / (double)n_pairs);
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_chain_stats), z, sizeof z));
    }
#endif
    int trip = 0;
    HIPCHK(c, hipMemcpyFromSymbol(&trip, HIP_SYMBOL(g_guard_trip), sizeof(int)));
    if (rtc_k && c->rtc_launches) {              // (a code object of its own has a loop guard of its own)
        int t2 = 0, zero = 0;
        HIPCHK(c, hipMemcpyDtoH(&t2, rtc_k->guard, sizeof(int)));
        if (t2) { HIPCHK(c, hipMemcpyHtoD(rtc_k->guard, &zero, sizeof(int))); if (!trip) trip = t2; }
    }
    if (trip) {
        int zero = 0;
        HIPCHK(c, hipMemcpyToSymbol(HIP_SYMBOL(g_guard_trip), &zero, sizeof(int)));
        return fail(c, LZANI_ERR_DEVICE, "pair kernel: loop guard " + std::to_string(trip) + " tripped (corrupt index or text)");
    }
    if (c->km_timed) {
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, c->ev_km[0], c->ev_km[1]));
        c->tm.kmers_ms = ms;
    }
    c->tm.kmers_ms += c->join_ms_pending;
    c->join_ms_pending = 0;
    for (u32 b = 0; b < n_batches; ++b) {
        hipEvent_t* ev = c->events.data() + (size_t)EV * b;
        float ms = 0;
        HIPCHK(c, hipEventElapsedTime(&ms, ev[0], ev[1]));
        c->tm.index_ms += ms;
        HIPCHK(c, hipEventElapsedTime(&ms, ev[1], ev[4]));
        c->tm.cand_ms += ms;
        if (launched[b]) {
            HIPCHK(c, hipEventElapsedTime(&ms, ev[2], ev[3]));
            c->tm.pairs_ms += ms;
        }
    }
    return LZANI_OK;
}

}  // namespace

extern "C" {

static void comm_release(lzani_ctx* c);      // lzani_multi.h

void lzani_default_params(lzani_params* p)
{
    p->min_anchor_len = 11; p->min_seed_len = 7; p->max_dist_in_ref = 40; p->max_dist_in_query = 40;
    p->min_region_len = 35; p->approx_window = 15; p->approx_mismatches = 7; p->approx_run_len = 3;
}

int lzani_create(const lzani_params* p, int device_id, lzani_ctx** out)
{
    if (!p || !out) return LZANI_ERR_ARG;
    *out = nullptr;
    Params P{p->min_anchor_len, p->min_seed_len, p->max_dist_in_ref, p->max_dist_in_query,
             p->min_region_len, p->approx_window, p->approx_mismatches, p->approx_run_len};
    if (!params_supported(P)) return LZANI_ERR_PARAMS;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return LZANI_ERR_DEVICE;
    lzani_ctx* c = new (std::nothrow) lzani_ctx();
    if (!c) return LZANI_ERR_NOMEM;
    c->P = P;
    c->dev = device_id;
    bool ok = hipSetDevice(device_id) == hipSuccess && hipStreamCreate(&c->stream) == hipSuccess &&
              hipMalloc(&c->d_cursor, NQUEUES * sizeof(unsigned long long)) == hipSuccess;
    if (ok) ok = hipEventCreate(&c->ev_km[0]) == hipSuccess && hipEventCreate(&c->ev_km[1]) == hipSuccess;
    if (ok) {
        hipDeviceProp_t prop;
        ok = hipGetDeviceProperties(&prop, device_id) == hipSuccess;
        if (ok) { c->n_cus = prop.multiProcessorCount; c->arch = prop.gcnArchName; }
    }
    if (!ok) { lzani_destroy(c); return LZANI_ERR_DEVICE; }
    *out = c;
    return LZANI_OK;
}

void lzani_destroy(lzani_ctx* c)
{
    if (!c) return;
    hipSetDevice(c->dev);
    comm_release(c);
    free_genomes(c);
    free_slabs(c);
    free_pm(c);
    lzani_rtc::release(c->rtc);
    hipFree(c->d_cursor);
    hipFree(c->d_blkctr);
    for (auto& e : c->events) if (e) hipEventDestroy(e);
    for (auto& e : c->ev_km) if (e) hipEventDestroy(e);
    if (c->stream) hipStreamDestroy(c->stream);
    delete c;
}

const char* lzani_last_error(const lzani_ctx* c) { return c ? c->err.c_str() : "null context"; }

int lzani_set_genomes(lzani_ctx* c, uint32_t n, const uint8_t* const* codes, const uint32_t* len)
{
    if (!c) return LZANI_ERR_ARG;
    if (!n || !codes || !len) return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: empty input");
    HIPCHK(c, hipSetDevice(c->dev));
    free_genomes(c);
    free_slabs(c);
    free_pm(c);
    c->L.resize(n);
    c->nmoff.resize(n);
    std::vector<u64> codeoff(n);
    u64 total_codes = 0, total_nm = 0;
    int Lmax = 0;
    for (u32 g = 0; g < n; ++g) {
        if (len[g] > 0x3FFFFFFFu - 3u * (u32)c->P.mrd)
            return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: sequence too long for 32-bit text positions");
        if (len[g] && !codes[g]) return fail(c, LZANI_ERR_ARG, "lzani_set_genomes: null sequence");
        c->L[g] = (int)len[g];
        Lmax = std::max(Lmax, c->L[g]);
        codeoff[g] = total_codes; total_codes += len[g];
        c->nmoff[g] = total_nm; total_nm += text_wordsN(ref_text_len(c->L[g], c->P.mrd));
    }
    c->Tmax = ref_text_len(Lmax, c->P.mrd);
    c->geo = index_geometry(c->Tmax, c->P.mal);
    c->dir_stride = ((u64)1 << c->geo.dirbits) + 1;
    c->ent_stride = (u64)c->Tmax;

    DevBuf<uint8_t> d_codes;
    DevBuf<u64> d_codeoff;
    HIPCHK(c, d_codes.alloc(total_codes));
    HIPCHK(c, d_codeoff.alloc(n));
    HIPCHK(c, hipMalloc(&c->d_t2, total_nm * 16));
    HIPCHK(c, hipMalloc(&c->d_nm, total_nm * 8));
    HIPCHK(c, hipMalloc(&c->d_nmoff, (size_t)n * 8));
    HIPCHK(c, hipMalloc(&c->d_L, (size_t)n * 4));
    HIPCHK(c, hipMalloc(&c->d_hasN, (size_t)n * 4));
    HIPCHK(c, hipMemset(c->d_hasN, 0, (size_t)n * 4));
    c->total_nm = total_nm;
    if (c->P.mal <= 15 && c->P.msl <= 15) {
        HIPCHK(c, hipMalloc(&c->d_kmL, total_nm * 64 * 4));
        HIPCHK(c, hipMalloc(&c->d_kmS, total_nm * 64 * 4));
    }
    c->n_pending = n;
    choose_index_form(c);
    // The caller's sequences are separate host buffers: they go up through two pinned 64 MB staging buffers, the
    // copy of one overlapping the fill of the other (the 4 GB of config 5 take as long as the PCIe link needs).
    {
        const u64 chunk = 64ull << 20;
        uint8_t* pin[2] = {nullptr, nullptr};
        hipEvent_t done[2] = {nullptr, nullptr};
        hipError_t e = hipSuccess;
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            e = hipHostMalloc((void**)&pin[k], chunk, hipHostMallocDefault);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&done[k], hipEventDisableTiming);
        }
        u64 at = 0;                                               // codes staged so far
        u32 g = 0; u64 goff = 0;                                  // next genome / offset inside it
        for (int k = 0; e == hipSuccess && at < total_codes; k ^= 1) {
            e = hipEventSynchronize(done[k]);                     // the previous copy out of this buffer (no-op the first time)
            u64 fill = 0;
            while (g < n && fill < chunk) {
                const u64 take = std::min<u64>(chunk - fill, (u64)len[g] - goff);
                if (take) memcpy(pin[k] + fill, codes[g] + goff, take);
                fill += take; goff += take;
                if (goff == len[g]) { ++g; goff = 0; }
            }
            if (e == hipSuccess) e = hipMemcpyAsync(d_codes.p + at, pin[k], fill, hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = hipEventRecord(done[k], c->stream);
            at += fill;
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        for (int k = 0; k < 2; ++k) { if (done[k]) hipEventDestroy(done[k]); if (pin[k]) hipHostFree(pin[k]); }
        if (e != hipSuccess) return fail(c, e == hipErrorOutOfMemory ? LZANI_ERR_NOMEM : LZANI_ERR_DEVICE, std::string("staging the sequences: ") + hipGetErrorString(e));
    }
    HIPCHK(c, hipMemcpy(d_codeoff, codeoff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_nmoff, c->nmoff.data(), (size_t)n * 8, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(c->d_L, c->L.data(), (size_t)n * 4, hipMemcpyHostToDevice));
    c->n = n;
    size_t maxblk = text_wordsN(c->Tmax);
    // gridDim.y is limited to 65535: pack in slices of genomes
    for (u32 g0 = 0; g0 < n; g0 += 32768) {
        u32 cnt = std::min<u32>(32768, n - g0);
        hipLaunchKernelGGL(k_pack, dim3((u32)((maxblk + 127) / 128), cnt), dim3(128), 0, c->stream,
                           d_codes.p, d_codeoff.p + g0, c->d_t2, c->d_nm, c->d_nmoff + g0, c->d_L + g0, c->d_hasN + g0, c->P.mrd, cnt);
    }
    HIPCHK(c, hipGetLastError());
    HIPCHK(c, hipStreamSynchronize(c->stream));
    {
        std::vector<int> hn(n);
        HIPCHK(c, hipMemcpy(hn.data(), c->d_hasN, (size_t)n * 4, hipMemcpyDeviceToHost));
        c->all_nfree = std::all_of(hn.begin(), hn.end(), [](int v) { return v == 0; });
    }
    TRACE("set_genomes: n=%u Tmax=%d dirbits=%d posbits=%d tagmask=%x", n, c->Tmax, c->geo.dirbits, c->geo.posbits, c->geo.tagmask);
    return LZANI_OK;
}

int lzani_run_rows_device(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                          const uint32_t* query_ids, void* d_out)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off || (!d_out && n_rows && row_off[n_rows]))
        return fail(c, LZANI_ERR_ARG, "lzani_run_rows_device: null argument");
    return run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out);
}

int lzani_run_rows(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                   const uint32_t* query_ids, lzani_result* out)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: null argument");
    const u64 n_pairs = n_rows ? row_off[n_rows] : 0;
    if (n_pairs && !out) return fail(c, LZANI_ERR_ARG, "lzani_run_rows: null output");
    HIPCHK(c, hipSetDevice(c->dev));
    DevBuf<lzani_result> d_out;
    if (n_pairs) HIPCHK(c, d_out.alloc(n_pairs));
    int rc = run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out.p);
    if (rc == LZANI_OK && n_pairs) {
        hipError_t e = hipMemcpy(out, d_out.p, n_pairs * sizeof(lzani_result), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, LZANI_ERR_DEVICE, std::string("copy results: ") + hipGetErrorString(e));
    }
    return rc;
}

int lzani_run_rows_regions(lzani_ctx* c, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                           const uint32_t* query_ids, lzani_result* out, lzani_region* regions,
                           uint64_t capacity, uint64_t* n_regions)
{
    if (!c) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off || !n_regions || (capacity && !regions))
        return fail(c, LZANI_ERR_ARG, "lzani_run_rows_regions: null argument");
    const u64 n_pairs = n_rows ? row_off[n_rows] : 0;
    if (n_pairs && !out) return fail(c, LZANI_ERR_ARG, "lzani_run_rows_regions: null output");
    *n_regions = 0;
    HIPCHK(c, hipSetDevice(c->dev));
    DevBuf<lzani_result> d_out;
    DevBuf<lzani_region> d_regions;
    DevBuf<unsigned long long> d_count;
    if (n_pairs) HIPCHK(c, d_out.alloc(n_pairs));
    HIPCHK(c, d_regions.alloc(capacity));
    HIPCHK(c, d_count.alloc(1));
    HIPCHK(c, hipMemset(d_count.p, 0, sizeof(unsigned long long)));
    RegionSink rs{d_regions.p, d_count.p, capacity};
    int rc = run_rows_impl(c, n_rows, ref_ids, row_off, query_ids, (int*)d_out.p, &rs);
    if (rc == LZANI_OK) {
        unsigned long long cnt = 0;
        hipError_t e = hipMemcpy(&cnt, rs.d_count, sizeof cnt, hipMemcpyDeviceToHost);
        if (e == hipSuccess && n_pairs) e = hipMemcpy(out, d_out.p, n_pairs * sizeof(lzani_result), hipMemcpyDeviceToHost);
        if (e == hipSuccess && cnt && capacity)
            e = hipMemcpy(regions, rs.d_regions, std::min<uint64_t>(cnt, capacity) * sizeof(lzani_region), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, LZANI_ERR_DEVICE, std::string("copy regions: ") + hipGetErrorString(e));
        *n_regions = cnt;
    }
    return rc;
}

int lzani_get_timing(const lzani_ctx* c, lzani_timing* t)
{
    if (!c || !t) return LZANI_ERR_ARG;
    *t = c->tm;
    return LZANI_OK;
}

int lzani_get_layout(const lzani_ctx* c, lzani_layout_info* o)
{
    if (!c || !o) return LZANI_ERR_ARG;
    o->key_bits = c->geo.kb; o->dir_bits = c->geo.dirbits; o->pos_bits = c->geo.posbits; o->tag_mask = c->geo.tagmask;
    o->kmer_words = c->d_kmL != nullptr;
    o->bucket_table = c->bk_stride != 0; o->tag_words = c->tw_stride != 0;
    o->n_free = c->all_nfree;
    o->slots = c->slots; o->batches_last_run = c->batches_last_run;
    o->bytes_per_slot = 4 * (c->dir_stride + c->ent_stride + c->bk_stride + c->tw_stride + c->fl_stride);
    o->bytes_genomes = c->total_nm * (16 + 8) + (c->d_kmL ? c->total_nm * 64 * 8 : 0);
    o->join_lists = c->join_mode; o->block_launches = c->blk_launches; o->bitmap_launches = c->pm_launches; o->rtc_launches = c->rtc_launches;
    o->lpt_launches = c->lpt_launches; o->matrix_from_index = c->pmfi_launches;
    o->split_launches = c->split_launches; o->split_segments = c->split_items;
    return LZANI_OK;
}

int lzani_get_rtc_info(const lzani_ctx* c, lzani_rtc_info* o)
{
    if (!c || !o) return LZANI_ERR_ARG;
    const Params& q = c->P;
    const bool aot = (q.mal == 11 && q.msl == 7 && q.mrd == 40 && q.mqd == 40 && q.reg == 35 && q.aw == 15 && q.am == 7 && q.ar == 3) ||
                     (q.mal == 15 && q.msl == 9 && q.mrd == 40 && q.mqd == 40 && q.reg == 60 && q.aw == 15 && q.am == 7 && q.ar == 3);
    o->folded_ahead_of_time = aot;
    o->null_chain = chain_params_ok(q);
    o->kernels_built = c->rtc.built; o->kernels_from_cache = c->rtc.from_cache; o->kernels_failed = c->rtc.failed;
    o->reserved_ = 0;
    o->build_ms = c->rtc.compile_ms;
    return LZANI_OK;
}

int64_t lzani_debug_rtc_compile(const lzani_params* p, int nfree, int cand, const char* arch, char* log, uint64_t log_cap)
{
    if (!p || !arch || cand < 0 || cand > 2) return LZANI_ERR_ARG;
    Params P{p->min_anchor_len, p->min_seed_len, p->max_dist_in_ref, p->max_dist_in_query,
             p->min_region_len, p->approx_window, p->approx_mismatches, p->approx_run_len};
    if (!params_supported(P) || P.mal > 15 || P.msl > 15) return LZANI_ERR_PARAMS;
    std::string lg;
    const size_t n = lzani_rtc::compile_only(P, nfree != 0, cand, arch, lg);
    if (log && log_cap) { snprintf(log, (size_t)log_cap, "%s", lg.c_str()); }
    return n ? (int64_t)n : (int64_t)LZANI_ERR_DEVICE;
}

int lzani_debug_get_index(lzani_ctx* c, uint32_t id, uint64_t* t2, uint64_t* nm, uint32_t* dirz,
                          uint32_t* ent, uint32_t* n_ent, uint32_t* geom)
{
    if (!c) return LZANI_ERR_ARG;
    if (!c->n || id >= c->n) return fail(c, LZANI_ERR_ARG, "lzani_debug_get_index: bad id");
    HIPCHK(c, hipSetDevice(c->dev));
    int rc = ensure_slabs(c, 1);
    if (rc) return rc;
    DevBuf<u32> d_ref;
    HIPCHK(c, d_ref.alloc(1));
    HIPCHK(c, hipMemcpy(d_ref.p, &id, 4, hipMemcpyHostToDevice));
    rc = build_indexes(c, d_ref, 1);
    if (rc) return rc;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    int T = ref_text_len(c->L[id], c->P.mrd);
    size_t wn = text_wordsN(T);
    if (nm) HIPCHK(c, hipMemcpy(nm, c->d_nm + c->nmoff[id], wn * 8, hipMemcpyDeviceToHost));
    if (t2) HIPCHK(c, hipMemcpy(t2, c->d_t2 + 2 * c->nmoff[id], wn * 16, hipMemcpyDeviceToHost));
    std::vector<u32> d(c->dir_stride);
    HIPCHK(c, hipMemcpy(d.data(), c->d_dirz, c->dir_stride * 4, hipMemcpyDeviceToHost));
    u32 ne = d[c->dir_stride - 1];
    if (dirz) memcpy(dirz, d.data(), c->dir_stride * 4);
    if (ent && ne) HIPCHK(c, hipMemcpy(ent, c->d_ent, (size_t)ne * 4, hipMemcpyDeviceToHost));
    if (n_ent) *n_ent = ne;
    if (geom) { geom[0] = c->geo.kb; geom[1] = c->geo.dirbits; geom[2] = c->geo.posbits; geom[3] = c->geo.tagmask; }
    return LZANI_OK;
}

// Test hook: the engine's radix sort (lzani_sort.hip) on host keys -- n_seg segments of seg_len keys, each sorted on its own by
// the bits [begin_bit, end_bit), stably.
int lzani_debug_sort_segments(lzani_ctx* c, const uint64_t* keys, uint64_t* out, uint64_t seg_len, uint32_t n_seg, int begin_bit, int end_bit)
{
    if (!c || !keys || !out) return LZANI_ERR_ARG;
    HIPCHK(c, hipSetDevice(c->dev));
    const size_t n = (size_t)seg_len * n_seg;
    if (n == 0) return LZANI_OK;
    DevBuf<unsigned long long> d_in, d_out;
    DevBuf<unsigned char> d_tmp;
    HIPCHK(c, d_in.alloc(n));
    HIPCHK(c, d_out.alloc(n));
    HIPCHK(c, hipMemcpy(d_in.p, keys, n * 8, hipMemcpyHostToDevice));
    size_t need = 0;
    if (lzani_sort_segments(d_in.p, d_out.p, seg_len, n_seg, begin_bit, end_bit, nullptr, &need, c->stream) != 0)
        return fail(c, LZANI_ERR_ARG, "lzani_debug_sort_segments: bad arguments");
    HIPCHK(c, d_tmp.alloc(need));
    if (lzani_sort_segments(d_in.p, d_out.p, seg_len, n_seg, begin_bit, end_bit, d_tmp.p, &need, c->stream) != 0)
        return fail(c, LZANI_ERR_DEVICE, "lzani_debug_sort_segments: sort failed");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipMemcpy(out, d_out.p, n * 8, hipMemcpyDeviceToHost));
    return LZANI_OK;
}

}  // extern "C"

#include "lzani_multi.h"
