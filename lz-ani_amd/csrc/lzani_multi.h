// lzani_multi.h -- the multi-GPU layer of the C-ABI (include/lzani.h, "sharding over GPUs").  Included by
// lzani_hip.hip only, after the single-GPU entry points it builds on.
//
// What shards is the reference's own work unit, the reference ROW (one index build, many queries;
// /root/reference/src/lz_matcher.cpp:196-255): rows are independent, the packed genome set is replicated on
// every GPU (10k x 40 kbp = 0.3 GB of text + 6.4 GB of k-mer words against 288 GB of HBM), and the only
// exchange is ONE gather of the per-pair int32[3] records after the compute -- RCCL over xGMI.
//
//   lzani_partition_rows / lzani_row_costs   rows -> shards: cyclic for equal-cost (dense) rows, greedy
//                                            longest-processing-time for the ragged rows of a kmer-db filter
//   lzani_comm_*                             one process per GPU (torchrun, MPI...): ncclCommInitRank from a
//                                            unique id the caller distributes, ncclAllGather of padded shards
//                                            or grouped ncclSend/ncclRecv of ragged shards to a root
//   lzani_group_*                            one process, n GPUs (the `lz-ani --gpus n` host binary): a context
//                                            per device on its own host thread, ncclCommInitAll, grouped
//                                            ncclSend/ncclRecv to device 0, a scatter kernel into the caller's
//                                            CSR order and ONE device-to-host copy
#pragma once
#include <rccl/rccl.h>

#include <numeric>
#include <thread>

namespace {

#define RCCLCHK(c, call)                                                                              \
    do {                                                                                              \
        ncclResult_t r_ = (call);                                                                     \
        if (r_ != ncclSuccess)                                                                        \
            return fail(c, LZANI_ERR_DEVICE, std::string(#call) + ": " + ncclGetErrorString(r_));     \
    } while (0)

// Rows of a gathered buffer (shard order) -> the caller's CSR order.  One block per row.
__global__ void k_scatter_rows(const int* __restrict__ src, int* __restrict__ dst, const u64* __restrict__ src_off,
                               const u64* __restrict__ dst_off, const u64* __restrict__ count)
{
    const u64 s = 3 * src_off[blockIdx.x], d = 3 * dst_off[blockIdx.x], n = 3 * count[blockIdx.x];
    for (u64 k = threadIdx.x; k < n; k += blockDim.x) dst[d + k] = src[s + k];
}

}  // namespace

struct lzani_group {
    std::vector<lzani_ctx*> ctx;
    std::vector<int> devs;
    std::vector<ncclComm_t> comms;      // one per device; empty when the group has one device or is a rehearsal
    bool rehearsal = false;             // the same device listed more than once: shards move by device copies
    std::string err;
    double gather_ms = 0;
    // buffers of lzani_group_run_rows, kept from call to call and grown when a call needs more (a tiled all2all makes ten
    // calls of the same size: no allocation after the first): the gathered / CSR-ordered results on the first device, a
    // shard buffer per peer, the scatter table, and two pinned staging buffers for the copy out
    int* d_all = nullptr;
    int* d_final = nullptr;
    size_t res_cap = 0;                 // results (12 B each) d_all / d_final hold
    std::vector<int*> d_shard;          // [0] unused (the first device writes into d_all)
    std::vector<size_t> shard_cap;
    unsigned long long* d_tab = nullptr;
    size_t tab_cap = 0;
    char* h_stage[2] = {nullptr, nullptr};
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
};
enum : size_t { GROUP_STAGE_BYTES = (size_t)32 << 20 };
static void group_free_buffers(lzani_group* g)
{
    if (!g->ctx.empty()) hipSetDevice(g->ctx[0]->dev);
    hipFree(g->d_all); hipFree(g->d_final); hipFree(g->d_tab);
    g->d_all = g->d_final = nullptr; g->d_tab = nullptr; g->res_cap = g->tab_cap = 0;
    for (int k = 0; k < 2; ++k) {
        if (g->h_stage[k]) hipHostFree(g->h_stage[k]);
        if (g->ev_stage[k]) hipEventDestroy(g->ev_stage[k]);
        g->h_stage[k] = nullptr; g->ev_stage[k] = nullptr;
    }
    for (size_t d = 1; d < g->d_shard.size(); ++d)
        if (g->d_shard[d]) { hipSetDevice(g->ctx[d]->dev); hipFree(g->d_shard[d]); g->d_shard[d] = nullptr; }
    g->shard_cap.assign(g->shard_cap.size(), 0);
}

static void comm_release(lzani_ctx* c)
{
    if (c->comm) { ncclCommDestroy((ncclComm_t)c->comm); c->comm = nullptr; }
}

extern "C" {

// ---- rows -> shards ---------------------------------------------------------------------------------
int lzani_row_costs(uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off, const uint32_t* query_ids,
                    uint32_t n, const uint32_t* len, uint64_t* cost)
{
    if (!ref_ids || !row_off || !len || !cost) return LZANI_ERR_ARG;
    u64 total = 0;
    if (!query_ids) for (u32 g = 0; g < n; ++g) total += len[g];
    for (u32 k = 0; k < n_rows; ++k) {
        if (ref_ids[k] >= n) return LZANI_ERR_ARG;
        u64 c = (u64)LZANI_ROW_COST_REF_WEIGHT * len[ref_ids[k]];
        if (!query_ids) c += total - len[ref_ids[k]];
        else
            for (u64 e = row_off[k]; e < row_off[k + 1]; ++e) {
                if (query_ids[e] >= n) return LZANI_ERR_ARG;
                c += len[query_ids[e]];
            }
        cost[k] = c;
    }
    return LZANI_OK;
}

int lzani_partition_rows(uint32_t n_rows, const uint64_t* row_cost, uint32_t n_parts, uint32_t* part_of_row)
{
    if (!n_parts || (n_rows && !part_of_row)) return LZANI_ERR_ARG;
    if (!row_cost) {                                          // equal rows: cyclic in the given (length-descending) order
        for (u32 k = 0; k < n_rows; ++k) part_of_row[k] = k % n_parts;
        return LZANI_OK;
    }
    // greedy LPT: heaviest row first onto the least loaded shard (ties: earlier row, lower shard) -- the 4/3 rule
    std::vector<u32> order(n_rows);
    std::iota(order.begin(), order.end(), 0u);
    std::stable_sort(order.begin(), order.end(), [&](u32 a, u32 b) { return row_cost[a] > row_cost[b]; });
    typedef std::pair<u64, u32> Load;                          // (load, shard): a min-heap
    std::vector<Load> heap;
    for (u32 p = 0; p < n_parts; ++p) heap.emplace_back(0, p);
    auto cmp = [](const Load& a, const Load& b) { return a > b; };
    std::make_heap(heap.begin(), heap.end(), cmp);
    for (u32 k : order) {
        std::pop_heap(heap.begin(), heap.end(), cmp);
        Load& l = heap.back();
        part_of_row[k] = l.second;
        l.first += row_cost[k];
        std::push_heap(heap.begin(), heap.end(), cmp);
    }
    return LZANI_OK;
}

// ---- one process per GPU ------------------------------------------------------------------------------
int lzani_comm_unique_id(uint8_t* id)
{
    if (!id) return LZANI_ERR_ARG;
    static_assert(sizeof(ncclUniqueId) == LZANI_UNIQUE_ID_BYTES, "unique id size");
    ncclUniqueId u;
    if (ncclGetUniqueId(&u) != ncclSuccess) return LZANI_ERR_DEVICE;
    memcpy(id, &u, sizeof u);
    return LZANI_OK;
}

int lzani_comm_init(lzani_ctx* c, uint32_t n_ranks, uint32_t rank, const uint8_t* id)
{
    if (!c) return LZANI_ERR_ARG;
    if (!id || !n_ranks || rank >= n_ranks) return fail(c, LZANI_ERR_ARG, "lzani_comm_init: bad rank / id");
    if (c->comm) return fail(c, LZANI_ERR_STATE, "lzani_comm_init: communicator exists already");
    HIPCHK(c, hipSetDevice(c->dev));
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    RCCLCHK(c, ncclCommInitRank(&comm, (int)n_ranks, u, (int)rank));
    c->comm = comm; c->n_ranks = n_ranks; c->rank = rank;
    return LZANI_OK;
}

int lzani_comm_allgather(lzani_ctx* c, const void* d_send, void* d_recv, uint64_t n_results)
{
    if (!c) return LZANI_ERR_ARG;
    if (!c->comm) return fail(c, LZANI_ERR_STATE, "lzani_comm_allgather: no communicator (lzani_comm_init)");
    if (n_results && (!d_send || !d_recv)) return fail(c, LZANI_ERR_ARG, "lzani_comm_allgather: null buffer");
    HIPCHK(c, hipSetDevice(c->dev));
    RCCLCHK(c, ncclAllGather(d_send, d_recv, (size_t)n_results * 3, ncclInt32, (ncclComm_t)c->comm, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LZANI_OK;
}

int lzani_comm_gatherv(lzani_ctx* c, const void* d_send, void* d_recv, const uint64_t* counts, uint32_t root)
{
    if (!c) return LZANI_ERR_ARG;
    if (!c->comm) return fail(c, LZANI_ERR_STATE, "lzani_comm_gatherv: no communicator (lzani_comm_init)");
    if (!counts || root >= c->n_ranks) return fail(c, LZANI_ERR_ARG, "lzani_comm_gatherv: bad argument");
    HIPCHK(c, hipSetDevice(c->dev));
    ncclComm_t comm = (ncclComm_t)c->comm;
    if (c->rank == root) {
        if (!d_recv) return fail(c, LZANI_ERR_ARG, "lzani_comm_gatherv: null receive buffer on the root");
        u64 off = 0;
        for (u32 p = 0; p < root; ++p) off += counts[p];
        if (counts[root] && (int*)d_recv + 3 * off != d_send)          // the root's own shard
            HIPCHK(c, hipMemcpyAsync((int*)d_recv + 3 * off, d_send, counts[root] * 12, hipMemcpyDeviceToDevice, c->stream));
        ncclResult_t r = ncclGroupStart();
        off = 0;
        for (u32 p = 0; p < c->n_ranks && r == ncclSuccess; ++p) {
            if (p != root && counts[p]) r = ncclRecv((int*)d_recv + 3 * off, (size_t)counts[p] * 3, ncclInt32, (int)p, comm, c->stream);
            off += counts[p];
        }
        const ncclResult_t r2 = ncclGroupEnd();
        RCCLCHK(c, r);
        RCCLCHK(c, r2);
    } else if (counts[c->rank])
        RCCLCHK(c, ncclSend(d_send, (size_t)counts[c->rank] * 3, ncclInt32, (int)root, comm, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return LZANI_OK;
}

// ---- one process, n GPUs ------------------------------------------------------------------------------
static int gfail(lzani_group* g, int code, const std::string& msg) { if (g) g->err = msg; return code; }
static thread_local std::string t_group_create_err;      // why the last lzani_group_create of this thread failed (there is no group to hold it)

// The shard bookkeeping of lzani_group_run_rows as a pure host function (exported so that it can be tested without a
// GPU): rows keep their order inside a shard, shards follow each other in the gathered buffer; entry j of the scatter
// table (j counts the rows shard by shard) says where the row's results sit in that buffer, where they belong in the
// caller's CSR order, and how many there are.
int lzani_plan_gather(uint32_t n_rows, const uint64_t* row_off, const uint32_t* part_of_row, uint32_t n_parts,
                      uint64_t* shard_base, uint64_t* src, uint64_t* dst, uint64_t* cnt, uint32_t* row_of_entry)
{
    if (!n_parts || (n_rows && (!row_off || !part_of_row)) || !shard_base || (n_rows && (!src || !dst || !cnt))) return LZANI_ERR_ARG;
    std::vector<u64> pairs(n_parts, 0);
    for (u32 k = 0; k < n_rows; ++k) {
        if (part_of_row[k] >= n_parts || row_off[k + 1] < row_off[k]) return LZANI_ERR_ARG;
        pairs[part_of_row[k]] += row_off[k + 1] - row_off[k];
    }
    shard_base[0] = 0;
    for (u32 d = 0; d < n_parts; ++d) shard_base[d + 1] = shard_base[d] + pairs[d];
    std::vector<u64> at(shard_base, shard_base + n_parts);          // next free result of every shard
    std::vector<u32> first(n_parts + 1, 0);                         // table entries of shard d: first[d] .. first[d + 1]
    for (u32 k = 0; k < n_rows; ++k) ++first[part_of_row[k] + 1];
    for (u32 d = 0; d < n_parts; ++d) first[d + 1] += first[d];
    std::vector<u32> fill(first.begin(), first.end() - 1);
    for (u32 k = 0; k < n_rows; ++k) {
        const u32 d = part_of_row[k], j = fill[d]++;
        src[j] = at[d]; dst[j] = row_off[k]; cnt[j] = row_off[k + 1] - row_off[k];
        if (row_of_entry) row_of_entry[j] = k;
        at[d] += cnt[j];
    }
    return LZANI_OK;
}

void lzani_group_destroy(lzani_group* g)
{
    if (!g) return;
    group_free_buffers(g);
    for (auto cm : g->comms) if (cm) ncclCommDestroy(cm);
    for (auto c : g->ctx) lzani_destroy(c);
    delete g;
}

// (a null group: the reason the last lzani_group_create of the calling thread failed, if it did)
const char* lzani_group_last_error(const lzani_group* g)
{
    return g ? g->err.c_str() : (t_group_create_err.empty() ? "null group" : t_group_create_err.c_str());
}

int lzani_group_create(const lzani_params* p, uint32_t n_devices, const int* device_ids, lzani_group** out)
{
    t_group_create_err.clear();
    if (!p || !out || !n_devices || !device_ids) { t_group_create_err = "lzani_group_create: null argument"; return LZANI_ERR_ARG; }
    *out = nullptr;
    lzani_group* g = new (std::nothrow) lzani_group();
    if (!g) return LZANI_ERR_NOMEM;
    g->devs.assign(device_ids, device_ids + n_devices);
    for (u32 d = 0; d < n_devices; ++d) {
        lzani_ctx* c = nullptr;
        int rc = lzani_create(p, device_ids[d], &c);
        if (rc != LZANI_OK) {
            t_group_create_err = "lzani_create on device " + std::to_string(device_ids[d]) + " failed with code " + std::to_string(rc) +
                                 (rc == LZANI_ERR_PARAMS ? " (LZ parameters outside the supported envelope)" : rc == LZANI_ERR_DEVICE ? " (no such HIP device, or its stream could not be made)" : "");
            lzani_group_destroy(g);
            return rc;
        }
        g->ctx.push_back(c);
    }
    std::vector<int> sorted(g->devs);
    std::sort(sorted.begin(), sorted.end());
    g->rehearsal = std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end();
    if (n_devices > 1 && !g->rehearsal) {
        g->comms.assign(n_devices, nullptr);
        ncclResult_t r = ncclCommInitAll(g->comms.data(), (int)n_devices, g->devs.data());
        if (r != ncclSuccess) {
            t_group_create_err = std::string("ncclCommInitAll: ") + ncclGetErrorString(r);
            g->comms.clear();
            lzani_group_destroy(g);
            return LZANI_ERR_DEVICE;
        }
    }
    *out = g;
    return LZANI_OK;
}

int lzani_group_set_genomes(lzani_group* g, uint32_t n, const uint8_t* const* codes, const uint32_t* len)
{
    if (!g) return LZANI_ERR_ARG;
    std::vector<int> rc(g->ctx.size(), LZANI_OK);
    std::vector<std::thread> th;
    auto one = [&](size_t d) { rc[d] = lzani_set_genomes(g->ctx[d], n, codes, len); };
    for (size_t d = 1; d < g->ctx.size(); ++d) th.emplace_back(one, d);
    one(0);
    for (auto& t : th) t.join();
    for (size_t d = 0; d < g->ctx.size(); ++d)
        if (rc[d] != LZANI_OK) return gfail(g, rc[d], "device " + std::to_string(g->devs[d]) + ": " + lzani_last_error(g->ctx[d]));
    return LZANI_OK;
}

int lzani_group_run_rows(lzani_group* g, uint32_t n_rows, const uint32_t* ref_ids, const uint64_t* row_off,
                         const uint32_t* query_ids, lzani_result* out)
{
    if (!g) return LZANI_ERR_ARG;
    if (!ref_ids || !row_off) return gfail(g, LZANI_ERR_ARG, "lzani_group_run_rows: null argument");
    const u32 nd = (u32)g->ctx.size();
    const u64 n_pairs = n_rows ? row_off[n_rows] : 0;
    if (n_pairs && !out) return gfail(g, LZANI_ERR_ARG, "lzani_group_run_rows: null output");
    g->gather_ms = 0;
    lzani_ctx* c0 = g->ctx[0];
    if (!c0->n) return gfail(g, LZANI_ERR_STATE, "lzani_group_run_rows: no genomes set");
    for (u32 k = 0; k < n_rows; ++k)
        if (ref_ids[k] >= c0->n || row_off[k + 1] < row_off[k]) return gfail(g, LZANI_ERR_ARG, "lzani_group_run_rows: bad row table");
    if (query_ids)
        for (u64 e = 0; e < n_pairs; ++e) if (query_ids[e] >= c0->n) return gfail(g, LZANI_ERR_ARG, "lzani_group_run_rows: query id out of range");

    // rows -> devices
    std::vector<u32> part(n_rows);
    {
        std::vector<u64> cost;
        if (query_ids) {
            std::vector<u32> len(c0->L.begin(), c0->L.end());
            cost.resize(n_rows);
            int rc = lzani_row_costs(n_rows, ref_ids, row_off, query_ids, c0->n, len.data(), cost.data());
            if (rc != LZANI_OK) return gfail(g, rc, "lzani_group_run_rows: row costs");
        }
        lzani_partition_rows(n_rows, query_ids ? cost.data() : nullptr, nd, part.data());
    }
    struct Shard { std::vector<u32> rows, ref, q; std::vector<u64> off; u64 base = 0; };
    std::vector<Shard> sh(nd);
    for (auto& s : sh) s.off.push_back(0);
    for (u32 k = 0; k < n_rows; ++k) {
        Shard& s = sh[part[k]];
        s.rows.push_back(k);
        s.ref.push_back(ref_ids[k]);
        if (query_ids) s.q.insert(s.q.end(), query_ids + row_off[k], query_ids + row_off[k + 1]);
        s.off.push_back(s.off.back() + (row_off[k + 1] - row_off[k]));
    }
    for (u32 d = 1; d < nd; ++d) sh[d].base = sh[d - 1].base + sh[d - 1].off.back();

    // device 0 holds the gathered buffer (its own shard is written in place) and the CSR-ordered copy
    {
        const hipError_t e = hipSetDevice(c0->dev);
        if (e != hipSuccess) return gfail(g, LZANI_ERR_DEVICE, std::string("hipSetDevice: ") + hipGetErrorString(e));
    }
    // (grown, never shrunk; a failed growth leaves the group without buffers and the call with LZANI_ERR_NOMEM)
    if (g->res_cap < std::max<u64>(n_pairs, 1)) {
        hipFree(g->d_all); hipFree(g->d_final);
        g->d_all = g->d_final = nullptr; g->res_cap = 0;
        if (hipMalloc(&g->d_all, std::max<u64>(n_pairs, 1) * 12) != hipSuccess || hipMalloc(&g->d_final, std::max<u64>(n_pairs, 1) * 12) != hipSuccess) {
            (void)hipGetLastError();
            hipFree(g->d_all); g->d_all = nullptr;
            return gfail(g, LZANI_ERR_NOMEM, "lzani_group_run_rows: result buffers on device 0");
        }
        g->res_cap = std::max<u64>(n_pairs, 1);
    }
    int* const d_all = g->d_all;
    int* const d_final = g->d_final;
    g->d_shard.resize(nd, nullptr);
    g->shard_cap.resize(nd, 0);
    std::vector<int*>& d_shard = g->d_shard;
    d_shard[0] = d_all;
    std::vector<int> rc(nd, LZANI_OK);
    auto one = [&](u32 d) {
        lzani_ctx* c = g->ctx[d];
        if (sh[d].ref.empty()) { c->tm = lzani_timing{}; return; }       // no row for this device
        if (d && g->shard_cap[d] < std::max<u64>(sh[d].off.back(), 1)) {
            if (hipSetDevice(c->dev) != hipSuccess) { rc[d] = LZANI_ERR_DEVICE; return; }
            hipFree(d_shard[d]); d_shard[d] = nullptr; g->shard_cap[d] = 0;
            if (hipMalloc(&d_shard[d], std::max<u64>(sh[d].off.back(), 1) * 12) != hipSuccess) { (void)hipGetLastError(); d_shard[d] = nullptr; rc[d] = LZANI_ERR_NOMEM; return; }
            g->shard_cap[d] = std::max<u64>(sh[d].off.back(), 1);
        }
        rc[d] = lzani_run_rows_device(c, (u32)sh[d].ref.size(), sh[d].ref.data(), sh[d].off.data(),
                                      query_ids ? sh[d].q.data() : nullptr, d_shard[d]);
    };
    {
        std::vector<std::thread> th;
        for (u32 d = 1; d < nd; ++d) th.emplace_back(one, d);
        one(0);
        for (auto& t : th) t.join();
    }
    int ret = LZANI_OK;
    for (u32 d = 0; d < nd && ret == LZANI_OK; ++d)
        if (rc[d] != LZANI_OK) ret = gfail(g, rc[d], "device " + std::to_string(g->devs[d]) + ": " + lzani_last_error(g->ctx[d]));

    // the gather: every peer's shard to device 0 (grouped ncclSend / ncclRecv over xGMI)
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ret == LZANI_OK) {
        hipSetDevice(c0->dev);
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, c0->stream);
        if (!g->comms.empty()) {
            ncclResult_t r = ncclGroupStart();
            for (u32 d = 1; d < nd && r == ncclSuccess; ++d) {
                const size_t cnt = (size_t)sh[d].off.back() * 3;
                if (!cnt) continue;
                hipSetDevice(g->ctx[d]->dev);
                r = ncclSend(d_shard[d], cnt, ncclInt32, 0, g->comms[d], g->ctx[d]->stream);
                hipSetDevice(c0->dev);
                if (r == ncclSuccess) r = ncclRecv(d_all + 3 * sh[d].base, cnt, ncclInt32, (int)d, g->comms[0], c0->stream);
            }
            ncclResult_t r2 = ncclGroupEnd();
            if (r == ncclSuccess) r = r2;
            if (r != ncclSuccess) ret = gfail(g, LZANI_ERR_DEVICE, std::string("RCCL gather: ") + ncclGetErrorString(r));
        } else {
            for (u32 d = 1; d < nd; ++d)                         // rehearsal on one device: plain device copies
                if (sh[d].off.back() && hipMemcpyAsync(d_all + 3 * sh[d].base, d_shard[d], sh[d].off.back() * 12, hipMemcpyDeviceToDevice, c0->stream) != hipSuccess)
                    ret = gfail(g, LZANI_ERR_DEVICE, "device copy of a shard failed");
        }
    }
    if (ret == LZANI_OK && n_rows) {
        // shard order -> the caller's CSR order on device 0, then the one device-to-host copy
        std::vector<u64> tab(3 * (size_t)n_rows), sbase((size_t)nd + 1);
        if (lzani_plan_gather(n_rows, row_off, part.data(), nd, sbase.data(), tab.data(), tab.data() + n_rows, tab.data() + 2 * (size_t)n_rows, nullptr) != LZANI_OK)
            ret = gfail(g, LZANI_ERR_ARG, "lzani_group_run_rows: gather plan");
        for (u32 d = 0; d < nd && ret == LZANI_OK; ++d)
            if (sbase[d] != sh[d].base) ret = gfail(g, LZANI_ERR_STATE, "lzani_group_run_rows: gather plan and shards disagree");
        hipError_t e = ret == LZANI_OK ? hipSuccess : hipErrorInvalidValue;
        if (e == hipSuccess && g->tab_cap < tab.size()) {
            hipFree(g->d_tab); g->d_tab = nullptr; g->tab_cap = 0;
            e = hipMalloc(&g->d_tab, tab.size() * 8);
            if (e == hipSuccess) g->tab_cap = tab.size();
        }
        unsigned long long* const d_tab = g->d_tab;
        if (e == hipSuccess) e = hipMemcpyAsync(d_tab, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, c0->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_scatter_rows, dim3(n_rows), dim3(256), 0, c0->stream, d_all, d_final, (const u64*)d_tab, (const u64*)d_tab + n_rows, (const u64*)d_tab + 2 * (size_t)n_rows);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipEventRecord(e1, c0->stream);
        // the copy out: through two pinned staging buffers that take turns -- the device-to-host copy of one piece flies
        // while the host moves the piece before it into the caller's (pageable) buffer
        for (int k = 0; k < 2 && e == hipSuccess; ++k) {
            if (!g->h_stage[k]) e = hipHostMalloc((void**)&g->h_stage[k], GROUP_STAGE_BYTES, hipHostMallocDefault);
            if (e == hipSuccess && !g->ev_stage[k]) e = hipEventCreateWithFlags(&g->ev_stage[k], hipEventDisableTiming);
        }
        {
            const size_t total = (size_t)n_pairs * 12;
            size_t issued = 0, moved = 0;
            size_t len[2] = {0, 0};
            for (int k = 0; e == hipSuccess && (issued < total || moved < total); k ^= 1) {
                if (len[k]) {                                  // the piece this buffer holds: wait for it, hand it over
                    e = hipEventSynchronize(g->ev_stage[k]);
                    if (e == hipSuccess) memcpy((char*)out + moved, g->h_stage[k], len[k]);
                    moved += len[k];
                    len[k] = 0;
                }
                if (e == hipSuccess && issued < total) {
                    len[k] = std::min<size_t>(GROUP_STAGE_BYTES, total - issued);
                    e = hipMemcpyAsync(g->h_stage[k], (const char*)d_final + issued, len[k], hipMemcpyDeviceToHost, c0->stream);
                    if (e == hipSuccess) e = hipEventRecord(g->ev_stage[k], c0->stream);
                    issued += len[k];
                }
            }
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c0->stream);
        for (u32 d = 1; d < nd && e == hipSuccess; ++d) { hipSetDevice(g->ctx[d]->dev); e = hipStreamSynchronize(g->ctx[d]->stream); }
        if (ret != LZANI_OK) {}                                   // (the gather plan failed: its message stands)
        else if (e != hipSuccess) ret = gfail(g, LZANI_ERR_DEVICE, std::string("gather / copy out: ") + hipGetErrorString(e));
        else { float ms = 0; hipEventElapsedTime(&ms, e0, e1); g->gather_ms = ms; }
    }
    hipSetDevice(c0->dev);
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    return ret;
}

int lzani_group_get_timing(const lzani_group* g, uint32_t device_index, lzani_timing* t, double* gather_ms)
{
    if (!g || device_index >= g->ctx.size() || !t) return LZANI_ERR_ARG;
    *t = g->ctx[device_index]->tm;
    if (gather_ms) *gather_ms = g->gather_ms;
    return LZANI_OK;
}

}  // extern "C"
