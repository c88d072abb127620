// lzani_kernels_cand.h -- candidate detection for DENSE rows, once per (query, group of references) instead of once
// per pair.  Included by lzani_hip.hip only (after lzani_kernels_index.h).
//
// What it replaces: the anchor lookup of every scan step of CParser::parse -- kmer -> hash_mm -> ht_long probe
// (/root/reference/src/parser.cpp:507-531 and 585-602), 40,000 probes per 40 kbp pair, 98 % of which find nothing.
// In a dense all2all every query meets every reference of the batch, so the question "which references hold this
// mal-mer" is asked ONCE per query position for a whole group of references:
//
//   k_pm_build   the PRESENCE MATRIX of a group of up to PM_GROUP reference slots: row h (the mixed mal-mer hash,
//                low mbits bits) holds one bit per slot -- set iff the slot's reference text holds a mal-mer with that
//                hash.  2^22 rows x 64 B at viral defaults (256 MB per group of 512 references; exact: the mixer is a
//                bijection on the 22 key bits).
//   k_pm_cand    one block per (query, tile of 1,024 query positions): every position reads its matrix row (one
//                64-byte read instead of 512 tag-word probes), the set bits are scattered into an LDS tile
//                [slot][1,024 positions] and the tile goes out as one 128-byte line per pair: the pair's CANDIDATE
//                BITMAP (bit p = query position p has an anchor candidate in that reference).
//
// The pair kernel then reads its candidates 64 positions per bitmap word (DevWave::refill, the form the join of the
// long genomes already feeds) and touches the reference's index only for the ~2 % of positions that are candidates.
// Per pair this removes the 160 KB stream of k-mer words, the ~11 k random tag-word lines and the ~36 k detect
// instructions of the probe form.
#pragma once

namespace lzani {

enum { PM_GROUP = 512, PM_TILE = 1024, PM_TILE_WORDS = PM_TILE / 32 };

struct PmArgs {
    GenomeTab G;
    const u32* ref_ids;       // device, batch-relative: slot -> genome
    const u64* row_off;       // device, batch-relative: absolute pair offset of the slot's dense row
    u32 slot0, rows;          // the group: slots [slot0, slot0 + rows), rows <= PM_GROUP
    u32* M;                   // 2^mbits rows of rw words
    u32 rw;                   // words per matrix row: rows rounded up to 128 slots (16-byte loads)
    u32 mmask;                // 2^mbits - 1
    int rshift;               // matrix row of a mixed hash h = h >> rshift (its top mbits bits; 0 = the matrix is exact: one row per k-mer).
                              // A matrix of fewer rows than k-mers (mid-size genomes with long k-mers: 2^30 rows would be cleared and
                              // written for 2^18 k-mers a slot) answers "maybe" for k-mers that share a row: a candidate bit the pair
                              // kernel then finds no bucket entry for and drops (refill: dead) -- every reader of a bitmap verifies.
    int mal, mrd;
    u32* cbits;               // candidate bitmaps of the batch's pairs, cb_words 32-bit words each
    u64 cb_words;
    u64 e0;                   // absolute pair offset of the batch's first pair
    u32 n;                    // genomes (a dense row holds the n - 1 others, ascending)
    u32 q0;                   // k_pm_cand: first query (dense rows) / first list entry (query lists) of this launch (gridDim.y is limited)
    // rows with query lists (the filtered form of lz_matcher.cpp:234-250, or the row x column blocks a host cuts a dense
    // all2all into): query_ids = the lists, CSR with row_off; k_pm_pairs turns them into
    //   pidx[q * RP + s]   the pair of query q in the row of slot s of the group (batch-relative), ~0 = none
    //   qlist / qcount     the queries that occur in the group at all -- only these get candidate bitmaps
    // nullptr = dense rows.
    const u32* query_ids;
    u32* pidx;
    u32* qflag;
    u32* qlist;
    u32* qcount;
    // batches of few, long pairs: pcount[pair of the batch] += the pair's candidates (the pair kernel then takes the pairs
    // with the most candidates -- the related ones, ten times the work of a chance pair -- first); nullptr = not counted
    u32* pcount;
};

// One thread per text position of the group's references: the slot's bit in the row of the position's mal-mer.
__global__ void __launch_bounds__(256) k_pm_build(PmArgs a, int Tmax)
{
    const u32 s = blockIdx.y;                          // slot inside the group
    const u32 g = a.ref_ids[a.slot0 + s];
    const int T = ref_text_len(a.G.L[g], a.mrd);
    const u32* km = a.G.kmL + 64 * a.G.nmoff[g];
    const u32 bit = 1u << (s & 31), w = s >> 5;
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p + a.mal <= T && p < Tmax; p += gridDim.x * blockDim.x) {
        const u32 h = km[p];
        if (h != KM_INVALID) atomicOr(&a.M[(u64)((h >> a.rshift) & a.mmask) * a.rw + w], bit);
    }
}

// The same matrix straight from the group's anchor indexes, without a global atomic and without clearing it first (round 4;
// long genomes: 1.3 G memory-side atomicOr per 128 x 5 Mbp references were 70 ms of a 480 ms step).  The index of a slot
// is its mal-mers sorted by mixed hash -- bucket = the hash's top dirbits (directory), the rest the entry's tag, exact
// here -- so the matrix rows [c << rcl, (c + 1) << rcl) are the buckets [c << (rcl - tb), ..) of EVERY slot: one block
// per chunk c reads those few buckets of each slot (a contiguous piece of its directory and of its entries), sets the
// bits in an LDS copy of the chunk and writes the chunk out whole -- every matrix row exactly once, coalesced.
//   needs: tags exact (tagmask = all tb bits), row = hash (mbits = kb), rcl >= tb
template <int RW>          // words per matrix row
__global__ void __launch_bounds__(1024) k_pm_from_index(PmArgs a, const u32* __restrict__ dirz, const u32* __restrict__ ent,
                                                        u64 dir_stride, u64 ent_stride, int tb, int posbits, int rcl)
{
    extern __shared__ u32 s_rows[];                    // (1 << rcl) rows of RW words
    const u32 nwords = (u32)RW << rcl, nbk_l2 = (u32)(rcl - tb), nbk = 1u << nbk_l2;
    const u32 b0 = blockIdx.x << nbk_l2;
    for (u32 k = threadIdx.x; k < nwords; k += 1024) s_rows[k] = 0;
    __syncthreads();
    const u32 items = a.rows << nbk_l2;                // (slot of the group, bucket of the chunk)
    enum { U = 8 };
    for (u32 it0 = threadIdx.x; it0 < items; it0 += U * 1024) {
        u32 s[U], e[U], first[U];
#pragma unroll
        for (int k = 0; k < U; ++k) {                  // the buckets' entry ranges, all requests first
            const u32 it = it0 + (u32)k * 1024u;
            s[k] = e[k] = 0;
            if (it < items) {
                const u32* d = dirz + (u64)(a.slot0 + (it >> nbk_l2)) * dir_stride + b0 + (it & (nbk - 1));
                s[k] = d[0]; e[k] = d[1];
            }
            if (e[k] < s[k] || e[k] - s[k] > (1u << 28)) { LZ_GUARD_TRIP(8); e[k] = s[k]; }
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {                  // a bucket holds 0.6 entries on average: the first of each, together
            const u32 it = it0 + (u32)k * 1024u;
            first[k] = s[k] < e[k] ? ent[(u64)(a.slot0 + (it >> nbk_l2)) * ent_stride + s[k]] : 0u;
        }
#pragma unroll
        for (int k = 0; k < U; ++k) {
            if (s[k] >= e[k]) continue;
            const u32 it = it0 + (u32)k * 1024u, slot = it >> nbk_l2, rbase = (it & (nbk - 1)) << tb;
            const u32 bit = 1u << (slot & 31), w = slot >> 5;
            atomicOr(&s_rows[(rbase | (first[k] >> posbits)) * RW + w], bit);
            const u32* v = ent + (u64)(a.slot0 + slot) * ent_stride;
            for (u32 j = s[k] + 1; j < e[k]; ++j) atomicOr(&s_rows[(rbase | (v[j] >> posbits)) * RW + w], bit);
        }
    }
    __syncthreads();
    uint4* out = reinterpret_cast<uint4*>(a.M + ((u64)blockIdx.x << rcl) * RW);
    const uint4* src = reinterpret_cast<const uint4*>(s_rows);
    for (u32 k = threadIdx.x; k < nwords / 4; k += 1024) out[k] = src[k];
}

// Rows with query lists: one block per slot of the group walks the row's list (no query twice in a row: the host checks).
__global__ void __launch_bounds__(256) k_pm_pairs(PmArgs a)
{
    const u32 s = blockIdx.x, RP = 32u * a.rw;
    const u64 b = a.row_off[a.slot0 + s], e = a.row_off[a.slot0 + s + 1];
    for (u64 k = b + threadIdx.x; k < e; k += blockDim.x) {
        const u32 q = a.query_ids[k];
        a.pidx[(u64)q * RP + s] = (u32)(k - a.e0);
        if (atomicExch(&a.qflag[q], 1u) == 0u) a.qlist[atomicAdd(a.qcount, 1u)] = q;
    }
}

// LDS tile of k_pm_cand: word c (32 positions) of slot s at c * (RP + 1) + s, RP = 32 * rw slots: the scatter of a
// wave (64 consecutive positions = two words c, any slots) and the gather of the write-out (one slot, 32 words c)
// both fall on distinct banks.
__device__ __forceinline__ u32 pm_tile_at(u32 rp1, u32 c, u32 s) { return c * rp1 + s; }

// PM_CAND_THREADS threads a block: PM_TILE / PM_CAND_THREADS positions per thread (1,024: two blocks = 32 waves a CU beside 2 x 66 KB
// of LDS tiles -- the matrix rows are random 64-byte reads, and the eight waves a CU of 256-thread blocks hid too little of
// their latency: candidate stage of the 10k bench 23.8 -> 18.7 ms with 512 threads, ~17 ms with 1,024: round 4)
#ifndef LZANI_PM_CAND_THREADS
#define LZANI_PM_CAND_THREADS 1024
#endif
enum { PM_CAND_THREADS = LZANI_PM_CAND_THREADS, PM_CAND_PER = PM_TILE / PM_CAND_THREADS };
template <int RW4>         // RW4 = rw / 4: 16-byte loads per matrix row
__global__ void __launch_bounds__(PM_CAND_THREADS) k_pm_cand(PmArgs a)
{
    extern __shared__ u32 s_tile[];                    // 32 x (RP + 1) words, then RP pair indexes
    const u32 RP = 128u * RW4, rp1 = RP + 1;
    u32* const s_pair = s_tile + PM_TILE_WORDS * rp1;  // per slot: the pair's index in the batch, or ~0 (no pair)
    u32 q = a.q0 + blockIdx.y;
    if (a.qlist) {                                     // (block-uniform) query lists: the queries that occur in this group
        if (q >= *a.qcount) return;
        q = a.qlist[q];
    }
    const int D = a.G.L[q] + a.mrd;
    const int p0 = (int)blockIdx.x * PM_TILE;
    if (p0 >= D + 320) return;                         // (block-uniform) the pair kernel reads five words beyond its scan position at most
    for (u32 k = threadIdx.x; k < PM_TILE_WORDS * rp1; k += PM_CAND_THREADS) s_tile[k] = 0;
    for (u32 s = threadIdx.x; s < RP; s += PM_CAND_THREADS) {
        u32 pe = 0xFFFFFFFFu;
        if (a.pidx) pe = a.pidx[(u64)q * RP + s];
        else if (s < a.rows) {
            const u32 r = a.ref_ids[a.slot0 + s];
            if (r != q) pe = (u32)(a.row_off[a.slot0 + s] - a.e0) + q - (q > r ? 1u : 0u);
        }
        s_pair[s] = pe;
    }
    const u32* km = a.G.kmL + 64 * a.G.nmoff[q];
    // PM_CAND_PER positions per thread; their k-mer words, then their matrix rows, requested together
    u32 h[PM_CAND_PER];
    uint4 row[PM_CAND_PER][RW4];
#pragma unroll
    for (int k = 0; k < PM_CAND_PER; ++k) {
        const int p = p0 + k * PM_CAND_THREADS + (int)threadIdx.x;
        h[k] = p < D ? km[p] : KM_INVALID;
    }
#pragma unroll
    for (int k = 0; k < PM_CAND_PER; ++k) {
        const uint4* src = reinterpret_cast<const uint4*>(a.M + (u64)(h[k] == KM_INVALID ? 0u : ((h[k] >> a.rshift) & a.mmask)) * a.rw);
#pragma unroll
        for (int j = 0; j < RW4; ++j) row[k][j] = src[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PM_CAND_PER; ++k) {
        const u32 pl = (u32)k * (u32)PM_CAND_THREADS + threadIdx.x;    // position inside the tile
        const u32 c = pl >> 5, bit = 1u << (pl & 31);
        const bool ok = h[k] != KM_INVALID;
#pragma unroll
        for (int j = 0; j < RW4; ++j) {
            const u32 xs[4] = {row[k][j].x, row[k][j].y, row[k][j].z, row[k][j].w};
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                u32 x = ok ? xs[d] : 0u;
                while (x) {
                    const u32 b = (u32)__builtin_ctz(x);
                    x &= x - 1;
                    atomicOr(&s_tile[pm_tile_at(rp1, c, 128u * j + 32u * d + b)], bit);
                }
            }
        }
    }
    __syncthreads();
    // write-out: 32 lanes = the 128-byte line of one pair
    const u32 c = threadIdx.x & 31;
    for (u32 s = threadIdx.x >> 5; s < a.rows; s += PM_CAND_THREADS / 32) {
        const u32 pe = s_pair[s];
        const u32 wv = s_tile[pm_tile_at(rp1, c, s)];
        if (pe != 0xFFFFFFFFu) a.cbits[(u64)pe * a.cb_words + (u64)blockIdx.x * PM_TILE_WORDS + c] = wv;
        if (a.pcount) {                                // (block-uniform) the tile's candidates of this pair: one atomic per line
            u32 n = pe != 0xFFFFFFFFu ? (u32)__builtin_popcount(wv) : 0u;
            for (int d = 16; d >= 1; d >>= 1) n += __shfl_xor(n, d, 32);
            if (c == 0 && n) atomicAdd(&a.pcount[pe], n);
        }
    }
}

// Longest pairs first (batches of few, long pairs: a 5 Mbp launch is over when its slowest pair is, and a related pair
// is ten chance pairs' work).  One key per pair ticket of the batch -- its XCD queue, then the pair's candidate count
// descending (k_pm_cand's pcount), then the ticket -- sorted, the low words are the order the queues hand their tickets out
// in (PairArgs::torder).  Placement only: every pair is still computed once, by the same code.
struct QueueBounds { u32 v[9]; };              // rows [v[x], v[x + 1]) of the batch's queue order belong to XCD queue x
__global__ void __launch_bounds__(256) k_lpt_keys(const u32* __restrict__ qorder, const u64* __restrict__ qcum, QueueBounds qb8,
                                                  const u64* __restrict__ row_off, u64 e0, const u32* __restrict__ pcount,
                                                  unsigned long long* __restrict__ keys, u32 rows, u64 n_tickets)
{
    const u64 base = qcum[0];
    for (u64 t = (u64)blockIdx.x * blockDim.x + threadIdx.x; t < n_tickets; t += (u64)gridDim.x * blockDim.x) {
        u32 lo = 0, hi = rows;                          // the row of ticket t: the last one with qcum[row] <= base + t
        while (hi - lo > 1) { const u32 mid = (lo + hi) >> 1; if (qcum[mid] <= base + t) lo = mid; else hi = mid; }
        u32 x = 0;
        while (x + 1 < 8 && lo >= qb8.v[x + 1]) ++x;      // its queue
        const u64 e = row_off[qorder[lo]] + (base + t - qcum[lo]);
        const u32 cnt = pcount[e - e0];
        const u32 inv = 0xFFFFFu - (cnt > 0xFFFFFu ? 0xFFFFFu : cnt);
        keys[t] = ((unsigned long long)x << 52) | ((unsigned long long)inv << 32) | (unsigned long long)t;
    }
}

}  // namespace lzani
