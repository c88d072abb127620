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
        if (h != KM_INVALID) atomicOr(&a.M[(u64)(h & a.mmask) * a.rw + w], bit);
    }
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

template <int RW4>         // RW4 = rw / 4: 16-byte loads per matrix row
__global__ void __launch_bounds__(256) k_pm_cand(PmArgs a)
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
    for (u32 k = threadIdx.x; k < PM_TILE_WORDS * rp1; k += 256) s_tile[k] = 0;
    for (u32 s = threadIdx.x; s < RP; s += 256) {
        u32 pe = 0xFFFFFFFFu;
        if (a.pidx) pe = a.pidx[(u64)q * RP + s];
        else if (s < a.rows) {
            const u32 r = a.ref_ids[a.slot0 + s];
            if (r != q) pe = (u32)(a.row_off[a.slot0 + s] - a.e0) + q - (q > r ? 1u : 0u);
        }
        s_pair[s] = pe;
    }
    const u32* km = a.G.kmL + 64 * a.G.nmoff[q];
    // four positions per thread; their k-mer words, then their matrix rows, requested together
    u32 h[4];
    uint4 row[4][RW4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = p0 + k * 256 + (int)threadIdx.x;
        h[k] = p < D ? km[p] : KM_INVALID;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint4* src = reinterpret_cast<const uint4*>(a.M + (u64)(h[k] == KM_INVALID ? 0u : (h[k] & a.mmask)) * a.rw);
#pragma unroll
        for (int j = 0; j < RW4; ++j) row[k][j] = src[j];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const u32 pl = (u32)k * 256u + threadIdx.x;    // position inside the tile
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
    for (u32 s = threadIdx.x >> 5; s < a.rows; s += 8) {
        const u32 pe = s_pair[s];
        if (pe == 0xFFFFFFFFu) continue;
        a.cbits[(u64)pe * a.cb_words + (u64)blockIdx.x * PM_TILE_WORDS + c] = s_tile[pm_tile_at(rp1, c, s)];
    }
}

}  // namespace lzani
