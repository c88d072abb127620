// lzani_model.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Host build of the product's pair state machine (lz-ani_amd/csrc/lzani_core.h) with a
// lane-emulating Wave policy: each cross-lane primitive is a plain loop over 64 lanes.  It lets
// the CPU test suite (-m "not gpu") check the exact kernel formulation -- rounds, mismatch-mask
// folds, chunked extensions, gap-fill split -- against the oracle without a GPU.  It is not a
// fallback: nothing in lz-ani_amd/ links or loads it.
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../lz-ani_amd/csrc/lzani_core.h"
#include "../../lz-ani_amd/csrc/lzani_layout.h"
#include "lane_wave.h"
#include "queue_wave.h"

using namespace lzani;

namespace {

struct Genome {
    int L, T, D;
    std::vector<u64> t2, nm;
    std::vector<u32> dirz, ent, sdirz, sent, kmL, kmS;
    IndexView iv, sv;
    int mrd;
    bool nfree;
    TextView rview() const { return ref_view(t2.data(), nm.data(), L, mrd, nfree); }
    TextView qview() const { return qry_view(t2.data(), nm.data(), L, mrd, nfree); }
};

void pack_genome(Genome& g, const uint8_t* codes, int L, const Params& P)
{
    g.L = L; g.T = ref_text_len(L, P.mrd); g.D = L + P.mrd; g.mrd = P.mrd;
    g.nfree = true;
    for (int j = 0; j < L; ++j) if (codes[j] >= 4) g.nfree = false;
    size_t w2 = text_words2(g.T), wn = text_wordsN(g.T);
    g.t2.assign(w2, 0);
    g.nm.assign(wn, ~0ULL);
    auto put = [&](int p, int c) {
        if (c < 4) {
            g.t2[p >> 5] |= (u64)c << ((p & 31) * 2);
            g.nm[p >> 6] &= ~(1ULL << (p & 63));
        }
    };
    for (int j = 0; j < L; ++j) put(j, codes[j] < 4 ? codes[j] : 4);
    int rc0 = L + 2 * P.mrd;
    for (int j = 0; j < L; ++j) { int c = codes[L - 1 - j]; put(rc0 + j, c < 4 ? 3 - c : 4); }
}

void build_one(const Genome& g, int k, const IndexGeom& geo, IndexView& iv, std::vector<u32>& dirz, std::vector<u32>& ent)
{
    iv.kb = geo.kb; iv.dirbits = geo.dirbits; iv.posbits = geo.posbits; iv.tagmask = geo.tagmask; iv.bk = nullptr;
    size_t nb = (size_t)1 << geo.dirbits;
    dirz.assign(nb + 1, 0);
    TextView R = g.rview();
    std::vector<std::pair<u32, u32>> items;   // (bucket, entry)
    for (int p = 0; p + k <= g.T; ++p) {
        u64 key;
        if (!kmer_at(R, p, k, key)) continue;
        u32 b, tag;
        key_slot(iv, key, b, tag);
        items.emplace_back(b, (tag << geo.posbits) | (u32)p);
    }
    std::sort(items.begin(), items.end());
    ent.resize(items.size() + 1);
    for (size_t j = 0; j < items.size(); ++j) { ent[j] = items[j].second; dirz[items[j].first + 1]++; }
    for (size_t b = 0; b < nb; ++b) dirz[b + 1] += dirz[b];
    iv.dirz = dirz.data(); iv.ent = ent.data();
}
void build_index(Genome& g, const Params& P, const IndexGeom& geo)
{
    build_one(g, P.mal, geo, g.iv, g.dirz, g.ent);
}
// seed (msl) index + per-position k-mer words, for the lane-serial policy
void build_seed_index(Genome& g, const Params& P, int Tmax)
{
    build_one(g, P.msl, index_geometry(Tmax, P.msl), g.sv, g.sdirz, g.sent);
    TextView R = g.rview();
    g.kmL.assign((size_t)g.T + 192, 0xFFFFFFFFu);
    g.kmS.assign((size_t)g.T + 192, 0xFFFFFFFFu);
    if (P.mal <= 15 && P.msl <= 15)
        for (int p = 0; p < g.T; ++p) {
            u64 key;
            if (kmer_at(R, p, P.mal, key)) g.kmL[p] = (u32)mix_key(key, 2 * P.mal);
            if (kmer_at(R, p, P.msl, key)) g.kmS[p] = (u32)key;
        }
}

struct HostWave {
    const Params& P;
    TextView R, Q;
    IndexView I;

    std::vector<RegionCoords>* regs = nullptr;
    void emit_region(const RegionCoords& c) const { if (regs) regs->push_back(c); }
    void stamp(int) const {}
    void split_limit(int) const {}
    void split_taint_set(bool) const {}
    int uniform(int v) const { return v; }
    bool ext_record(u32&) const { return false; }
    u64 mism_fwd(int q0, int r0, int n) const
    {
        u64 m = 0;
        for (int j = 0; j < n; ++j) if (!sym_match(R, r0 + j, Q, q0 + j)) m |= 1ULL << j;
        return m;
    }
    u64 mism_bwd(int q0, int r0, int n) const
    {
        u64 m = 0;
        for (int j = 0; j < n; ++j) if (!sym_match(R, r0 - 1 - j, Q, q0 - 1 - j)) m |= 1ULL << j;
        return m;
    }
    void mism2(int qa, int ra, int da, int na, int qb, int rb, int db, int nb, u64& A, u64& B) const
    {
        A = da > 0 ? mism_fwd(qa, ra, na) : mism_bwd(qa + 1, ra + 1, na);
        B = db > 0 ? mism_fwd(qb, rb, nb) : mism_bwd(qb + 1, rb + 1, nb);
    }
    bool find_event(int i, int n, bool trk, int r_end, int lit, int& adv, int& bpos, int& blen) const
    {
        n = imin(n, 64);                                   // a round of up to 64 steps, as the device's round path
        for (int l = 0; l < n; ++l) {
            int bp, bl;
            eval_step(P, R, Q, I, i + l, trk && (lit + l <= P.mqd), r_end, lit + l, bp, bl);
            if (bl >= P.msl) { adv = l; bpos = bp; blen = bl; return true; }
        }
        adv = n;
        return false;
    }
    ExtMasks ext_scan(u64 prevB, u64 B, int n) const
    {
        ExtMasks m{0, 0};
        for (int j = 0; j < 64; ++j) {
            bool b, q;
            ext_lane(prevB, B, j, n, P.aw, P.am, P.ar, b, q);
            if (b) m.brk |= 1ULL << j;
            if (q) m.qual |= 1ULL << j;
        }
        return m;
    }
    int best_split(u64 Lm, u64 Rm, int to_scan) const
    {
        int best = -1, bs = 0;
        for (int s = 0; s <= to_scan; ++s) {
            int v = popc64(Lm & lowmask(s)) + (s >= 64 ? 0 : popc64(Rm >> s));
            if (v >= best) { best = v; bs = s; }
        }
        return bs;
    }
};

}  // namespace

extern "C" {

// out[(r*n + q)*3 ..] = parse(query=q, ref=r); diagonal zero.  Returns 0, or -1 on unsupported params.
int model_all2all(uint32_t n, const uint8_t* const* codes, const uint32_t* len, const int32_t* p8, int32_t* out)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P)) return -1;
    uint32_t maxL = 0;
    for (uint32_t i = 0; i < n; ++i) maxL = std::max(maxL, len[i]);
    IndexGeom geo = index_geometry(ref_text_len((int)maxL, P.mrd), P.mal);
    std::vector<Genome> G(n);
    for (uint32_t i = 0; i < n; ++i) { pack_genome(G[i], codes[i], (int)len[i], P); build_index(G[i], P, geo); }
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t q = 0; q < n; ++q) {
            int32_t* o = out + ((size_t)r * n + q) * 3;
            if (r == q) { o[0] = o[1] = o[2] = 0; continue; }
            HostWave w{P, G[r].rview(), G[q].qview(), G[r].iv};
            PairMachine<HostWave> m(w, P, G[r].T, G[q].D);
            int res[3];
            m.run(res);
            o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
        }
    return 0;
}

// One pair by several segments (lzani_core.h: SplitStart / run_checkpoint / run_segment / split_stitch), cuts every `seglen`
// query positions: out as model_all2all; stats[0] = pairs stitched, [1] = pairs the stitch voided (scanned whole instead),
// [2] = segments in all, [3] = segments whose work the chain of hand-overs skipped.
int model_split_all2all(uint32_t n, const uint8_t* const* codes, const uint32_t* len, const int32_t* p8, int32_t seglen, int32_t* out, int64_t* stats)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P) || seglen < 1) return -1;
    uint32_t maxL = 0;
    for (uint32_t i = 0; i < n; ++i) maxL = std::max(maxL, len[i]);
    IndexGeom geo = index_geometry(ref_text_len((int)maxL, P.mrd), P.mal);
    std::vector<Genome> G(n);
    for (uint32_t i = 0; i < n; ++i) { pack_genome(G[i], codes[i], (int)len[i], P); build_index(G[i], P, geo); }
    for (int k = 0; k < 4; ++k) stats[k] = 0;
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t q = 0; q < n; ++q) {
            int32_t* o = out + ((size_t)r * n + q) * 3;
            if (r == q) { o[0] = o[1] = o[2] = 0; continue; }
            const int D = G[q].D, ncut = std::max(1, (D + seglen - 1) / seglen);
            std::vector<SplitStart> cuts(ncut);
            std::vector<SplitOut> segs(ncut);
            for (int j = 1; j < ncut; ++j) {
                HostWave w{P, G[r].rview(), G[q].qview(), G[r].iv};
                PairMachine<HostWave> m(w, P, G[r].T, D);
                m.run_checkpoint(j * seglen, &cuts[j]);
            }
            cuts[0] = SplitStart{0, 0, -1, 0, 0, 0};
            for (int j = 0; j < ncut; ++j) {
                HostWave w{P, G[r].rview(), G[q].qview(), G[r].iv};
                PairMachine<HostWave> m(w, P, G[r].T, D);
                if (j > 0 && cuts[j].i < 0) { segs[j] = SplitOut{}; segs[j].first = 2; continue; }
                m.run_segment(j, j ? &cuts[j] : nullptr, cuts.data(), ncut, &segs[j]);
            }
            int res[3];
            stats[2] += ncut;
            // a void segment: its cut is disabled and the segment that handed over to it runs again (as the device does, round by round)
            bool ok = false;
            for (int round = 0; round < 8 && !ok; ++round) {
                int at = -1, from = -1;
                ok = split_stitch(cuts.data(), segs.data(), ncut, P.reg, res, &at, &from);
                if (ok || at < 0 || from < 0) break;
                cuts[at].i = -1;
                HostWave w{P, G[r].rview(), G[q].qview(), G[r].iv};
                PairMachine<HostWave> m(w, P, G[r].T, D);
                m.resume_segment(from, cuts.data(), ncut, &segs[from]);
                stats[1] += 1;                       // (counted: segments run again)
            }
            if (ok) {
                stats[0] += 1;
                int used = 0;
                for (int j = 0; j >= 0 && j < ncut; j = segs[j].stop) { ++used; if (segs[j].stop < 0) break; }
                stats[3] += ncut - used;
            } else {
                stats[3] += 1000000;                 // (counted apart: pairs scanned whole after all)
                HostWave w{P, G[r].rview(), G[q].qview(), G[r].iv};
                PairMachine<HostWave> m(w, P, G[r].T, D);
                m.run(res);
            }
            o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
        }
    return 0;
}

// The anchor-queue formulation of the device's find_event (queue_wave.h) on the host: out as model_all2all.
int model_queue_all2all(uint32_t n, const uint8_t* const* codes, const uint32_t* len, const int32_t* p8, int32_t* out)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P) || P.mal > 15 || P.msl > 15) return -1;
    uint32_t maxL = 0;
    for (uint32_t i = 0; i < n; ++i) maxL = std::max(maxL, len[i]);
    const int Tmax = ref_text_len((int)maxL, P.mrd);
    IndexGeom geo = index_geometry(Tmax, P.mal);
    if (geo.tagmask != (u32)lowmask(geo.kb - geo.dirbits)) return -1;        // the queue needs exact tags
    std::vector<Genome> G(n);
    std::vector<QueueWaveTables> tabs(n);
    for (uint32_t i = 0; i < n; ++i) {
        pack_genome(G[i], codes[i], (int)len[i], P); build_index(G[i], P, geo); build_seed_index(G[i], P, Tmax);
        const size_t nb = (size_t)1 << geo.dirbits;
        tabs[i].bk.assign(4 * nb, BK_EMPTY);
        for (size_t b = 0; b < nb; ++b) {
            const u32 s = G[i].dirz[b], e = G[i].dirz[b + 1];
            for (u32 k = 0; k < 4 && s + k < e; ++k) tabs[i].bk[4 * b + k] = G[i].ent[s + k];
            if (e - s > 4) tabs[i].bk[4 * b + 3] = BK_OVERFLOW;
        }
    }
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t q = 0; q < n; ++q) {
            int32_t* o = out + ((size_t)r * n + q) * 3;
            if (r == q) { o[0] = o[1] = o[2] = 0; continue; }
            HostWave base{P, G[r].rview(), G[q].qview(), G[r].iv};
            QueueWave<HostWave> w(base, tabs[r], G[q].kmL.data(), G[q].D - P.msl);
            PairMachine<QueueWave<HostWave>> m(w, P, G[r].T, G[q].D);
            int res[3];
            m.run(res);
            o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
        }
    return 0;
}

// The lane-serial policy (thread-per-pair formulation) on the host: out as model_all2all.
// use_words != 0 feeds it the per-position k-mer words as the device does.
int model_lane_all2all(uint32_t n, const uint8_t* const* codes, const uint32_t* len, const int32_t* p8, int use_words, int32_t* out)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P)) return -1;
    uint32_t maxL = 0;
    for (uint32_t i = 0; i < n; ++i) maxL = std::max(maxL, len[i]);
    const int Tmax = ref_text_len((int)maxL, P.mrd);
    IndexGeom geo = index_geometry(Tmax, P.mal);
    std::vector<Genome> G(n);
    for (uint32_t i = 0; i < n; ++i) { pack_genome(G[i], codes[i], (int)len[i], P); build_index(G[i], P, geo); build_seed_index(G[i], P, Tmax); }
    const bool words = use_words && P.mal <= 15 && P.msl <= 15;
    for (uint32_t r = 0; r < n; ++r)
        for (uint32_t q = 0; q < n; ++q) {
            int32_t* o = out + ((size_t)r * n + q) * 3;
            if (r == q) { o[0] = o[1] = o[2] = 0; continue; }
            LaneWave w{P, G[r].rview(), G[q].qview(), G[r].iv, G[r].sv, words ? G[q].kmL.data() : nullptr, words ? G[q].kmS.data() : nullptr};
            PairMachine<LaneWave> m(w, P, G[r].T, G[q].D);
            int res[3];
            m.run(res);
            o[0] = res[0]; o[1] = res[1]; o[2] = res[2];
        }
    return 0;
}

// calc_regions through the streaming machine (ALN instantiation): regions of parse(query, ref) in
// emission order, 6 ints each (ref_start, ref_end, seq_start, seq_end, num_matches, num_mismatches).
int model_pair_regions(const uint8_t* ref, uint32_t ref_len, const uint8_t* qry, uint32_t qry_len, const int32_t* p8,
                       int32_t* res, int32_t* regions, uint32_t max_regions, uint32_t* n_regions)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P)) return -1;
    IndexGeom geo = index_geometry(ref_text_len((int)std::max(ref_len, qry_len), P.mrd), P.mal);
    Genome R, Q;
    pack_genome(R, ref, (int)ref_len, P); build_index(R, P, geo);
    pack_genome(Q, qry, (int)qry_len, P);
    std::vector<RegionCoords> regs;
    HostWave w{P, R.rview(), Q.qview(), R.iv};
    w.regs = &regs;
    PairMachine<HostWave, true> m(w, P, R.T, Q.D);
    m.run(res);
    *n_regions = (uint32_t)regs.size();
    for (uint32_t k = 0; k < regs.size() && k < max_regions; ++k) {
        int32_t* o = regions + 6 * k;
        o[0] = regs[k].ref_start; o[1] = regs[k].ref_end; o[2] = regs[k].seq_start; o[3] = regs[k].seq_end;
        o[4] = regs[k].nm; o[5] = regs[k].nmm;
    }
    return 0;
}

// Packed text + index of one genome exactly as the device builds them (for the GPU index test).
int model_index(const uint8_t* codes, uint32_t len, uint32_t max_len, const int32_t* p8,
                uint64_t* t2, uint64_t* nm, uint32_t* dirz, uint32_t* ent, uint32_t* n_ent)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P)) return -1;
    IndexGeom geo = index_geometry(ref_text_len((int)max_len, P.mrd), P.mal);
    Genome g;
    pack_genome(g, codes, (int)len, P);
    build_index(g, P, geo);
    if (t2) memcpy(t2, g.t2.data(), g.t2.size() * 8);
    if (nm) memcpy(nm, g.nm.data(), g.nm.size() * 8);
    if (dirz) memcpy(dirz, g.dirz.data(), g.dirz.size() * 4);
    if (ent) memcpy(ent, g.ent.data(), (g.ent.size() - 1) * 4);
    if (n_ent) *n_ent = (uint32_t)(g.ent.size() - 1);
    return 0;
}

// The null-extension record of every (query position, reference position, length) triple given, by the 16-symbol form
// refill uses for aw <= 15 and by the 32-symbol form: returns the number of triples whose records differ (0 expected).
int model_ext_records_agree(const uint8_t* rcodes, uint32_t rlen, const uint8_t* qcodes, uint32_t qlen, const int32_t* p8,
                            uint32_t n, const int32_t* qp, const int32_t* rp, const int32_t* al)
{
    Params P{p8[0], p8[1], p8[2], p8[3], p8[4], p8[5], p8[6], p8[7]};
    if (!params_supported(P)) return -1;
    Genome gr, gq;
    pack_genome(gr, rcodes, (int)rlen, P);
    pack_genome(gq, qcodes, (int)qlen, P);
    const TextView R = gr.rview(), Q = gq.qview();
    int bad = 0;
    for (uint32_t k = 0; k < n; ++k)
        bad += null_ext_record(P, R, Q, qp[k], rp[k], al[k], true) != null_ext_record(P, R, Q, qp[k], rp[k], al[k], false);
    return bad;
}

}  // extern "C"
