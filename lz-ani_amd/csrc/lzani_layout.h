// lzani_layout.h -- HBM data layout shared by the host side of the C-ABI, the kernels and the
// test model: packed text sizes and the geometry of the per-reference anchor index.
#pragma once
#include "lzani_core.h"

namespace lzani {

// |R| = fwd + 2*mrd N + RC + mrd N   (parser.cpp:18-24)
LZ_HD int ref_text_len(int L, int mrd) { return 2 * L + 3 * mrd; }

// u64 words of the N mask / the 2-bit text of a T-symbol text: whole 64-symbol blocks (one
// mask word + two text words each) plus two spare blocks that keep the unaligned 128-bit
// window reads (win2/winN) in bounds; spare symbols are flagged N.
LZ_HD size_t text_wordsN(int T) { return (size_t)((T + 63) >> 6) + 2; }
LZ_HD size_t text_words2(int T) { return 2 * text_wordsN(T); }

struct IndexGeom {
    int kb, dirbits, posbits;
    u32 tagmask;
};

LZ_HD int ceil_log2(u64 x) { int b = 0; while (((u64)1 << b) < x) ++b; return b; }

// One geometry for every reference of a context (sized for the longest text), so that a
// directory is a fixed-stride slab: dirz stride = 2^dirbits + 1 words.
LZ_HD IndexGeom index_geometry(int Tmax, int mal)
{
    IndexGeom g;
    g.kb = 2 * mal;
    g.posbits = imax(1, ceil_log2((u64)Tmax + 1));
    int d = ceil_log2((u64)imax(Tmax, 1));
    d = imax(8, imin(d, 26));
    g.dirbits = imin(d, g.kb);
    int tb = imin(g.kb - g.dirbits, 32 - g.posbits);
    g.tagmask = (u32)lowmask(tb);
    return g;
}

// Parameter envelope of the wave formulation (64-bit masks for the literal run, the
// approximate-extension window and the match run).  The reference accepts any ints
// (lz-ani.cpp:205-260); outside this envelope the C-ABI reports LZANI_ERR_PARAMS.
LZ_HD bool params_supported(const Params& P)
{
    return P.msl >= 1 && P.msl <= 32 && P.mal >= 1 && P.mal <= 32 && P.mrd >= 0 && P.mrd <= (1 << 20) &&
           P.mqd >= 0 && P.mqd <= 64 && P.aw >= 1 && P.aw <= 64 && P.ar <= 64 && P.am >= 0;
}

}  // namespace lzani
