/*
 * lzani_oracle.c -- TEST INFRASTRUCTURE ONLY (see lzani_oracle.h).
 *
 * Literal CPU restatement of the LZ-ANI pair path, written from the behaviour of
 * /root/reference/src/parser.cpp (cited per function).  It keeps the explicit factor
 * list (v_parsing) so calc_stats / calc_regions can be restated one to one.  The two
 * reference indexes (ht_long open addressing, ht_short counting sort) are replaced by
 * exact k-mer chains in ascending position order: the hash geometry of the reference
 * is not observable, only the candidate sets and their order are.
 */
#include "lzani_oracle.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

enum { FL_CLOSE = 1, FL_DISTANT = 2, FL_LIT = 4 };
enum { SYM_N_REF = 4, SYM_N_QRY = 5 };

struct lzo_ref {
    lzo_params p;
    int32_t T;          /* |R| = 2L + 3*mrd            (parser.cpp:18-24) */
    uint8_t *text;      /* R, one symbol per byte, N = 4 */
    int64_t *km_long;   /* mal-mer at each position or -1 (parser.cpp:53-103) */
    int64_t *km_short;  /* msl-mer at each position or -1 */
    int32_t *head_long, *next_long;   /* exact chains, ascending position */
    int32_t *head_short, *next_short;
    uint32_t mask_long, mask_short;
};

void lzo_default_params(lzo_params *p)
{
    p->mal = 11; p->msl = 7; p->mrd = 40; p->mqd = 40;
    p->reg = 35; p->aw = 15; p->am = 7; p->ar = 3;
}

static uint64_t mix64(uint64_t x)
{
    x ^= x >> 31; x *= 0x7fb5d329728ea185ULL;
    x ^= x >> 27; x *= 0x81dadef4bc2dd44dULL;
    x ^= x >> 33;
    return x;
}

/* prepare_kmers (parser.cpp:53-103): value of the k-mer starting at j, -1 if the window
 * holds a symbol >= 4 or runs past the end. */
static void make_kmers(const uint8_t *s, int n, int k, int64_t *out)
{
    uint64_t mask = (k >= 32) ? ~0ULL : ((1ULL << (2 * k)) - 1);
    uint64_t v = 0;
    int run = 0;
    for (int j = 0; j < n; ++j) out[j] = -1;
    for (int j = 0; j < n; ++j) {
        if (s[j] >= 4) { run = 0; v = 0; }
        else { v = ((v << 2) | s[j]) & mask; ++run; }
        if (run >= k) out[j + 1 - k] = (int64_t)v;
    }
}

static uint32_t pow2_at_least(uint32_t x)
{
    uint32_t r = 16;
    while (r < x) r <<= 1;
    return r;
}

static void make_chains(const int64_t *km, int n, int32_t **head, int32_t **next, uint32_t *mask)
{
    uint32_t size = pow2_at_least((uint32_t)n * 2u + 16u);
    *mask = size - 1;
    *head = (int32_t *)malloc(sizeof(int32_t) * size);
    *next = (int32_t *)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    for (uint32_t i = 0; i < size; ++i) (*head)[i] = -1;
    for (int j = n - 1; j >= 0; --j) {       /* descending, so every chain ascends */
        if (km[j] < 0) { (*next)[j] = -1; continue; }
        uint32_t h = (uint32_t)mix64((uint64_t)km[j]) & *mask;
        (*next)[j] = (*head)[h];
        (*head)[h] = j;
    }
}

lzo_ref *lzo_prepare_reference(const uint8_t *codes, uint32_t len, const lzo_params *p)
{
    lzo_ref *r = (lzo_ref *)calloc(1, sizeof(lzo_ref));
    r->p = *p;
    int L = (int)len, mrd = p->mrd;
    r->T = 2 * L + 3 * mrd;
    r->text = (uint8_t *)malloc((size_t)r->T + 1);
    int o = 0;
    for (int j = 0; j < L; ++j) r->text[o++] = codes[j] < 4 ? codes[j] : SYM_N_REF;
    for (int j = 0; j < 2 * mrd; ++j) r->text[o++] = SYM_N_REF;
    for (int j = L - 1; j >= 0; --j)          /* append_rc (parser.h:79-96) */
        r->text[o++] = codes[j] < 4 ? (uint8_t)(3 - codes[j]) : SYM_N_REF;
    for (int j = 0; j < mrd; ++j) r->text[o++] = SYM_N_REF;

    r->km_long = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r->T + 1));
    r->km_short = (int64_t *)malloc(sizeof(int64_t) * (size_t)(r->T + 1));
    make_kmers(r->text, r->T, p->mal, r->km_long);
    make_kmers(r->text, r->T, p->msl, r->km_short);
    make_chains(r->km_long, r->T, &r->head_long, &r->next_long, &r->mask_long);
    make_chains(r->km_short, r->T, &r->head_short, &r->next_short, &r->mask_short);
    return r;
}

void lzo_free_reference(lzo_ref *r)
{
    if (!r) return;
    free(r->text); free(r->km_long); free(r->km_short);
    free(r->head_long); free(r->next_long); free(r->head_short); free(r->next_short);
    free(r);
}

/* ---- per-query working state ------------------------------------------------------ */
typedef struct {
    const lzo_ref *r;
    const uint8_t *R; int T;
    uint8_t *Q; int D;
    int64_t *qk_long, *qk_short;
    lzo_factor *f; uint32_t nf, cap;
} work_t;

static void push(work_t *w, int data_pos, int flag, int offset, int len)
{
    if (w->nf == w->cap) {
        w->cap = w->cap ? w->cap * 2 : 1024;
        w->f = (lzo_factor *)realloc(w->f, sizeof(lzo_factor) * w->cap);
    }
    lzo_factor x = { data_pos, flag, offset, len };
    w->f[w->nf++] = x;
}

/* equal_len (parser.cpp:192-207); note it returns `start` even when the bound is lower. */
static int equal_len(const work_t *w, int rp, int qp, int start)
{
    int bound = w->T - rp < w->D - qp ? w->T - rp : w->D - qp;
    int n = start;
    while (n < bound && w->R[rp + n] == w->Q[qp + n]) ++n;
    return n;
}

/* best_anchor (parser.cpp:514-531 == 585-602): longest >= mal, first (smallest pos) wins ties. */
static void best_anchor(const work_t *w, int qp, int *pos, int *len)
{
    *pos = 0; *len = 0;
    int64_t key = w->qk_long[qp];
    if (key < 0) return;
    const lzo_ref *r = w->r;
    for (int32_t c = r->head_long[(uint32_t)mix64((uint64_t)key) & r->mask_long]; c >= 0; c = r->next_long[c]) {
        if (r->km_long[c] != key) continue;
        int m = equal_len(w, c, qp, 0);
        if (m < r->p.mal) continue;
        if (m > *len) { *len = m; *pos = c; }
    }
}

/* R[] read that tolerates the positions the reference reads out of bounds only under
 * non-default parameters (mqd > mrd near the end of R): treated as "never equal". */
static int req(const work_t *w, int rp, int qp)
{
    if (rp < 0 || rp >= w->T || qp < 0 || qp >= w->D) return 0;
    return w->R[rp] == w->Q[qp];
}

/* compare_ranges (parser.cpp:210-248) */
static void compare_ranges(work_t *w, int qs, int rs, int len, int backward)
{
    int flag = backward ? FL_DISTANT : FL_CLOSE;
    int run = 0, matching = 0;
    for (int j = 0; j < len; ++j) {
        int eq = req(w, rs + j, qs + j);
        if (eq) {
            if (matching) ++run;
            else {
                if (run) push(w, qs + j - run, FL_LIT, 0, run);
                run = 1; matching = 1;
            }
        } else {
            if (matching) {
                push(w, qs + j - run, flag, rs + j - run, run);
                run = 1; matching = 0; flag = FL_CLOSE;
            } else ++run;
        }
    }
    if (matching) push(w, qs + len - run, flag, rs + len - run, run);
    else if (run) push(w, qs + len - run, FL_LIT, 0, run);
}

/* compare_ranges_both_ways (parser.cpp:251-374) */
static void gap_fill(work_t *w, int qs, int r_left, int r_right_end, int len)
{
    int to_scan = (r_right_end < r_left) ? len : (r_right_end - r_left < len ? r_right_end - r_left : len);
    int *lc = (int *)malloc(sizeof(int) * (size_t)(2 * to_scan + 2) * 2);
    int *lm = lc + (to_scan + 1), *rc = lm + (to_scan + 1), *rm = rc + (to_scan + 1);
    int c = 0;
    lc[0] = 0; lm[0] = 0;
    for (int j = 0; j < to_scan; ++j) {
        int eq = req(w, r_left + j, qs + j);
        c += eq; lc[j + 1] = c; lm[j + 1] = eq;
    }
    c = 0;
    rc[0] = 0; rm[0] = 0;
    int lim = to_scan < r_right_end ? to_scan : r_right_end;
    for (int j = 1; j <= to_scan; ++j) {
        if (j <= lim) {
            int eq = req(w, r_right_end - j, qs + len - j);
            c += eq; rc[j] = c; rm[j] = eq;
        } else { rc[j] = 0; rm[j] = 0; }     /* resize(to_scan+1, (0,false)), line 300 */
    }
    int best = 0, best_split = 0;
    for (int s = 0; s <= to_scan; ++s) {
        int v = lc[s] + rc[to_scan - s];
        if (v >= best) { best = v; best_split = s; }
    }
    /* left */
    if (best_split > 0) {
        push(w, qs, lm[1] ? FL_CLOSE : FL_LIT, lm[1] ? r_left : 0, 1);
        for (int s = 2; s <= best_split; ++s) {
            int fl = lm[s] ? FL_CLOSE : FL_LIT;
            if (w->f[w->nf - 1].flag == fl) w->f[w->nf - 1].len++;
            else push(w, qs + s - 1, fl, lm[s] ? r_left + s - 1 : 0, 1);
        }
    }
    /* middle */
    if (to_scan < len) {
        if (best_split > 0 && w->f[w->nf - 1].flag == FL_LIT) w->f[w->nf - 1].len += len - to_scan;
        else push(w, qs + best_split, FL_LIT, 0, len - to_scan);
    }
    /* right */
    if (best_split < to_scan) {
        int shift = len - to_scan;
        int from_right = to_scan - best_split;
        int qp = qs + best_split + shift;
        int eq = rm[from_right];
        if (!eq && (best_split > 0 || shift > 0) && w->f[w->nf - 1].flag == FL_LIT)
            w->f[w->nf - 1].len++;
        else { push(w, qp, eq ? FL_CLOSE : FL_LIT, eq ? r_right_end - from_right : 0, 1); ++qp; }
        /* NB the reference advances data_p only on the emplace path (line 358) */
        for (int j = from_right - 1; j > 0; --j, ++qp) {
            eq = rm[j];
            int fl = eq ? FL_CLOSE : FL_LIT;
            if (w->f[w->nf - 1].flag == fl) w->f[w->nf - 1].len++;
            else push(w, qp, fl, eq ? r_right_end - j : 0, 1);
        }
    }
    free(lc);
}

/* try_extend_forward (parser.cpp:377-409) */
static int extend_forward(const work_t *w, int qs, int rs)
{
    const lzo_params *p = &w->r->p;
    int *win = (int *)calloc((size_t)(p->aw > 0 ? p->aw : 1), sizeof(int));
    int mism = 0, last = 0, run = p->ar, e;
    for (e = 0; qs + e < w->D && rs + e < w->T; ++e) {
        int bad = w->Q[qs + e] != w->R[rs + e];
        mism -= win[e % p->aw];
        win[e % p->aw] = bad;
        mism += bad;
        if (!bad) { if (++run >= p->ar) last = e + 1; }
        else run = 0;
        if (mism > p->am) break;
    }
    free(win);
    return last;
}

/* try_extend_backward (parser.cpp:412-441) */
static int extend_backward(const work_t *w, int qs, int rs, int max_len)
{
    const lzo_params *p = &w->r->p;
    int *win = (int *)calloc((size_t)(p->aw > 0 ? p->aw : 1), sizeof(int));
    int mism = 0, last = 0, run = p->ar, e;
    for (e = 0; qs - e > 0 && rs - e > 0 && e < max_len; ++e) {
        int bad = w->Q[qs - e - 1] != w->R[rs - e - 1];
        mism -= win[e % p->aw];
        win[e % p->aw] = bad;
        mism += bad;
        if (!bad) { if (++run >= p->ar) last = e + 1; }
        else run = 0;
        if (mism > p->am) break;
    }
    free(win);
    return last;
}

/* prob_len (parser.h:134-172): exactly 4^-len */
static double prob_len(int len)
{
    double v = 1.0;
    if (len > 600) return 0.0;
    for (int j = 0; j < len; ++j) v *= 0.25;   /* exact: powers of two, gradual underflow */
    return v;
}

/* ipow<double> (parser.h:174-188) */
static double ipow_u32(double base, uint32_t e)
{
    double r = 1.0;
    while (e) {
        if (e & 1u) r *= base;
        base *= base;
        e /= 2;
    }
    return r;
}

/* parse (parser.cpp:482-716) */
static void parse(work_t *w)
{
    const lzo_ref *r = w->r;
    const lzo_params *p = &r->p;
    const int D = w->D;
    int ref_pred = -D, lit = 0, i;
    int prev_rs = -1, prev_re = 0;
    w->nf = 0;

    for (i = 0; i + p->msl < D;) {
        int best_pos = 0, best_len = 0;

        if (ref_pred < 0) {
            best_anchor(w, i, &best_pos, &best_len);
        } else {
            int64_t sk = w->qk_short[i];
            if (sk >= 0) {
                int lo = ref_pred - lit, hi = ref_pred + p->mrd;
                for (int32_t c = r->head_short[(uint32_t)mix64((uint64_t)sk) & r->mask_short]; c >= 0; c = r->next_short[c]) {
                    if (c < lo) continue;
                    if (c >= hi) break;
                    if (r->km_short[c] != sk) continue;
                    int m = equal_len(w, c, i, p->msl);
                    if (m >= best_len) {
                        if (m == best_len) {
                            if (abs(c - ref_pred) < abs(best_pos - ref_pred)) best_pos = c;
                        } else { best_len = m; best_pos = c; }
                    }
                }
            }
            int ap, al;
            best_anchor(w, i, &ap, &al);
            if (ap) {
                if (!best_pos) { best_pos = ap; best_len = al; }
                else {
                    double anchor_prob = ipow_u32(1 - prob_len(al), (uint32_t)(int)(2 * ((uint64_t)w->T + 1 - (uint64_t)(int64_t)al)));
                    double close_prob = ipow_u32(1 - prob_len(best_len), (uint32_t)(lit + p->mrd + 1 - best_len));
                    if (anchor_prob > close_prob) { best_pos = ap; best_len = al; }
                }
            }
        }

        if (best_len >= p->msl) {
            int flag = FL_DISTANT;
            if (ref_pred >= 0 && abs(best_pos - ref_pred) <= p->mrd) {
                gap_fill(w, i - lit, ref_pred - lit, best_pos + best_len, lit);
                push(w, i, FL_CLOSE, best_pos, best_len);
            } else {
                if (lit) push(w, i - lit, FL_LIT, 0, lit);
                if (prev_rs >= 0 && !(prev_re - prev_rs >= p->reg)) {   /* eval_region, 446-449 */
                    while (w->nf && w->f[w->nf - 1].data_pos >= prev_rs) --w->nf;
                    int run = i - prev_rs;
                    while (w->nf && w->f[w->nf - 1].flag == FL_LIT) { run += w->f[w->nf - 1].len; --w->nf; }
                    push(w, i - run, FL_LIT, 0, run);
                    prev_rs = -1;
                }
                if (w->nf && w->f[w->nf - 1].flag == FL_LIT) {
                    int b = extend_backward(w, i, best_pos, w->f[w->nf - 1].len);
                    if (b) {
                        w->f[w->nf - 1].len -= b;
                        if (w->f[w->nf - 1].len == 0) --w->nf;
                        compare_ranges(w, i - b, best_pos - b, b, 1);
                        flag = FL_CLOSE;
                        prev_rs = i - b;
                    }
                }
                push(w, i, flag, best_pos, best_len);
                if (flag == FL_DISTANT) prev_rs = i;
                if (prev_rs < 0)                                        /* 678-684 (unreachable) */
                    for (int j = (int)w->nf - 1; j >= 0; --j)
                        if (w->f[j].flag == FL_DISTANT) { prev_rs = w->f[j].data_pos; break; }
            }
            i += best_len;
            ref_pred = best_pos + best_len;
            lit = 0;
            int e = extend_forward(w, i, ref_pred);
            compare_ranges(w, i, ref_pred, e, 0);
            i += e;
            ref_pred += e;
            prev_re = i;
        } else {
            ++i; ++ref_pred; ++lit;
        }
        if (lit > p->mqd) ref_pred = -D;
    }

    if (ref_pred < 0) push(w, i - lit, FL_LIT, 0, lit + (D - i));
    else    /* tail (parser.cpp:713): reference start is r_end - msl (quirk Q3) */
        compare_ranges(w, i - lit, ref_pred - lit - p->msl, lit + (D - i), 0);
}

/* calc_stats (parser.cpp:734-783) */
static void calc_stats(const work_t *w, lzo_result *out)
{
    int reg = w->r->p.reg;
    int cl = 0, clit = 0, nl = 0;
    int tm = 0, tl = 0, tc = 0;
    for (uint32_t k = 0; k < w->nf; ++k) {
        const lzo_factor *x = &w->f[k];
        if (x->flag == FL_DISTANT) {
            if (cl && cl + clit >= reg) { tm += cl; tl += clit; ++tc; }
            cl = x->len; clit = 0; nl = 0;
        } else if (x->flag == FL_CLOSE) { cl += x->len; clit += nl; nl = 0; }
        else nl += x->len;
    }
    if (cl && cl + clit >= reg) { tm += cl; tl += clit; ++tc; }
    out->sym_in_matches = tm; out->sym_in_literals = tl; out->no_components = tc;
}

static void reg_clear(lzo_region *g)
{
    g->ref_start = g->ref_end = g->seq_start = g->seq_end = -1;
    g->num_matches = g->num_mismatches = 0;
}
static void reg_touch(lzo_region *g, const lzo_factor *x)
{
    if (g->seq_start < 0 || x->data_pos < g->seq_start) g->seq_start = x->data_pos;
    if (g->seq_end < 0 || x->data_pos + x->len > g->seq_end) g->seq_end = x->data_pos + x->len;
    if (g->ref_start < 0 || x->offset < g->ref_start) g->ref_start = x->offset;
    if (g->ref_end < 0 || x->offset + x->len > g->ref_end) g->ref_end = x->offset + x->len;
    g->num_matches += x->len;
}
static int reg_cmp(const void *a, const void *b)
{
    const lzo_region *x = (const lzo_region *)a, *y = (const lzo_region *)b;
    int lx = x->seq_end - x->seq_start, ly = y->seq_end - y->seq_start;
    if (lx != ly) return lx > ly ? -1 : 1;
    return (x->seq_start > y->seq_start) - (x->seq_start < y->seq_start);
}

/* calc_regions (parser.cpp:786-837) */
static uint32_t calc_regions(const work_t *w, lzo_region *out, uint32_t max_out)
{
    int reg = w->r->p.reg;
    uint32_t n = 0, cap = 64;
    lzo_region *v = (lzo_region *)malloc(sizeof(lzo_region) * cap);
    lzo_region cur; reg_clear(&cur);
    int buf = 0;
    for (uint32_t k = 0; k <= w->nf; ++k) {
        const lzo_factor *x = k < w->nf ? &w->f[k] : NULL;
        if (!x || x->flag == FL_DISTANT) {
            if (cur.seq_end - cur.seq_start >= reg) {
                if (n == cap) { cap *= 2; v = (lzo_region *)realloc(v, sizeof(lzo_region) * cap); }
                v[n++] = cur;
            }
            if (!x) break;
            reg_clear(&cur);
            reg_touch(&cur, x);
            buf = 0;
        } else if (x->flag == FL_CLOSE) {
            cur.ref_end += buf; cur.seq_end += buf;     /* extend_region */
            cur.num_mismatches += buf;
            buf = 0;
            reg_touch(&cur, x);
        } else buf += x->len;
    }
    qsort(v, n, sizeof(lzo_region), reg_cmp);   /* keys are unique per region: seq_start differs */
    for (uint32_t k = 0; k < n && k < max_out; ++k) out[k] = v[k];
    free(v);
    return n;
}

int lzo_query(const lzo_ref *r, const uint8_t *codes, uint32_t len, lzo_result *out,
              lzo_region *regions, uint32_t max_regions, uint32_t *n_regions,
              lzo_factor *factors, uint32_t max_factors, uint32_t *n_factors)
{
    work_t w;
    memset(&w, 0, sizeof w);
    w.r = r; w.R = r->text; w.T = r->T;
    w.D = (int)len + r->p.mrd;                       /* prepare_data (parser.cpp:37-50) */
    w.Q = (uint8_t *)malloc((size_t)w.D + 1);
    for (uint32_t j = 0; j < len; ++j) w.Q[j] = codes[j] < 4 ? codes[j] : SYM_N_QRY;
    for (int j = (int)len; j < w.D; ++j) w.Q[j] = SYM_N_QRY;
    w.qk_long = (int64_t *)malloc(sizeof(int64_t) * (size_t)(w.D + 1));
    w.qk_short = (int64_t *)malloc(sizeof(int64_t) * (size_t)(w.D + 1));
    make_kmers(w.Q, w.D, r->p.mal, w.qk_long);
    make_kmers(w.Q, w.D, r->p.msl, w.qk_short);

    parse(&w);
    calc_stats(&w, out);
    if (regions || n_regions) {
        uint32_t n = calc_regions(&w, regions, regions ? max_regions : 0);
        if (n_regions) *n_regions = n;
    }
    if (n_factors) *n_factors = w.nf;
    if (factors)
        for (uint32_t k = 0; k < w.nf && k < max_factors; ++k) factors[k] = w.f[k];

    free(w.Q); free(w.qk_long); free(w.qk_short); free(w.f);
    return 0;
}

int lzo_pair(const uint8_t *ref, uint32_t ref_len, const uint8_t *qry, uint32_t qry_len,
             const lzo_params *p, lzo_result *out)
{
    lzo_ref *r = lzo_prepare_reference(ref, ref_len, p);
    int rc = lzo_query(r, qry, qry_len, out, NULL, 0, NULL, NULL, 0, NULL);
    lzo_free_reference(r);
    return rc;
}

/* ---- do_matching restated (lz_matcher.cpp:172-277) -------------------------------- */
typedef struct {
    uint32_t n;
    const uint8_t *const *codes;
    const uint32_t *len;
    const lzo_params *p;
    lzo_result *out;
    volatile uint32_t *next;
} a2a_t;

static void *a2a_worker(void *arg)
{
    a2a_t *a = (a2a_t *)arg;
    for (;;) {
        uint32_t ref = __sync_fetch_and_add(a->next, 1u);
        if (ref >= a->n) break;
        lzo_ref *r = lzo_prepare_reference(a->codes[ref], a->len[ref], a->p);
        for (uint32_t q = 0; q < a->n; ++q) {
            lzo_result *o = &a->out[(size_t)ref * a->n + q];
            if (q == ref) { memset(o, 0, sizeof *o); continue; }
            lzo_query(r, a->codes[q], a->len[q], o, NULL, 0, NULL, NULL, 0, NULL);
        }
        lzo_free_reference(r);
    }
    return NULL;
}

int lzo_all2all(uint32_t n, const uint8_t *const *codes, const uint32_t *len,
                const lzo_params *p, uint32_t n_threads, lzo_result *out)
{
    if (n_threads == 0) n_threads = 1;
    volatile uint32_t next = 0;
    a2a_t a = { n, codes, len, p, out, &next };
    pthread_t *th = (pthread_t *)malloc(sizeof(pthread_t) * n_threads);
    for (uint32_t t = 0; t < n_threads; ++t) pthread_create(&th[t], NULL, a2a_worker, &a);
    for (uint32_t t = 0; t < n_threads; ++t) pthread_join(th[t], NULL);
    free(th);
    return 0;
}
