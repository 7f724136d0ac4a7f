/*
 * tdc_oracle.c -- TEST INFRASTRUCTURE ONLY (see tdc_oracle.h).
 *
 * Plain C restatement of the tudocomp CPU path  lcpcomp(coder=huff, comp=arrays)  and of the
 * pieces it is built from.  Written from the reference's behaviour; every function cites the
 * reference file:line (relative to /root/reference/include/tudocomp/) it follows.
 *
 * The suffix array is built with an own SA-IS (induced sorting) -- the reference calls its vendored
 * divsufsort (util/divsufsort.hpp:46-279); the suffix array of a text is unique, so any correct
 * construction is bit-compatible (checked against a naive sort in tests/).
 *
 * Pinned against the reference's own known-answer tests and recorded outputs (tests/golden/) for lcpcomp(huff / arithmetic,
 * comp=arrays), lz78(gamma), the bit stream, Huffman tables and escaping.  PARITY UNPINNED (no recorded reference output
 * exists): SLECoder (sle_*), PLCPPeaksStrategy (orc_plcp_peaks), MaxLCPStrategy (orc_max_lcp) -- restated from the
 * reference's sources and checked by round trips / properties / an independent model only.
 */
#include "tdc_oracle.h"

#include <stdlib.h>
#include <string.h>
#include <time.h>

static double now_s(void) {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

void orc_free(void* p) { free(p); }

/* ------------------------------------------------------------------------------------------------
 * util.hpp:175-196  bits_for(0) = 1, else index of the highest set bit + 1
 * ---------------------------------------------------------------------------------------------- */
unsigned orc_bits_for(uint64_t v) {
    unsigned b = 0;
    if (v == 0) return 1;
    while (v) { ++b; v >>= 1; }
    return b;
}

/* ------------------------------------------------------------------------------------------------
 * Escaping: io/EscapeMap.hpp:10-24,39-64 (escape byte 0xFF, replacement pool starts at 0xFE),
 * io/RestrictedBuffer.hpp:43-74.  With escape set {0}: 0x00 -> FF FE, 0xFF -> FF FF; a 0 is appended.
 * KAT: test/tudocomp_tests.cpp:528-556.
 * ---------------------------------------------------------------------------------------------- */
size_t orc_escape(const uint8_t* in, size_t n, uint8_t* out) {
    size_t o = 0;
    for (size_t i = 0; i < n; ++i) {
        uint8_t c = in[i];
        if (c == 0x00)      { out[o++] = 0xFF; out[o++] = 0xFE; }
        else if (c == 0xFF) { out[o++] = 0xFF; out[o++] = 0xFF; }
        else                out[o++] = c;
    }
    out[o++] = 0;
    return o;
}

/* io/RestrictedIOStream.hpp:32-61 : un-escape and drop the final 0 */
size_t orc_unescape(const uint8_t* in, size_t n, uint8_t* out) {
    size_t o = 0;
    if (n && in[n - 1] == 0) --n;
    for (size_t i = 0; i < n; ++i) {
        uint8_t c = in[i];
        if (c == 0xFF && i + 1 < n) {
            uint8_t d = in[++i];
            out[o++] = (d == 0xFE) ? 0x00 : d;   /* FF FE -> 00, FF FF -> FF */
        } else out[o++] = c;
    }
    return o;
}

/* ------------------------------------------------------------------------------------------------
 * Suffix array by induced sorting (SA-IS, Nong/Zhang/Chan).  Requires T[n-1] to be the unique
 * smallest symbol -- guaranteed by the 0 sentinel (ds/TextDS.hpp:132-138, ds/SADivSufSort.hpp:20-25).
 * ---------------------------------------------------------------------------------------------- */
#define SAIS_CHR(i) (cs == 4 ? ((const int32_t*)T)[i] : (int32_t)((const uint8_t*)T)[i])
#define TGET(i)     ((t[(i) >> 3] >> ((i) & 7)) & 1)
#define TSET(i, b)  (t[(i) >> 3] = (uint8_t)((b) ? (t[(i) >> 3] | (1u << ((i) & 7))) : (t[(i) >> 3] & ~(1u << ((i) & 7)))))
#define IS_LMS(i)   ((i) > 0 && TGET(i) && !TGET((i) - 1))

static void sais_buckets(const void* T, int32_t* bkt, int32_t n, int32_t K, int cs, int end) {
    int32_t i, sum = 0;
    for (i = 0; i < K; ++i) bkt[i] = 0;
    for (i = 0; i < n; ++i) bkt[SAIS_CHR(i)]++;
    for (i = 0; i < K; ++i) { sum += bkt[i]; bkt[i] = end ? sum : sum - bkt[i]; }
}

static void sais_induce_l(const uint8_t* t, int32_t* SA, const void* T, int32_t* bkt, int32_t n, int32_t K, int cs) {
    sais_buckets(T, bkt, n, K, cs, 0);
    for (int32_t i = 0; i < n; ++i) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && !TGET(j)) SA[bkt[SAIS_CHR(j)]++] = j;
    }
}

static void sais_induce_s(const uint8_t* t, int32_t* SA, const void* T, int32_t* bkt, int32_t n, int32_t K, int cs) {
    sais_buckets(T, bkt, n, K, cs, 1);
    for (int32_t i = n - 1; i >= 0; --i) {
        int32_t j = SA[i] - 1;
        if (j >= 0 && TGET(j)) SA[--bkt[SAIS_CHR(j)]] = j;
    }
}

static int sais_main(const void* T, int32_t* SA, int32_t n, int32_t K, int cs) {
    if (n == 1) { SA[0] = 0; return 0; }
    uint8_t* t = (uint8_t*)calloc((size_t)n / 8 + 1, 1);
    int32_t* bkt = (int32_t*)malloc(sizeof(int32_t) * (size_t)K);
    if (!t || !bkt) { free(t); free(bkt); return -1; }
    int32_t i, j;

    /* classify: S-type = 1, L-type = 0 */
    TSET(n - 1, 1);
    TSET(n - 2, 0);
    for (i = n - 3; i >= 0; --i) {
        int32_t a = SAIS_CHR(i), b = SAIS_CHR(i + 1);
        TSET(i, (a < b || (a == b && TGET(i + 1))) ? 1 : 0);
    }

    /* stage 1: sort all LMS substrings */
    sais_buckets(T, bkt, n, K, cs, 1);
    for (i = 0; i < n; ++i) SA[i] = -1;
    for (i = 1; i < n; ++i) if (IS_LMS(i)) SA[--bkt[SAIS_CHR(i)]] = i;
    sais_induce_l(t, SA, T, bkt, n, K, cs);
    sais_induce_s(t, SA, T, bkt, n, K, cs);

    /* compact the sorted LMS substrings into SA[0..n1) */
    int32_t n1 = 0;
    for (i = 0; i < n; ++i) if (IS_LMS(SA[i])) SA[n1++] = SA[i];
    for (i = n1; i < n; ++i) SA[i] = -1;

    /* name them */
    int32_t name = 0, prev = -1;
    for (i = 0; i < n1; ++i) {
        int32_t pos = SA[i];
        int diff = 0;
        for (int32_t d = 0; d < n; ++d) {
            if (prev == -1 || SAIS_CHR(pos + d) != SAIS_CHR(prev + d) || TGET(pos + d) != TGET(prev + d)) { diff = 1; break; }
            if (d > 0 && (IS_LMS(pos + d) || IS_LMS(prev + d))) break;
        }
        if (diff) { ++name; prev = pos; }
        SA[n1 + pos / 2] = name - 1;
    }
    for (i = n - 1, j = n - 1; i >= n1; --i) if (SA[i] >= 0) SA[j--] = SA[i];

    /* stage 2: solve the reduced problem */
    int32_t* SA1 = SA;
    int32_t* s1 = SA + n - n1;
    if (name < n1) {
        if (sais_main(s1, SA1, n1, name, 4) != 0) { free(t); free(bkt); return -1; }
    } else {
        for (i = 0; i < n1; ++i) SA1[s1[i]] = i;
    }

    /* stage 3: induce the result */
    sais_buckets(T, bkt, n, K, cs, 1);
    for (i = 1, j = 0; i < n; ++i) if (IS_LMS(i)) s1[j++] = i;
    for (i = 0; i < n1; ++i) SA1[i] = s1[SA1[i]];
    for (i = n1; i < n; ++i) SA[i] = -1;
    for (i = n1 - 1; i >= 0; --i) {
        j = SA[i]; SA[i] = -1;
        SA[--bkt[SAIS_CHR(j)]] = j;
    }
    sais_induce_l(t, SA, T, bkt, n, K, cs);
    sais_induce_s(t, SA, T, bkt, n, K, cs);
    free(t); free(bkt);
    return 0;
}

int orc_suffix_array(const uint8_t* text, size_t n, uint32_t* sa) {
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFu) return -2;                 /* reference limit: n < 2^31 */
    if (text[n - 1] != 0) return -3;                 /* ds/TextDS.hpp:132-138 */
    return sais_main(text, (int32_t*)sa, (int32_t)n, 256, 1);
}

/* ds/ISAFromSA.hpp:37-39 */
void orc_isa(const uint32_t* sa, size_t n, uint32_t* isa) {
    for (size_t i = 0; i < n; ++i) isa[sa[i]] = (uint32_t)i;
}

/* ds/PhiFromSA.hpp:37-41 */
void orc_phi(const uint32_t* sa, size_t n, uint32_t* phi) {
    if (!n) return;
    uint32_t prev = sa[0];
    for (size_t i = 1; i < n; ++i) { phi[sa[i]] = prev; prev = sa[i]; }
    phi[sa[0]] = sa[n - 1];
}

/* ds/PLCPFromPhi.hpp:38-44 ; entry n-1 is never read by this path, we define it 0 */
uint32_t orc_plcp(const uint8_t* text, size_t n, const uint32_t* phi, uint32_t* plcp) {
    uint32_t max = 0, l = 0;
    if (!n) return 0;
    for (size_t i = 0; i + 1 < n; ++i) {
        const uint32_t phii = phi[i];
        while (text[i + l] == text[phii + l]) ++l;
        if (l > max) max = l;
        plcp[i] = l;
        if (l) --l;
    }
    plcp[n - 1] = 0;
    return max;
}

/* ds/LCPFromPLCP.hpp:43-47 */
void orc_lcp(const uint32_t* sa, const uint32_t* plcp, size_t n, uint32_t* lcp) {
    if (!n) return;
    lcp[0] = 0;
    for (size_t i = 1; i < n; ++i) lcp[i] = plcp[sa[i]];
}

/* ------------------------------------------------------------------------------------------------
 * compressors/lcpcomp/compress/ArraysComp.hpp:36-117
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t* a; size_t size, cap; } u32vec;
static int vec_push(u32vec* v, uint32_t x) {
    if (v->size == v->cap) {
        size_t nc = v->cap ? v->cap * 2 : 4;
        uint32_t* na = (uint32_t*)realloc(v->a, nc * sizeof(uint32_t));
        if (!na) return -1;
        v->a = na; v->cap = nc;
    }
    v->a[v->size++] = x;
    return 0;
}

size_t orc_arrays_comp(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n,
                       uint32_t maxlcp, uint32_t threshold, orc_factor** out) {
    *out = NULL;
    if (n == 0) return 0;
    if ((uint64_t)maxlcp + 1 <= threshold) return 0;                         /* :50 */
    const size_t cand_length = (size_t)maxlcp + 1 - threshold;               /* :51 */
    u32vec* cand = (u32vec*)calloc(cand_length, sizeof(u32vec));
    /* pre-size the lists (pure allocation detail) */
    for (size_t i = 1; i < n; ++i) if (lcp[i] >= threshold) cand[lcp[i] - threshold].cap++;
    for (size_t k = 0; k < cand_length; ++k)
        if (cand[k].cap) cand[k].a = (uint32_t*)malloc(cand[k].cap * sizeof(uint32_t));
    for (size_t i = 1; i < n; ++i) {                                         /* :54-58 fill candidates */
        if (lcp[i] < threshold) continue;
        u32vec* v = &cand[lcp[i] - threshold];
        v->a[v->size++] = (uint32_t)i;
    }
    size_t z = 0, zcap = 1024;
    orc_factor* F = (orc_factor*)malloc(zcap * sizeof(orc_factor));
    for (size_t L = maxlcp; L >= threshold && L > 0; --L) {                  /* :72 */
        u32vec* col = &cand[L - threshold];
        for (size_t k = 0; k < col->size; ++k) {                             /* :82 */
            const uint32_t index = col->a[k];
            const uint32_t v = lcp[index];
            if (v < L) {                                                     /* :85-89 lazy push-down */
                if (v < threshold) continue;
                vec_push(&cand[v - threshold], index);
                continue;
            }
            const uint32_t pos = sa[index], src = sa[index - 1], len = lcp[index];   /* :91-94 */
            if (z == zcap) { zcap *= 2; F = (orc_factor*)realloc(F, zcap * sizeof(orc_factor)); }
            F[z].pos = pos; F[z].src = src; F[z].len = len; ++z;              /* :96 */
            for (uint32_t j = 0; j < len; ++j) lcp[isa[pos + j]] = 0;         /* :99-101 */
            const uint32_t max_affect = len < pos ? len : pos;                /* :103 */
            for (uint32_t j = 0; j < max_affect; ++j) {                       /* :105-109 */
                const uint32_t ind = isa[pos - j - 1];
                if (j + 1 < lcp[ind]) lcp[ind] = j + 1;
            }
        }
        free(col->a); col->a = NULL; col->size = col->cap = 0;               /* :112-113 */
        if (L == 0) break;
    }
    for (size_t k = 0; k < cand_length; ++k) free(cand[k].a);
    free(cand);
    *out = F;
    return z;
}

/* ------------------------------------------------------------------------------------------------
 * lcpcomp::MaxLCPStrategy::factorize (compressors/lcpcomp/compress/MaxLCPStrategy.hpp:36-100) over MaxLCPSuffixList
 * (compressors/lcpcomp/MaxLCPSuffixList.hpp): a doubly linked list of SA indices in descending LCP order with an index
 * of the first entry per LCP value; the head is taken, the covered suffixes are removed and the ones in front of the
 * factor get their key decreased at once (remove + insert in front of their new level).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t* lcp; size_t undef, first, last, size;
    uint32_t* prev, *next, *lcp_index; uint8_t* contained; size_t maxlcp;
} mlsl;
static size_t mlsl_lookup(const mlsl* l, size_t lcp) {                       /* lookup_lcp_index :40-49 */
    size_t result = l->undef;
    while (lcp > 0 && result == l->undef) result = l->lcp_index[--lcp];
    return result;
}
static void mlsl_insert(mlsl* l, size_t i) {                                 /* :86-124 */
    const size_t lcp = l->lcp[i];
    const size_t pos = mlsl_lookup(l, lcp);
    if (pos == l->undef) {                                                   /* insert at end */
        if (l->last != l->undef) l->next[l->last] = (uint32_t)i;
        l->next[i] = (uint32_t)l->undef;
        l->prev[i] = (uint32_t)l->last;
        l->last = i;
    } else {                                                                 /* insert in front of pos */
        const size_t prev = l->prev[pos];
        l->prev[i] = (uint32_t)prev;
        l->next[i] = (uint32_t)pos;
        if (prev != l->undef) l->next[prev] = (uint32_t)i; else l->first = i;
        l->prev[pos] = (uint32_t)i;
    }
    l->lcp_index[lcp - 1] = (uint32_t)i;
    if (l->first == l->undef) l->first = i;
    l->contained[i] = 1;
    ++l->size;
}
static void mlsl_remove(mlsl* l, size_t i) {                                 /* :129-160 */
    if (l->prev[i] != l->undef) l->next[l->prev[i]] = l->next[i]; else l->first = l->next[i];
    if (l->next[i] != l->undef) l->prev[l->next[i]] = l->prev[i]; else l->last = l->prev[i];
    const size_t lcp = l->lcp[i];
    if (l->lcp_index[lcp - 1] == i) {
        const size_t k = l->next[i];
        if (k != l->undef && l->lcp[k] == lcp) l->lcp_index[lcp - 1] = (uint32_t)k;
        else l->lcp_index[lcp - 1] = (uint32_t)l->undef;
    }
    l->contained[i] = 0;
    --l->size;
}
/* ------------------------------------------------------------------------------------------------
 * lcpcomp::MaxHeapStrategy (compressors/lcpcomp/compress/MaxHeapStrategy.hpp:36-101) over tdc::ArrayMaxHeap
 * (ds/ArrayMaxHeap.hpp:14-241): a binary max-heap of suffix-array indices keyed by lcp[], with a back mapping.
 * Restated operation by operation, quirks included: remove() moves the LAST heap element into the hole and only ever
 * sifts it DOWN (:156-170; the heap property may be violated towards the parent afterwards, the reference does not care),
 * and the tie rules of perlocate_down compare the moved ELEMENT k with the child's heap POSITION (:114-119).
 * The reference's tests hold no vector for this strategy: PARITY UNPINNED (properties + tests/models only).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const uint32_t* key; uint32_t* heap; uint32_t* pos; size_t size, undef; } amh;
static void amh_put(amh* h, size_t p, uint32_t i) { h->heap[p] = i; h->pos[i] = (uint32_t)p; }
static void amh_insert(amh* h, uint32_t i) {                                 /* :61-78 */
    size_t p = h->size++;
    const uint32_t ki = h->key[i];
    while (p > 0 && ki > h->key[h->heap[(p - 1) / 2]]) { amh_put(h, p, h->heap[(p - 1) / 2]); p = (p - 1) / 2; }
    amh_put(h, p, i);
}
static void amh_perlocate_down(amh* h, size_t p, uint32_t k) {               /* :81-137 */
    const uint32_t kk = h->key[k];
    int dir;
    do {
        const size_t lc = 2 * p + 1, rc = 2 * p + 2;
        const uint32_t kl = (lc < h->size) ? h->key[h->heap[lc]] : 0;
        const uint32_t kr = (rc < h->size) ? h->key[h->heap[rc]] : 0;
        if (kk < kl && kk < kr) dir = (kl > kr) ? 1 : 2;
        else if (kk < kl) dir = 1;
        else if (kk < kr) dir = 2;
        else if (kk == kl && kk == kr) dir = (lc < rc) ? 1 : 2;
        else if (kk == kl && k > lc) dir = 1;                                /* element index against heap position, as written */
        else if (kk == kr && k > rc) dir = 2;
        else dir = 0;
        if (dir == 1) { amh_put(h, p, h->heap[lc]); p = lc; }
        else if (dir == 2) { amh_put(h, p, h->heap[rc]); p = rc; }
    } while (dir != 0);
    amh_put(h, p, k);
}
static void amh_remove(amh* h, uint32_t i) {                                 /* :140-154 */
    const size_t p = h->pos[i];
    if (p != h->undef) {
        const uint32_t k = h->heap[--h->size];
        amh_perlocate_down(h, p, k);
        h->pos[i] = (uint32_t)h->undef;
    }
}
size_t orc_max_heap(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n, uint32_t threshold, orc_factor** out) {
    *out = NULL;
    if (n == 0) return 0;
    size_t heap_size = 0;
    for (size_t i = 1; i < n; ++i) if (lcp[i] >= threshold) ++heap_size;       /* MaxHeapStrategy.hpp:52-55 */
    amh h; h.key = lcp; h.size = 0; h.undef = heap_size;
    h.heap = (uint32_t*)malloc((heap_size ? heap_size : 1) * 4);
    h.pos = (uint32_t*)malloc(n * 4);
    for (size_t i = 0; i < n; ++i) h.pos[i] = (uint32_t)heap_size;
    for (size_t i = 1; i < n; ++i) if (lcp[i] >= threshold) amh_insert(&h, (uint32_t)i);   /* :58-61 */
    size_t z = 0, zcap = 1024;
    orc_factor* F = (orc_factor*)malloc(zcap * sizeof(orc_factor));
    while (h.size > 0) {                                                     /* :68-97 */
        const size_t m = h.heap[0];
        const uint32_t fpos = sa[m], fsrc = sa[m - 1], flen = lcp[m];
        if (z == zcap) { zcap *= 2; F = (orc_factor*)realloc(F, zcap * sizeof(orc_factor)); }
        F[z].pos = fpos; F[z].src = fsrc; F[z].len = flen; ++z;
        for (uint32_t k = 0; k < flen; ++k) amh_remove(&h, isa[fpos + k]);   /* :79-81 */
        for (uint32_t k = 0; k < flen && fpos > k; ++k) {                    /* :84-96 */
            const size_t sp = fpos - k - 1;
            const uint32_t i = isa[sp];
            if (h.pos[i] != h.undef && sp + lcp[i] > fpos) {
                const uint32_t nl = (uint32_t)(fpos - sp);
                if (nl >= threshold) { lcp[i] = nl; amh_perlocate_down(&h, h.pos[i], i); }    /* decrease_key :157-170 */
                else amh_remove(&h, i);
            }
        }
    }
    free(h.heap); free(h.pos);
    *out = F;
    return z;
}

size_t orc_max_lcp(const uint32_t* sa, const uint32_t* isa, uint32_t* lcp, size_t n,
                   uint32_t maxlcp, uint32_t threshold, orc_factor** out) {
    *out = NULL;
    if (n == 0) return 0;
    mlsl l; memset(&l, 0, sizeof(l));
    l.lcp = lcp; l.undef = n; l.first = n; l.last = n; l.maxlcp = maxlcp;
    l.prev = (uint32_t*)malloc(n * 4); l.next = (uint32_t*)malloc(n * 4);
    l.lcp_index = (uint32_t*)malloc(((size_t)maxlcp + 1) * 4);
    l.contained = (uint8_t*)calloc(n, 1);
    for (size_t i = 0; i < n; ++i) { l.prev[i] = (uint32_t)n; l.next[i] = (uint32_t)n; }
    for (size_t i = 0; i <= maxlcp; ++i) l.lcp_index[i] = (uint32_t)n;
    for (size_t i = 1; i < n; ++i) if (lcp[i] >= threshold) mlsl_insert(&l, i);   /* ctor :74-78 */
    size_t z = 0, zcap = 1024;
    orc_factor* F = (orc_factor*)malloc(zcap * sizeof(orc_factor));
    while (l.size > 0) {                                                     /* MaxLCPStrategy.hpp:62-95 */
        const size_t m = l.first;
        const uint32_t fpos = sa[m], fsrc = sa[m - 1], flen = lcp[m];
        if (z == zcap) { zcap *= 2; F = (orc_factor*)realloc(F, zcap * sizeof(orc_factor)); }
        F[z].pos = fpos; F[z].src = fsrc; F[z].len = flen; ++z;
        for (uint32_t k = 0; k < flen; ++k) {                                /* :74-79 */
            const size_t i = isa[fpos + k];
            if (l.contained[i]) mlsl_remove(&l, i);
        }
        for (uint32_t k = 0; k < flen && fpos > k; ++k) {                    /* :82-94 */
            const size_t sp = fpos - k - 1;
            const size_t i = isa[sp];
            if (l.contained[i] && sp + lcp[i] > fpos) {
                const uint32_t nl = (uint32_t)(fpos - sp);
                if (nl >= threshold) { mlsl_remove(&l, i); lcp[i] = nl; mlsl_insert(&l, i); }   /* decrease_key :163-167 */
                else mlsl_remove(&l, i);
            }
        }
    }
    free(l.prev); free(l.next); free(l.lcp_index); free(l.contained);
    *out = F;
    return z;
}

/* LZSSFactors.hpp:69-76 (positions are unique, so the order is fully determined) */
/* ------------------------------------------------------------------------------------------------
 * lcpcomp::PLCPPeaksStrategy::factorize (compressors/lcpcomp/compress/PLCPPeaksStrategy.hpp:36-80): one left-to-right scan
 * over the PLCP array; a position whose PLCP value is a strict local maximum (and >= threshold) becomes a factor
 * (i, sa[isa[i]-1], plcp[i]) and the scan jumps behind it.  `plcp` is the array as PLCPFromPhi leaves it: entry n-1 still
 * holds Phi[n-1] (ds/PLCPFromPhi.hpp:27-53), and the scan does read it at i = n-2.
 * ---------------------------------------------------------------------------------------------- */
size_t orc_plcp_peaks(const uint32_t* sa, const uint32_t* isa, const uint32_t* plcp, size_t n, uint32_t threshold,
                      orc_factor** out) {
    u32vec P = {0}, S = {0}, Ln = {0};
    *out = NULL;
    uint32_t last_replacement_pos = 0;                                       /* :52 */
    for (uint32_t i = 0; (size_t)i + 1 < n; ) {                              /* :53 */
        if ((i == last_replacement_pos || plcp[i] > plcp[i - 1]) && plcp[i] > plcp[i + 1] && plcp[i] >= threshold) {   /* :54 */
            const uint32_t len = plcp[i];
            vec_push(&P, i); vec_push(&S, sa[isa[i] - 1]); vec_push(&Ln, len);   /* :57-60 */
            i += len;                                                         /* :72 */
            last_replacement_pos = i - 1;                                    /* :73 */
        } else ++i;                                                          /* :76 */
    }
    const size_t z = P.size;
    orc_factor* f = (orc_factor*)malloc((z ? z : 1) * sizeof(orc_factor));
    for (size_t k = 0; k < z; ++k) { f[k].pos = P.a[k]; f[k].src = S.a[k]; f[k].len = Ln.a[k]; }
    free(P.a); free(S.a); free(Ln.a);
    *out = f;
    return z;
}

static int cmp_factor_pos(const void* a, const void* b) {
    const uint32_t x = ((const orc_factor*)a)->pos, y = ((const orc_factor*)b)->pos;
    return x < y ? -1 : x > y;
}
void orc_sort_factors(orc_factor* f, size_t z) { if (z) qsort(f, z, sizeof(orc_factor), cmp_factor_pos); }

/* LZSSFactors.hpp:79-132 */
void orc_flatten(orc_factor* f, size_t z, uint64_t* num_flattened, uint64_t* max_depth) {
    uint64_t nf = 0, md = 0;
    if (z) {
        const size_t fsize = (size_t)f[z - 1].pos + f[z - 1].len;           /* :86-90 */
        uint32_t* fmap = (uint32_t*)calloc(fsize ? fsize : 1, sizeof(uint32_t));
        for (size_t i = 0; i < z; ++i)                                       /* :92-97 */
            for (size_t j = 0; j < f[i].len; ++j) fmap[f[i].pos + j] = (uint32_t)(i + 1);
        for (size_t i = 0; i < z; ++i) {                                     /* :100-128 */
            uint64_t depth = 0;
            size_t src = f[i].src;
            while (src < fsize && fmap[src]) {
                const orc_factor* s = &f[fmap[src] - 1];
                const size_t d = src - s->pos;
                if (d + f[i].len <= s->len) { src = (size_t)s->src + d; ++depth; }
                else break;
            }
            if (depth) { f[i].src = (uint32_t)src; ++nf; if (depth > md) md = depth; }
        }
        free(fmap);
    }
    if (num_flattened) *num_flattened = nf;
    if (max_depth) *max_depth = md;
}

/* LZSSLiterals.hpp:10-50 : positions not covered by a factor, in text order */
size_t orc_literal_positions(size_t n, const orc_factor* f, size_t z, uint32_t* positions) {
    size_t cnt = 0, p = 0, k = 0;
    while (k < z && p == f[k].pos) { p += f[k].len; ++k; }                  /* skip_factors() in the ctor */
    while (p < n) {
        positions[cnt++] = (uint32_t)p;
        ++p;
        while (k < z && p == f[k].pos) { p += f[k].len; ++k; }
    }
    return cnt;
}

/* HuffmanCoder.hpp:37-48 over TextLiterals */
void orc_literal_histogram(const uint8_t* text, size_t n, const orc_factor* f, size_t z, uint32_t C[256]) {
    memset(C, 0, 256 * sizeof(uint32_t));
    size_t p = 0, k = 0;
    while (k < z && p == f[k].pos) { p += f[k].len; ++k; }
    while (p < n) {
        C[text[p]]++;
        ++p;
        while (k < z && p == f[k].pos) { p += f[k].len; ++k; }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Huffman table.  HuffmanCoder.hpp:88-169 (gen_codelengths) uses std::make_heap/pop_heap/push_heap,
 * :453-455 uses an (unstable) std::sort.  Their libstdc++ behaviour is restated here.
 * ---------------------------------------------------------------------------------------------- */
static size_t* HA;                                            /* the array A of gen_codelengths      */
static int hcomp(size_t a, size_t b) { return HA[a] > HA[b]; }   /* :96 comp(a,b) = A[a] > A[b]       */
typedef int (*cmp_fn)(size_t, size_t);

static void h_push_heap_(size_t* first, long hole, long top, size_t val, cmp_fn comp) {
    long parent = (hole - 1) / 2;
    while (hole > top && comp(first[parent], val)) {
        first[hole] = first[parent];
        hole = parent;
        parent = (hole - 1) / 2;
    }
    first[hole] = val;
}
static void h_adjust_heap(size_t* first, long hole, long len, size_t val, cmp_fn comp) {
    const long top = hole;
    long c = hole;
    while (c < (len - 1) / 2) {
        c = 2 * (c + 1);
        if (comp(first[c], first[c - 1])) --c;
        first[hole] = first[c];
        hole = c;
    }
    if ((len & 1) == 0 && c == (len - 2) / 2) {
        c = 2 * (c + 1);
        first[hole] = first[c - 1];
        hole = c - 1;
    }
    h_push_heap_(first, hole, top, val, comp);
}
static void h_make_heap(size_t* first, long len, cmp_fn comp) {
    if (len < 2) return;
    long parent = (len - 2) / 2;
    for (;;) {
        size_t val = first[parent];
        h_adjust_heap(first, parent, len, val, comp);
        if (parent == 0) return;
        --parent;
    }
}
static void h_pop_heap(size_t* first, long len, cmp_fn comp) {          /* std::pop_heap(first, first+len) */
    if (len > 1) {
        size_t val = first[len - 1];
        first[len - 1] = first[0];
        h_adjust_heap(first, 0, len - 1, val, comp);
    }
}
static void h_push_heap(size_t* first, long len, cmp_fn comp) {         /* std::push_heap(first, first+len) */
    h_push_heap_(first, len - 1, 0, first[len - 1], comp);
}

/* libstdc++ std::sort (introsort + final insertion sort), comparator c(i,j) = len[i] < len[j] */
static const uint8_t* SLEN;
static int scomp(size_t i, size_t j) { return SLEN[(uint8_t)i] < SLEN[(uint8_t)j]; }
static void s_swap(size_t* a, long i, long j) { size_t x = a[i]; a[i] = a[j]; a[j] = x; }
static void s_unguarded_linear_insert(size_t* a, long i) {
    size_t v = a[i];
    while (scomp(v, a[i - 1])) { a[i] = a[i - 1]; --i; }
    a[i] = v;
}
static void s_insertion_sort(size_t* a, long f, long l) {
    if (f == l) return;
    for (long i = f + 1; i < l; ++i) {
        if (scomp(a[i], a[f])) {
            size_t v = a[i];
            memmove(&a[f + 1], &a[f], (size_t)(i - f) * sizeof(size_t));
            a[f] = v;
        } else s_unguarded_linear_insert(a, i);
    }
}
static void s_move_median_to_first(size_t* a, long r, long x, long y, long z) {
    if (scomp(a[x], a[y])) {
        if (scomp(a[y], a[z])) s_swap(a, r, y);
        else if (scomp(a[x], a[z])) s_swap(a, r, z);
        else s_swap(a, r, x);
    } else if (scomp(a[x], a[z])) s_swap(a, r, x);
    else if (scomp(a[y], a[z])) s_swap(a, r, z);
    else s_swap(a, r, y);
}
static long s_unguarded_partition(size_t* a, long f, long l, long p) {
    for (;;) {
        while (scomp(a[f], a[p])) ++f;
        --l;
        while (scomp(a[p], a[l])) --l;
        if (!(f < l)) return f;
        s_swap(a, f, l);
        ++f;
    }
}
static void s_heapsort_fallback(size_t* a, long f, long l) {
    /* std::__partial_sort(f,l,l) = make_heap + sort_heap; not reached for sigma <= 256 in practice
     * (depth limit 2*floor(log2 n)), restated for completeness. */
    h_make_heap(a + f, l - f, scomp);
    for (long len = l - f; len > 1; --len) h_pop_heap(a + f, len, scomp);
}
static void s_introsort_loop(size_t* a, long f, long l, long depth) {
    while (l - f > 16) {
        if (depth == 0) { s_heapsort_fallback(a, f, l); return; }
        --depth;
        long m = f + (l - f) / 2;
        s_move_median_to_first(a, f, f + 1, m, l - 1);
        long cut = s_unguarded_partition(a, f + 1, l, f);
        s_introsort_loop(a, cut, l, depth);
        l = cut;
    }
}
static void s_sort(size_t* a, long n) {
    if (n <= 1) return;
    long lg = 0; { long x = n; while (x > 1) { x >>= 1; ++lg; } }
    s_introsort_loop(a, 0, n, 2 * lg);
    if (n > 16) {
        s_insertion_sort(a, 0, 16);
        for (long i = 16; i < n; ++i) s_unguarded_linear_insert(a, i);
    } else s_insertion_sort(a, 0, n);
}

void orc_huffman_table(const uint32_t C[256], orc_hufftable* t) {
    memset(t, 0, sizeof(*t));
    size_t sigma = 0;
    uint8_t from_eff[256];
    for (int i = 0; i < 256; ++i) if (C[i]) from_eff[sigma++] = (uint8_t)i;      /* :51-78 */
    t->sigma = (uint32_t)sigma;
    if (sigma <= 1) return;                                                      /* :529-536 no table */

    /* gen_codelengths :88-141 */
    size_t A[512];
    for (size_t i = 0; i < sigma; ++i) { A[sigma + i] = C[from_eff[i]]; A[i] = sigma + i; }
    HA = A;
    h_make_heap(A, (long)sigma, hcomp);
    size_t h = sigma - 1;
    while (h > 0) {
        h_pop_heap(A, (long)h + 1, hcomp);
        const size_t m1 = A[h];
        --h;
        h_pop_heap(A, (long)h + 1, hcomp);
        const size_t m2 = A[h];
        A[h + 1] = A[m1] + A[m2];
        A[h] = h + 1;
        A[m1] = A[m2] = h + 1;
        h_push_heap(A, (long)h + 1, hcomp);
    }
    A[1] = 0;
    for (size_t i = 2; i < 2 * sigma; ++i) A[i] = A[A[i]] + 1;
    uint8_t codelengths[256];
    for (size_t i = 0; i < sigma; ++i) codelengths[i] = (uint8_t)A[sigma + i];

    /* gen_huffmantable :450-466 : order by code length with libstdc++'s std::sort */
    size_t order[256];
    for (size_t i = 0; i < sigma; ++i) order[i] = i;
    SLEN = codelengths;
    s_sort(order, (long)sigma);
    uint8_t longest = 0;
    for (size_t i = 0; i < sigma; ++i) if (codelengths[i] > longest) longest = codelengths[i];
    uint8_t ordered_len[256];
    for (size_t i = 0; i < sigma; ++i) { ordered_len[i] = codelengths[order[i]]; t->order[i] = from_eff[order[i]]; }
    t->longest = longest;
    /* gen_numl :173-187 (u8 counters like the reference) */
    for (size_t i = 0; i < sigma; ++i) t->numl[ordered_len[i] - 1]++;
    /* gen_first_codes :192-198 */
    uint64_t firstcode[256];
    firstcode[longest - 1] = 0;
    for (size_t i = longest - 1; i > 0; --i) firstcode[i - 1] = (firstcode[i] + t->numl[i]) / 2;
    /* gen_codewords :202-218 */
    for (size_t i = 0; i < sigma; ++i) {
        const uint8_t sym = t->order[i];
        t->len_of[sym] = ordered_len[i];
        t->code_of[sym] = firstcode[ordered_len[i] - 1]++;
    }
}

/* ------------------------------------------------------------------------------------------------
 * io/BitOStream.hpp:17-164
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint8_t* buf; size_t len, cap; uint8_t next; int cursor; int dirty; } bitout;

static void bo_init(bitout* b) { b->buf = NULL; b->len = b->cap = 0; b->next = 0; b->cursor = 7; b->dirty = 0; }
static void bo_put(bitout* b, uint8_t c) {
    if (b->len == b->cap) { b->cap = b->cap ? b->cap * 2 : 256; b->buf = (uint8_t*)realloc(b->buf, b->cap); }
    b->buf[b->len++] = c;
}
static void bo_write_next(bitout* b) { if (b->dirty) { bo_put(b, b->next); b->next = 0; b->cursor = 7; b->dirty = 0; } }   /* :33-38 */
static void bo_write_bit(bitout* b, int set) {                                                                            /* :79-88 */
    if (set) b->next |= (uint8_t)(1u << b->cursor);
    b->dirty = 1;
    if (--b->cursor < 0) bo_write_next(b);
}
static void bo_write_int(bitout* b, uint64_t v, unsigned bits) {                                                          /* :98-102 */
    for (int i = (int)bits - 1; i >= 0; --i) bo_write_bit(b, i < 64 ? (int)((v >> i) & 1) : 0);
}
static void bo_write_compressed_int(bitout* b, uint64_t v, unsigned blk) {                                                /* :151-163 */
    do {
        uint64_t cur = v;                  /* write_int emits only the low blk bits; the odd mask at :155 is harmless */
        v >>= blk;
        bo_write_bit(b, v > 0);
        bo_write_int(b, cur, blk);
    } while (v > 0);
}
static void bo_finish(bitout* b) {                                                                                        /* dtor :53-64 */
    uint8_t set = (uint8_t)(7 - b->cursor);
    if (b->cursor >= 2) b->next |= set;
    else { bo_write_next(b); b->next = set; }
    b->dirty = 1;
    bo_write_next(b);
}

int orc_bitstream_script(const uint64_t* ops, size_t n_ops, uint8_t** out, size_t* out_len) {
    bitout b; bo_init(&b);
    for (size_t i = 0; i < n_ops; ++i) {
        const uint64_t kind = ops[3 * i], v = ops[3 * i + 1], bits = ops[3 * i + 2];
        if (kind == 0) bo_write_bit(&b, v != 0);
        else if (kind == 1) bo_write_int(&b, v, (unsigned)bits);
        else bo_write_compressed_int(&b, v, (unsigned)bits);
    }
    bo_finish(&b);
    *out = b.buf; *out_len = b.len;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * io/BitIStream.hpp:16-195
 * ---------------------------------------------------------------------------------------------- */
typedef struct { const uint8_t* p; size_t n, idx; uint8_t current, next; int is_final; uint8_t final_bits; uint8_t cursor; } bitin;

static void bi_read_next(bitin* b) {                                     /* :27-63 */
    b->current = b->next;
    b->cursor = 7;
    if (b->idx < b->n) {
        b->next = b->p[b->idx++];
        if (b->idx < b->n) {
            /* stream still going */
        } else {
            b->final_bits = b->next & 0x7;     /* note: c still holds the last successfully read byte */
            if (b->final_bits >= 6) { b->is_final = 1; b->next = 0; }
        }
    } else {
        b->is_final = 1;
        b->final_bits = b->current & 0x7;
        b->next = 0;
    }
}
static void bi_init(bitin* b, const uint8_t* p, size_t n) {              /* :71-83 */
    memset(b, 0, sizeof(*b));
    b->p = p; b->n = n;
    if (n) { b->next = p[b->idx++]; bi_read_next(b); }
    else { b->is_final = 1; b->final_bits = 0; }
}
static int bi_eof(const bitin* b) { return b->is_final && b->cursor <= (7 - b->final_bits); }   /* :191-193 */
static unsigned bi_read_bit(bitin* b) {                                  /* :93-110 */
    if (!bi_eof(b)) {
        unsigned bit = (b->current >> b->cursor) & 1;
        if (b->cursor) --b->cursor; else bi_read_next(b);
        return bit;
    }
    return 0;
}
static uint64_t bi_read_int(bitin* b, unsigned amount) {                 /* :119-127 */
    uint64_t v = 0;
    for (unsigned i = 0; i < amount; ++i) { v <<= 1; v |= bi_read_bit(b); }
    return v;
}
static uint64_t bi_read_compressed_int(bitin* b, unsigned blk) {         /* :174-188 */
    uint64_t value = 0; unsigned i = 0; unsigned has_next;
    do { has_next = bi_read_bit(b); value |= bi_read_int(b, blk) << (blk * (i++)); } while (has_next);
    return value;
}

size_t orc_bitstream_count_bits(const uint8_t* in, size_t n) {
    bitin b; bi_init(&b, in, n);
    size_t cnt = 0;
    while (!bi_eof(&b)) { bi_read_bit(&b); ++cnt; }
    return cnt;
}

/* ------------------------------------------------------------------------------------------------
 * HuffmanCoder::Encoder  (HuffmanCoder.hpp:521-570)
 * ---------------------------------------------------------------------------------------------- */
static void huff_write_header(bitout* b, const orc_hufftable* t) {
    if (t->sigma <= 1) { bo_write_bit(b, 0); return; }                    /* :538-540 */
    bo_write_bit(b, 1);                                                   /* :542 */
    bo_write_compressed_int(b, t->longest, 7);                            /* huffmantable_encode :264-273 */
    for (uint32_t i = 0; i < t->longest; ++i) bo_write_compressed_int(b, t->numl[i], 7);
    bo_write_compressed_int(b, t->sigma, 7);
    for (uint32_t i = 0; i < t->sigma; ++i) bo_write_int(b, t->order[i], 8);
}
static void huff_encode_literal(bitout* b, const orc_hufftable* t, uint8_t c) {   /* :562-569 */
    if (t->sigma == 1) bo_write_int(b, c, 8);
    else bo_write_int(b, t->code_of[c], t->len_of[c]);
}

int orc_huff_encode_literals(const uint8_t* lits, size_t n, int interleave, uint8_t** out, size_t* out_len) {
    uint32_t C[256]; memset(C, 0, sizeof(C));
    for (size_t i = 0; i < n; ++i) C[lits[i]]++;
    orc_hufftable t; orc_huffman_table(C, &t);
    bitout b; bo_init(&b);
    huff_write_header(&b, &t);
    int was_zero = 1;
    for (size_t i = 0; i < n; ++i) {                                      /* test/test/util.hpp:586-597 */
        if (was_zero && interleave) { bo_write_int(&b, 0x55, 8); was_zero = 0; }
        huff_encode_literal(&b, &t, lits[i]);
        if (lits[i] == 0) was_zero = 1;
    }
    bo_finish(&b);
    *out = b.buf; *out_len = b.len;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * ArithmeticCoder::Encoder  (coders/ArithmeticCoder.hpp:35-177): static model, 64-bit interval, codes interleaved
 * into the shared bit stream.  In lcpcomp this combination is not registered and not decodable by the reference
 * (SURVEY 0.3); parity is defined on the compress side only.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    uint32_t C[256];                 /* normalised cumulative counts (:72-92) */
    uint64_t lower, upper;           /* :39-40 */
    uint8_t codebook_size;           /* :41 (uliteral_t: wraps at 256) */
    uint32_t literal_count, literal_counter;   /* :43-44 */
    uint64_t min_range;              /* :45 */
} arith_enc;

static void arith_init(arith_enc* a, const uint32_t counts[256], bitout* b) {
    memcpy(a->C, counts, sizeof(a->C));
    a->lower = 0; a->upper = ~0ull; a->codebook_size = 0; a->literal_counter = 0;
    uint32_t* c = a->C;
    if (c[0] != 0u) a->codebook_size++;                                   /* build_intervals :73-75 */
    uint32_t min = 0xFFFFFFFFu;
    for (int i = 1; i <= 255; ++i) {                                      /* :78-84 */
        if (c[i] != 0u) { a->codebook_size++; if (c[i] < min) min = c[i]; }
        c[i] = c[i] + c[i - 1];
    }
    a->literal_count = c[254];                                            /* :85 */
    for (int i = 0; i <= 255; ++i) c[i] = c[i] / min;                     /* :88-90 */
    a->min_range = c[254];                                                /* :91 */
    bo_write_int(b, a->literal_count, 32);                                /* writeCodebook :128-143 */
    bo_write_int(b, a->codebook_size, 8);
    if (c[0] != 0u) { bo_write_int(b, 0, 8); bo_write_int(b, c[0], 32); }
    for (int i = 1; i <= 255; ++i) if (c[i] != c[i - 1]) { bo_write_int(b, (uint64_t)i, 8); bo_write_int(b, c[i], 32); }
}
/* returns 0, or -1 when the reference itself would divide by zero (no literal byte >= 1 with a non-zero count) */
static int arith_encode(arith_enc* a, bitout* b, uint8_t v) {            /* encode :169-176 */
    a->literal_counter++;
    uint64_t range = a->upper - a->lower;                                 /* setNewBounds :96-117 */
    if (range < a->min_range) {
        bo_write_int(b, a->lower, 64);
        a->lower = 0; a->upper = ~0ull;
        range = a->upper - a->lower;
    }
    const uint64_t tot = a->C[255];
    if (tot == 0) return -1;
    const uint64_t off_u = range <= tot ? (range * a->C[v]) / tot : (range / tot) * a->C[v];
    a->upper = a->lower + off_u;
    if (v != 0) {
        const uint64_t off_l = range <= tot ? (range * a->C[v - 1]) / tot : (range / tot) * a->C[v - 1];
        a->lower = a->lower + off_l;
    }
    if (a->literal_counter == a->literal_count) {                         /* postProcessing :151-155 */
        bo_write_int(b, a->lower, 64);
        bo_write_int(b, ~0ull, 64);
    }
    return 0;
}


/* ------------------------------------------------------------------------------------------------
 * SLECoder (coders/SLECoder.hpp): "static low entropy" coder.  The alphabet is the literal bytes plus the eta most
 * frequent k-mers of the literal runs (k = option "kmer", default 3); symbols are ranked by (count descending, symbol
 * value ascending) (util/Counter.hpp:44-57; a k-mer's symbol value is its bytes, first byte most significant, with
 * 0xFF in the top byte :18-26) and a rank is written in one of a few fixed classes (:165-213).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint64_t sym, cnt; } sle_ent;
static int sle_cmp_sorted(const void* a, const void* b) {                  /* Counter::getSorted :44-57 */
    const sle_ent* x = (const sle_ent*)a, *y = (const sle_ent*)b;
    if (x->cnt != y->cnt) return x->cnt > y->cnt ? -1 : 1;
    return x->sym < y->sym ? -1 : (x->sym > y->sym ? 1 : 0);
}
static int sle_cmp_u64(const void* a, const void* b) {
    const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
    return x < y ? -1 : (x > y ? 1 : 0);
}
static int sle_cmp_sym(const void* a, const void* b) { return sle_cmp_u64(&((const sle_ent*)a)->sym, &((const sle_ent*)b)->sym); }
#define SLE_KMER_MASK (0xFFull << 56)
typedef struct {
    unsigned k, sigma_bits, cur;
    size_t sigma;
    sle_ent* ranked;           /* alphabet in rank order (sym, count) */
    sle_ent* by_sym;           /* (sym, rank) sorted by sym */
    uint8_t buf[8];
} sle_enc;
static uint64_t sle_compile_kmer(const uint8_t* kmer, unsigned k) {        /* :18-26 */
    uint64_t x = 0;
    for (unsigned i = 0; i < k; ++i) x |= (uint64_t)kmer[k - 1 - i] << (8 * i);
    return x | SLE_KMER_MASK;
}
static long sle_rank(const sle_enc* e, uint64_t sym) {
    sle_ent key = { sym, 0 };
    const sle_ent* r = (const sle_ent*)bsearch(&key, e->by_sym, e->sigma, sizeof(sle_ent), sle_cmp_sym);
    return r ? (long)r->cnt : -1;
}
/* Encoder ctor :83-160: count literals and the k-mers of the literal runs, extend the alphabet, write the ranking */
static int sle_init(sle_enc* e, unsigned k, const uint8_t* text, size_t n, const orc_factor* f, size_t z, bitout* b) {
    memset(e, 0, sizeof(*e));
    e->k = k;
    uint32_t* lpos = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!lpos) return -1;
    const size_t nl = orc_literal_positions(n, f, z, lpos);
    uint64_t C[256]; memset(C, 0, sizeof(C));
    uint64_t* km = (uint64_t*)malloc((nl ? nl : 1) * sizeof(uint64_t));
    if (!km) { free(lpos); return -1; }
    size_t nk = 0;
    unsigned cur = 0; size_t last = 0; uint8_t buf[8];
    for (size_t i = 0; i < nl; ++i) {                                      /* :96-120 */
        const size_t pos = lpos[i]; const uint8_t c = text[pos];
        if (k > 1) {
            if (pos != last + 1) cur = 0;
            if (cur == k) { for (unsigned j = 0; j + 1 < k; ++j) buf[j] = buf[j + 1]; --cur; }   /* kmer_roll :55-66 */
            buf[cur++] = c;
            if (cur == k) km[nk++] = sle_compile_kmer(buf, k);
        }
        ++C[c];
        last = pos;
    }
    free(lpos);
    size_t sigma = 0;
    for (int c = 0; c < 256; ++c) sigma += C[c] != 0;
    e->sigma_bits = orc_bits_for(sigma - 1);
    sle_ent* alpha = (sle_ent*)malloc((256 + 2048 + 8) * sizeof(sle_ent));
    size_t na = 0;
    for (int c = 0; c < 256; ++c) if (C[c]) { alpha[na].sym = (uint64_t)c; alpha[na].cnt = C[c]; ++na; }
    if (k > 1) {                                                           /* :126-143 */
        const unsigned add = (((size_t)1 << e->sigma_bits) == sigma) ? 1 : 2;
        size_t eta = ((size_t)1 << (e->sigma_bits + add)) - sigma;
        qsort(km, nk, sizeof(uint64_t), sle_cmp_u64);
        size_t nd = 0;
        sle_ent* kd = (sle_ent*)malloc((nk ? nk : 1) * sizeof(sle_ent));
        for (size_t i = 0; i < nk;) { size_t j = i; while (j < nk && km[j] == km[i]) ++j; kd[nd].sym = km[i]; kd[nd].cnt = j - i; ++nd; i = j; }
        qsort(kd, nd, sizeof(sle_ent), sle_cmp_sorted);
        for (size_t i = 0; i < nd; ++i) { alpha[na++] = kd[i]; if (--eta == 0) break; }
        free(kd);
        sigma = na;
        e->sigma_bits = orc_bits_for(sigma - 1);
    }
    free(km);
    qsort(alpha, na, sizeof(sle_ent), sle_cmp_sorted);                     /* createRanking :60-70 */
    e->sigma = na;
    e->ranked = alpha;
    e->by_sym = (sle_ent*)malloc((na ? na : 1) * sizeof(sle_ent));
    for (size_t i = 0; i < na; ++i) { e->by_sym[i].sym = alpha[i].sym; e->by_sym[i].cnt = i; }
    qsort(e->by_sym, na, sizeof(sle_ent), sle_cmp_sym);
    bo_write_compressed_int(b, sigma, 7);                                  /* :155-158 */
    for (size_t i = 0; i < na; ++i) bo_write_compressed_int(b, alpha[i].sym, 7);
    e->cur = 0;
    return 0;
}
static void sle_free(sle_enc* e) { free(e->ranked); free(e->by_sym); }
static void sle_encode_sym(const sle_enc* e, bitout* b, uint64_t x) {      /* :182-245 */
    const uint64_t r = (uint64_t)sle_rank(e, x);
    const unsigned sb = e->sigma_bits;
    if (sb < 4) bo_write_int(b, r, sb);
    else if (sb < 6) {
        if (r < 4) { bo_write_bit(b, 0); bo_write_int(b, r, 2); }
        else { bo_write_bit(b, 1); bo_write_int(b, r, sb); }
    } else if (sb == 6) {
        if (r < 8) { bo_write_int(b, 0, 2); bo_write_int(b, r, 3); }
        else if (r < 16) { bo_write_int(b, 1, 2); bo_write_int(b, r - 8, 3); }
        else if (r < 32) { bo_write_int(b, 2, 2); bo_write_int(b, r - 16, 4); }
        else { bo_write_int(b, 3, 2); bo_write_int(b, r, sb); }
    } else {
        if (r < 4) { bo_write_int(b, 0, 3); bo_write_int(b, r, 2); }
        else if (r < 8) { bo_write_int(b, 1, 3); bo_write_int(b, r - 4, 2); }
        else if (r < 12) { bo_write_int(b, 2, 3); bo_write_int(b, r - 8, 2); }
        else if (r < 16) { bo_write_int(b, 3, 3); bo_write_int(b, r - 12, 2); }
        else if (r < 24) { bo_write_int(b, 4, 3); bo_write_int(b, r - 16, 3); }
        else if (r < 32) { bo_write_int(b, 5, 3); bo_write_int(b, r - 24, 3); }
        else if (r < 40) { bo_write_int(b, 6, 3); bo_write_int(b, r - 32, 3); }
        else { bo_write_int(b, 7, 3); bo_write_int(b, r, sb); }
    }
}
static void sle_flush(sle_enc* e, bitout* b) {                             /* flush_kmer :172-180 */
    for (unsigned i = 0; i < e->cur; ++i) sle_encode_sym(e, b, e->buf[i]);
    e->cur = 0;
}
static void sle_literal(sle_enc* e, bitout* b, uint8_t c) {                /* encode(v, LiteralRange) :249-266 */
    if (e->cur == e->k) {                                                  /* kmer_roll: the oldest byte leaves as a single symbol */
        const uint8_t out = e->buf[0];
        for (unsigned j = 0; j + 1 < e->k; ++j) e->buf[j] = e->buf[j + 1];
        --e->cur;
        e->buf[e->cur++] = c;
        sle_encode_sym(e, b, out);
    } else e->buf[e->cur++] = c;
    if (e->cur == e->k) {
        const uint64_t x = sle_compile_kmer(e->buf, e->k);
        if (sle_rank(e, x) >= 0) { sle_encode_sym(e, b, x); e->cur = 0; }
    }
}
static void sle_min_distributed(bitout* b, uint64_t v, unsigned bits) {    /* encode(v, MinDistributedRange) :274-296, v already minus min */
    if (bits <= 5) bo_write_int(b, v, bits);
    else if (v < 8) { bo_write_int(b, 0, 2); bo_write_int(b, v, 3); }
    else if (v < 16) { bo_write_int(b, 1, 2); bo_write_int(b, v - 8, 3); }
    else if (v < 32) { bo_write_int(b, 2, 2); bo_write_int(b, v - 16, 4); }
    else { bo_write_int(b, 3, 2); bo_write_int(b, v, bits); }
}
static unsigned g_sle_k = 3;    /* option "kmer" (SLECoder.hpp:38); set around one call, the oracle is single-threaded test code */

/* ------------------------------------------------------------------------------------------------
 * lzss::encode_text (LZSSCoding.hpp:18-92) with tdc::Encoder's binary integer coding (Coder.hpp:61-77)
 * coder: 0 = HuffmanCoder, 1 = ArithmeticCoder, 2 = ASCIICoder (coders/ASCIICoder.hpp:29-50: integers in decimal
 * followed by ':', bits as '0' / '1', literals raw, and NO subtraction of the range minimum), 3 = SLECoder (above; every
 * non-literal write first flushes the pending k-mer buffer, and the factor length -- a MinDistributedRange,
 * LZSSCoding.hpp:42 -- has its own class code)
 * ---------------------------------------------------------------------------------------------- */
static void ascii_int(bitout* b, uint64_t v) {                            /* ASCIICoder.hpp:33-39 */
    char tmp[24]; int k = 0;
    do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
    while (k) bo_write_int(b, (uint8_t)tmp[--k], 8);
    bo_write_int(b, ':', 8);
}
static int encode_stream(const uint8_t* text, size_t n, const orc_factor* f, size_t z, int coder,
                         uint8_t** out, size_t* out_len, orc_stats* st) {
    uint32_t C[256];
    orc_literal_histogram(text, n, f, z, C);
    orc_hufftable t; memset(&t, 0, sizeof(t));
    arith_enc ac;
    sle_enc se;
    bitout b; bo_init(&b);
    if (coder == 0) { orc_huffman_table(C, &t); huff_write_header(&b, &t); }   /* Encoder ctor */
    else if (coder == 1) arith_init(&ac, C, &b);
    else if (coder == 3) { if (sle_init(&se, g_sle_k, text, n, f, z, &b)) return -1; }
    int rc = 0;
#define ENC_LITERAL(ch) do { if (coder == 0) huff_encode_literal(&b, &t, (ch)); else if (coder == 2) bo_write_int(&b, (ch), 8); \
                             else if (coder == 3) sle_literal(&se, &b, (ch)); \
                             else if (arith_encode(&ac, &b, (ch))) rc = -8; } while (0)
#define ENC_INT(v, bits) do { if (coder == 3) sle_flush(&se, &b); \
                              if (coder == 2) ascii_int(&b, (v)); else bo_write_int(&b, (v), (bits)); } while (0)
#define ENC_BIT(x) do { if (coder == 3) sle_flush(&se, &b); \
                        if (coder == 2) bo_write_int(&b, (x) ? '1' : '0', 8); else bo_write_bit(&b, (x)); } while (0)

    uint64_t flen_min = 0xFFFFFFFFull, flen_max = 0, fdist_max = 0;       /* LZSSFactors.hpp:33-38 INDEX_MAX / 0 */
    {
        size_t p = 0;
        for (size_t i = 0; i < z; ++i) {                                   /* :28-38 */
            if (f[i].len < flen_min) flen_min = f[i].len;
            if (f[i].len > flen_max) flen_max = f[i].len;
            if (f[i].pos - p > fdist_max) fdist_max = f[i].pos - p;
            p = (size_t)f[i].pos + f[i].len;
        }
        if (n - p > fdist_max) fdist_max = n - p;
    }
    const unsigned W = orc_bits_for(n);                                    /* Range text_r(n) */
    const unsigned lbits = orc_bits_for(flen_max - flen_min);              /* MinDistributedRange; wraps like size_t when z=0, unused then */
    const unsigned dbits = orc_bits_for(fdist_max);
    const uint64_t lsub = (coder == 2) ? 0 : flen_min;                     /* ASCIICoder ignores the range */
    ENC_INT(n, 32);                                                        /* :47 len_r */
    ENC_INT(flen_min, W);                                                  /* :48 */
    ENC_INT(flen_max, W);                                                  /* :49 */
    ENC_INT(fdist_max, W);                                                 /* :50 */
    size_t p = 0;
    for (size_t i = 0; i < z; ++i) {                                       /* :54-81 */
        if (f[i].pos == p) ENC_BIT(0);
        else { ENC_BIT(1); ENC_INT(f[i].pos - p, dbits); }
        while (p < f[i].pos) ENC_LITERAL(text[p++]);
        ENC_INT(f[i].src, W);
        if (coder == 3) { sle_flush(&se, &b); sle_min_distributed(&b, f[i].len - lsub, lbits); }
        else ENC_INT(f[i].len - lsub, lbits);
        p += f[i].len;
    }
    if (p < n) { ENC_BIT(1); ENC_INT(n - p, dbits); }                      /* :83-86 */
    while (p < n) ENC_LITERAL(text[p++]);                                  /* :88-91 */
#undef ENC_LITERAL
#undef ENC_INT
#undef ENC_BIT
    if (coder == 3) { sle_flush(&se, &b); sle_free(&se); }                 /* ~Encoder :161-167 */
    bo_finish(&b);                                                         /* ~BitOStream */
    *out = b.buf; *out_len = b.len;
    if (st) { st->flen_min = flen_min; st->flen_max = flen_max; st->fdist_max = fdist_max; }
    return rc;
}

int orc_encode_huff(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                    uint8_t** out, size_t* out_len, orc_stats* st) {
    return encode_stream(text, n, f, z, 0, out, out_len, st);
}
int orc_encode_arith(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                     uint8_t** out, size_t* out_len, orc_stats* st) {
    return encode_stream(text, n, f, z, 1, out, out_len, st);
}
int orc_encode_ascii(const uint8_t* text, size_t n, const orc_factor* f, size_t z,
                     uint8_t** out, size_t* out_len, orc_stats* st) {
    return encode_stream(text, n, f, z, 2, out, out_len, st);
}

int orc_encode_sle(const uint8_t* text, size_t n, const orc_factor* f, size_t z, unsigned kmer,
                   uint8_t** out, size_t* out_len, orc_stats* st) {
    if (kmer < 1 || kmer > 7) return -2;
    g_sle_k = kmer;
    const int rc = encode_stream(text, n, f, z, 3, out, out_len, st);
    g_sle_k = 3;
    return rc;
}

/* LCPCompressor.hpp:100-138 */
static int lcpcomp_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                            uint8_t** out, size_t* out_len, orc_stats* stats);
static int g_strategy = 0;      /* 0 = ArraysComp, 1 = PLCPPeaksStrategy, 2 = MaxLCPStrategy (set around one call; the oracle is single-threaded test code) */
int orc_lcpcomp_peaks_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                    uint8_t** out, size_t* out_len, orc_stats* stats) {
    g_strategy = 1;
    const int rc = lcpcomp_compress(text, n, threshold, flatten, 0, out, out_len, stats);
    g_strategy = 0;
    return rc;
}
/* lcpcomp(coder=huff, comp=max_lcp) */
int orc_lcpcomp_maxlcp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                     uint8_t** out, size_t* out_len, orc_stats* stats) {
    g_strategy = 2;
    const int rc = lcpcomp_compress(text, n, threshold, flatten, 0, out, out_len, stats);
    g_strategy = 0;
    return rc;
}
/* lcpcomp(coder=huff, comp=heap): MaxHeapStrategy */
int orc_lcpcomp_heap_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                                   uint8_t** out, size_t* out_len, orc_stats* stats) {
    g_strategy = 3;
    const int rc = lcpcomp_compress(text, n, threshold, flatten, 0, out, out_len, stats);
    g_strategy = 0;
    return rc;
}
int orc_lcpcomp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                              uint8_t** out, size_t* out_len, orc_stats* stats) {
    return lcpcomp_compress(text, n, threshold, flatten, 0, out, out_len, stats);
}
/* LCPCompressor<ArithmeticCoder, ArraysComp, ...>::compress (BASELINE.json configs[2]) */
int orc_lcpcomp_arith_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                               uint8_t** out, size_t* out_len, orc_stats* stats) {
    return lcpcomp_compress(text, n, threshold, flatten, 1, out, out_len, stats);
}
/* lcpcomp(coder=ascii): the human-readable form of the same token stream */
int orc_lcpcomp_ascii_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten,
                               uint8_t** out, size_t* out_len, orc_stats* stats) {
    return lcpcomp_compress(text, n, threshold, flatten, 2, out, out_len, stats);
}
/* lcpcomp(coder=sle(kmer)): the coder of the reference's published lcpcomp runs (etc/compare-suites/default.suite:5) */
int orc_lcpcomp_sle_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten, unsigned kmer,
                             uint8_t** out, size_t* out_len, orc_stats* stats) {
    if (kmer < 1 || kmer > 7) return -2;
    g_sle_k = kmer;
    const int rc = lcpcomp_compress(text, n, threshold, flatten, 3, out, out_len, stats);
    g_sle_k = 3;
    return rc;
}
static int lcpcomp_compress(const uint8_t* text, size_t n, uint32_t threshold, int flatten, int coder,
                            uint8_t** out, size_t* out_len, orc_stats* stats) {
    orc_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof(*stats));
    stats->n = n;
    if (n == 0 || text[n - 1] != 0) return -3;                             /* ds/TextDS.hpp:132-138 */
    const double t0 = now_s();
    uint32_t* sa = (uint32_t*)malloc(n * 4), *isa = (uint32_t*)malloc(n * 4);
    uint32_t* phi = (uint32_t*)malloc(n * 4), *lcp = (uint32_t*)malloc(n * 4);
    if (!sa || !isa || !phi || !lcp) { free(sa); free(isa); free(phi); free(lcp); return -1; }
    double t = now_s();
    int rc = orc_suffix_array(text, n, sa);
    if (rc) { free(sa); free(isa); free(phi); free(lcp); return rc; }
    stats->t_sa = now_s() - t; t = now_s();
    orc_phi(sa, n, phi);                       stats->t_phi = now_s() - t; t = now_s();
    const uint32_t phi_last = phi[n - 1];                  /* what PLCPFromPhi leaves in plcp[n-1] */
    const uint32_t maxlcp = orc_plcp(text, n, phi, phi);   /* in place over phi like the reference */
    stats->t_plcp = now_s() - t; t = now_s();
    orc_lcp(sa, phi, n, lcp);
    orc_isa(sa, n, isa);                       stats->t_isa = now_s() - t; t = now_s();
    stats->maxlcp = maxlcp;
    orc_factor* F = NULL;
    size_t z;
    if (g_strategy == 1) { phi[n - 1] = phi_last; z = orc_plcp_peaks(sa, isa, phi, n, threshold, &F); }
    else if (g_strategy == 2) z = orc_max_lcp(sa, isa, lcp, n, maxlcp, threshold, &F);
    else if (g_strategy == 3) z = orc_max_heap(sa, isa, lcp, n, threshold, &F);
    else z = orc_arrays_comp(sa, isa, lcp, n, maxlcp, threshold, &F);
    free(phi);
    stats->t_factorize = now_s() - t; t = now_s();
    free(sa); free(isa); free(lcp);
    stats->factors = z;
    orc_sort_factors(F, z);                    stats->t_sort = now_s() - t; t = now_s();
    if (flatten) orc_flatten(F, z, &stats->num_flattened, &stats->max_depth_lb);
    stats->t_flatten = now_s() - t; t = now_s();
    rc = encode_stream(text, n, F, z, coder, out, out_len, stats);
    stats->t_encode = now_s() - t;
    free(F);
    stats->t_total = now_s() - t0;
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * compressors/LZSSLCPCompressor.hpp:60-115  (lzss_lcp): for every text position the nearest SA neighbours that start
 * EARLIER in the text (previous / next smaller value in SA), with the LCP minimum on the way; the longer one wins
 * (ties: the upward / PSV side), then the parse jumps over the factor.
 * ---------------------------------------------------------------------------------------------- */
size_t orc_lzss_lcp_factorize(const uint32_t* sa, const uint32_t* isa, const uint32_t* lcp, size_t n, uint32_t threshold,
                              orc_factor** out) {
    size_t z = 0, zcap = 1024;
    orc_factor* F = (orc_factor*)malloc(zcap * sizeof(orc_factor));
    for (size_t i = 0; i + 1 < n;) {                                       /* :62 */
        const size_t cur_pos = isa[i];
        size_t psv_lcp = lcp[cur_pos];                                      /* :71-77 */
        long long psv_pos = (long long)cur_pos - 1;
        if (psv_lcp > 0) {
            while (psv_pos >= 0 && sa[psv_pos] > sa[cur_pos]) {
                const size_t l = lcp[psv_pos--];
                if (l < psv_lcp) psv_lcp = l;
            }
        }
        size_t nsv_lcp = 0;                                                 /* :82-96 */
        size_t nsv_pos = cur_pos + 1;
        if (nsv_pos < n) {
            nsv_lcp = (size_t)-1 >> 1;
            do {
                if (lcp[nsv_pos] < nsv_lcp) nsv_lcp = lcp[nsv_pos];
                if (sa[nsv_pos] < sa[cur_pos]) break;
            } while (++nsv_pos < n);
            if (nsv_pos >= n) nsv_lcp = 0;
        }
        const size_t max_lcp = psv_lcp > nsv_lcp ? psv_lcp : nsv_lcp;       /* :99 */
        if (max_lcp >= threshold) {
            const long long max_pos = (max_lcp == psv_lcp) ? psv_pos : (long long)nsv_pos;   /* :101 */
            if (z == zcap) { zcap *= 2; F = (orc_factor*)realloc(F, zcap * sizeof(orc_factor)); }
            F[z].pos = (uint32_t)i; F[z].src = sa[max_pos]; F[z].len = (uint32_t)max_lcp; ++z;   /* :105 */
            i += max_lcp;
        } else ++i;
    }
    *out = F;
    return z;
}

int orc_lzss_lcp_huff_compress(const uint8_t* text, size_t n, uint32_t threshold, uint8_t** out, size_t* out_len, orc_stats* stats) {
    orc_stats local; if (!stats) stats = &local;
    memset(stats, 0, sizeof(*stats));
    stats->n = n;
    if (n == 0 || text[n - 1] != 0) return -3;
    const double t0 = now_s();
    uint32_t* sa = (uint32_t*)malloc(n * 4), *isa = (uint32_t*)malloc(n * 4);
    uint32_t* phi = (uint32_t*)malloc(n * 4), *lcp = (uint32_t*)malloc(n * 4);
    if (!sa || !isa || !phi || !lcp) { free(sa); free(isa); free(phi); free(lcp); return -1; }
    int rc = orc_suffix_array(text, n, sa);
    if (rc) { free(sa); free(isa); free(phi); free(lcp); return rc; }
    orc_phi(sa, n, phi);
    stats->maxlcp = orc_plcp(text, n, phi, phi);
    orc_lcp(sa, phi, n, lcp);
    orc_isa(sa, n, isa);
    free(phi);
    orc_factor* F = NULL;
    const size_t z = orc_lzss_lcp_factorize(sa, isa, lcp, n, threshold, &F);
    free(sa); free(isa); free(lcp);
    stats->factors = z;
    rc = orc_encode_huff(text, n, F, z, out, out_len, stats);
    free(F);
    stats->t_total = now_s() - t0;
    return rc;
}

/* ------------------------------------------------------------------------------------------------
 * Decoder: HuffmanCoder::Decoder (HuffmanCoder.hpp:572-612) + lcpcomp::decode_text_internal
 * (LCPCompressor.hpp:23-76).  The reference resolves forward references with ScanDec; the decoded
 * text is unique, so references are resolved here by following source chains.
 * ---------------------------------------------------------------------------------------------- */
/* resolve references: every chain ends in a literal (no cycles in a valid stream); resolved chains are cut short */
static int resolve_refs(uint8_t* text, uint32_t* ref, uint64_t n) {
    uint32_t* stack = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!stack) return -1;
    for (uint64_t i = 0; i < n; ++i) {
        if (ref[i] == 0xFFFFFFFFu) continue;
        size_t sp = 0; uint32_t q = (uint32_t)i;
        while (ref[q] != 0xFFFFFFFFu) {
            if (sp >= n) { free(stack); return -6; }
            stack[sp++] = q; q = ref[q];
        }
        const uint8_t c = text[q];
        while (sp) { uint32_t r = stack[--sp]; text[r] = c; ref[r] = 0xFFFFFFFFu; }
    }
    free(stack);
    return 0;
}
/* The same with ASCIICoder::Decoder (ASCIICoder.hpp:53-84): every read is 8 bits; an integer ends at the first
 * non-digit (the ':'), a bit is anything but '0'.  The stream ends with the BitOStream terminator byte. */
static int ascii_read_int(const uint8_t* in, size_t len, size_t* at, uint64_t* v) {
    uint64_t x = 0; int digits = 0;
    while (*at < len) {
        const uint8_t c = in[(*at)++];
        if (c < '0' || c > '9') { *v = x; return digits ? 0 : -4; }
        x = x * 10 + (c - '0'); ++digits;
    }
    return -4;
}
int orc_lcpcomp_ascii_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    if (in_len == 0) return -4;
    const size_t len = in_len - 1;                                          /* minus the terminator byte (all writes are whole bytes) */
    size_t at = 0;
    uint64_t n, flen_min, flen_max, fdist_max;
    if (ascii_read_int(in, len, &at, &n) || ascii_read_int(in, len, &at, &flen_min) ||
        ascii_read_int(in, len, &at, &flen_max) || ascii_read_int(in, len, &at, &fdist_max)) return -4;
    (void)flen_min; (void)flen_max; (void)fdist_max;
    uint8_t* text = (uint8_t*)malloc(n ? n : 1);
    uint32_t* ref = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!text || !ref) { free(text); free(ref); return -1; }
    uint64_t p = 0;
    while (at < len) {                                                      /* LCPCompressor.hpp:45-66 */
        uint64_t num = 0;
        if (in[at++] != '0') { if (ascii_read_int(in, len, &at, &num)) { free(text); free(ref); return -4; } }
        if (p + num > n || at + num > len) { free(text); free(ref); return -4; }
        while (num--) { text[p] = in[at++]; ref[p] = 0xFFFFFFFFu; ++p; }
        if (at < len) {
            uint64_t src, l;
            if (ascii_read_int(in, len, &at, &src) || ascii_read_int(in, len, &at, &l)) { free(text); free(ref); return -4; }
            if (p + l > n || src + l > n) { free(text); free(ref); return -4; }
            for (uint64_t j = 0; j < l; ++j) ref[p + j] = (uint32_t)(src + j);
            p += l;
        }
    }
    if (p != n) { free(text); free(ref); return -5; }
    if (resolve_refs(text, ref, n)) { free(text); free(ref); return -6; }
    free(ref);
    *out = text; *out_len = n;
    return 0;
}

/* The same with SLECoder::Decoder (SLECoder.hpp:301-453): ranking header, rank classes, k-mer symbols expand to k
 * literals; every non-literal read drops the rest of a pending k-mer (never needed for streams the encoder writes). */
static uint64_t sle_read_rank(bitin* b, unsigned sb) {                      /* :367-397 */
    if (sb < 4) return bi_read_int(b, sb);
    if (sb < 6) return bi_read_bit(b) ? bi_read_int(b, sb) : bi_read_int(b, 2);
    if (sb == 6) {
        switch (bi_read_int(b, 2)) {
            case 0: return bi_read_int(b, 3);
            case 1: return 8 + bi_read_int(b, 3);
            case 2: return 16 + bi_read_int(b, 4);
            default: return bi_read_int(b, sb);
        }
    }
    switch (bi_read_int(b, 3)) {
        case 0: return bi_read_int(b, 2);
        case 1: return 4 + bi_read_int(b, 2);
        case 2: return 8 + bi_read_int(b, 2);
        case 3: return 12 + bi_read_int(b, 2);
        case 4: return 16 + bi_read_int(b, 3);
        case 5: return 24 + bi_read_int(b, 3);
        case 6: return 32 + bi_read_int(b, 3);
        default: return bi_read_int(b, sb);
    }
}
int orc_lcpcomp_sle_decompress(const uint8_t* in, size_t in_len, unsigned k, uint8_t** out, size_t* out_len) {
    if (k < 1 || k > 7) return -2;
    bitin b; bi_init(&b, in, in_len);
    const size_t sigma = (size_t)bi_read_compressed_int(&b, 7);             /* Decoder ctor :325-340 */
    if (sigma == 0 || sigma > 4096) return -4;
    const unsigned sb = orc_bits_for(sigma - 1);
    uint64_t* inv = (uint64_t*)malloc(sigma * sizeof(uint64_t));
    for (size_t r = 0; r < sigma; ++r) inv[r] = bi_read_compressed_int(&b, 7);
    uint8_t kmer[8]; size_t kread = (size_t)-1;
    const uint64_t n = bi_read_int(&b, 32);
    const unsigned W = orc_bits_for(n);
    const uint64_t flen_min = bi_read_int(&b, W), flen_max = bi_read_int(&b, W);
    const uint64_t fdist_max = bi_read_int(&b, W);
    const unsigned lbits = orc_bits_for(flen_max - flen_min), dbits = orc_bits_for(fdist_max);
    uint8_t* text = (uint8_t*)malloc(n ? n : 1);
    uint32_t* ref = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    if (!text || !ref || !inv) { free(text); free(ref); free(inv); return -1; }
    uint64_t p = 0;
#define SLE_EOF() (kread < k ? 0 : bi_eof(&b))                              /* eof :351-359 */
    while (!SLE_EOF()) {
        kread = (size_t)-1;
        uint64_t num = bi_read_bit(&b) ? bi_read_int(&b, dbits) : 0;
        while (num--) {
            uint8_t c;
            if (kread < k) c = kmer[kread++];                               /* :362-365 */
            else {
                const uint64_t r = sle_read_rank(&b, sb);
                if (r >= sigma) { free(text); free(ref); free(inv); return -4; }
                const uint64_t x = inv[r];
                if ((x & SLE_KMER_MASK) == SLE_KMER_MASK) {                  /* decompile_kmer :28-32 */
                    for (unsigned i = 0; i < k; ++i) kmer[k - 1 - i] = (uint8_t)(x >> (8 * i));
                    kread = 1; c = kmer[0];
                } else c = (uint8_t)x;
            }
            if (p >= n) { free(text); free(ref); free(inv); return -4; }
            text[p] = c; ref[p] = 0xFFFFFFFFu; ++p;
        }
        if (!SLE_EOF()) {
            kread = (size_t)-1;
            const uint64_t src = bi_read_int(&b, W);
            uint64_t v;                                                     /* decode(MinDistributedRange) :413-431 */
            if (lbits <= 5) v = bi_read_int(&b, lbits);
            else switch (bi_read_int(&b, 2)) {
                case 0: v = bi_read_int(&b, 3); break;
                case 1: v = 8 + bi_read_int(&b, 3); break;
                case 2: v = 16 + bi_read_int(&b, 4); break;
                default: v = bi_read_int(&b, lbits); break;
            }
            const uint64_t len = flen_min + v;
            if (len == 0 || p + len > n || src + len > n) { free(text); free(ref); free(inv); return -4; }
            for (uint64_t j = 0; j < len; ++j) ref[p + j] = (uint32_t)(src + j);
            p += len;
        }
    }
#undef SLE_EOF
    free(inv);
    if (p != n) { free(text); free(ref); return -5; }
    if (resolve_refs(text, ref, n)) { free(text); free(ref); return -6; }
    free(ref);
    *out = text; *out_len = n;
    return 0;
}

int orc_lcpcomp_huff_decompress(const uint8_t* in, size_t in_len, uint8_t** out, size_t* out_len) {
    bitin b; bi_init(&b, in, in_len);
    int have_table = (int)bi_read_bit(&b);
    uint8_t order[256]; uint64_t firstcode[256]; size_t prefix_sum[256];
    unsigned longest = 0; size_t sigma = 0;
    if (have_table) {                                                       /* huffmantable_decode :278-290 */
        longest = (unsigned)(bi_read_compressed_int(&b, 7) & 0xFF);
        if (longest == 0) return -4;
        uint8_t numl[256];
        for (unsigned i = 0; i < longest; ++i) numl[i] = (uint8_t)bi_read_compressed_int(&b, 7);
        sigma = (size_t)bi_read_compressed_int(&b, 7);
        if (sigma > 256) return -4;
        for (size_t i = 0; i < sigma; ++i) order[i] = (uint8_t)bi_read_int(&b, 8);
        firstcode[longest - 1] = 0;                                         /* gen_first_codes :192-198 */
        for (unsigned i = longest - 1; i > 0; --i) firstcode[i - 1] = (firstcode[i] + numl[i]) / 2;
        size_t acc = 0;                                                      /* gen_prefix_sum_lengths :350-370 */
        for (unsigned l = 0; l < longest; ++l) { prefix_sum[l] = acc; acc += numl[l]; }
    }
    const uint64_t n = bi_read_int(&b, 32);                                 /* :27 */
    const unsigned W = orc_bits_for(n);
    const uint64_t flen_min = bi_read_int(&b, W), flen_max = bi_read_int(&b, W);   /* :36-37 */
    const uint64_t fdist_max = bi_read_int(&b, W);                          /* :41 */
    const unsigned lbits = orc_bits_for(flen_max - flen_min), dbits = orc_bits_for(fdist_max);
    uint8_t* text = (uint8_t*)malloc(n ? n : 1);
    uint32_t* ref = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));      /* 0xFFFFFFFF = literal */
    if (!text || !ref) { free(text); free(ref); return -1; }
    uint64_t p = 0;
    while (!bi_eof(&b)) {                                                   /* :45-66 */
        uint64_t num = bi_read_bit(&b) ? bi_read_int(&b, dbits) : 0;
        while (num--) {
            uint8_t c;
            if (!have_table) c = (uint8_t)bi_read_int(&b, 8);               /* :606-607 */
            else {                                                           /* huffman_decode :377-397 */
                uint64_t value = 0; unsigned length = 0;
                do { value = (value << 1) + bi_read_bit(&b); ++length; } while (length <= longest && value < firstcode[length - 1]);
                if (length > longest) { free(text); free(ref); return -4; }
                --length;
                c = order[prefix_sum[length] + (value - firstcode[length])];
            }
            if (p >= n) { free(text); free(ref); return -4; }
            text[p] = c; ref[p] = 0xFFFFFFFFu; ++p;
        }
        if (!bi_eof(&b)) {
            const uint64_t src = bi_read_int(&b, W);
            const uint64_t len = flen_min + bi_read_int(&b, lbits);
            if (p + len > n || src + len > n) { free(text); free(ref); return -4; }
            for (uint64_t j = 0; j < len; ++j) ref[p + j] = (uint32_t)(src + j);
            p += len;
        }
    }
    if (p != n) { free(text); free(ref); return -5; }
    /* resolve references (chains end in literals; no cycles in a valid stream) */
    uint32_t* stack = (uint32_t*)malloc((n ? n : 1) * sizeof(uint32_t));
    for (uint64_t i = 0; i < n; ++i) {
        if (ref[i] == 0xFFFFFFFFu) continue;
        size_t sp = 0; uint32_t q = (uint32_t)i;
        while (ref[q] != 0xFFFFFFFFu) {
            if (sp >= n) { free(stack); free(text); free(ref); return -6; }
            stack[sp++] = q; q = ref[q];
        }
        const uint8_t c = text[q];
        while (sp) { uint32_t r = stack[--sp]; text[r] = c; ref[r] = 0xFFFFFFFFu; }
    }
    free(stack); free(ref);
    *out = text; *out_len = n;
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * LZ78 (compressors/LZ78Compressor.hpp:64-140) with a first-child/next-sibling trie (all tries yield the
 * same factor ids by contract, test/lz78_trie_tests.cpp:61-100) and EliasGammaCoder
 * (coders/EliasGammaCoder.hpp:26-29, io/BitOStream.hpp:105-129).
 * ---------------------------------------------------------------------------------------------- */
typedef struct { uint32_t* first_child; uint32_t* next_sibling; uint8_t* literal; size_t size, cap; } lz78trie;
#define LZ78_UNDEF 0xFFFFFFFFu
static void trie_add(lz78trie* t, uint8_t c) {
    if (t->size == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->first_child = (uint32_t*)realloc(t->first_child, t->cap * 4);
        t->next_sibling = (uint32_t*)realloc(t->next_sibling, t->cap * 4);
        t->literal = (uint8_t*)realloc(t->literal, t->cap);
    }
    t->first_child[t->size] = LZ78_UNDEF; t->next_sibling[t->size] = LZ78_UNDEF; t->literal[t->size] = c; t->size++;
}
/* returns child id; *is_new set when inserted (lz78/BinaryTrie.hpp:73-107) */
static uint32_t trie_find_or_insert(lz78trie* t, uint32_t parent, uint8_t c, int* is_new) {
    const uint32_t newleaf = (uint32_t)t->size;
    *is_new = 0;
    if (t->first_child[parent] == LZ78_UNDEF) t->first_child[parent] = newleaf;
    else {
        uint32_t node = t->first_child[parent];
        for (;;) {
            if (c == t->literal[node]) return node;
            if (t->next_sibling[node] == LZ78_UNDEF) { t->next_sibling[node] = newleaf; break; }
            node = t->next_sibling[node];
        }
    }
    trie_add(t, c);
    *is_new = 1;
    return newleaf;
}

size_t orc_lz78_factors(const uint8_t* in, size_t n, uint32_t** ids, uint8_t** chars) {
    lz78trie t; memset(&t, 0, sizeof(t));
    trie_add(&t, 0);                                                        /* root, id 0 */
    size_t z = 0, cap = 1024;
    uint32_t* I = (uint32_t*)malloc(cap * 4); uint8_t* Cc = (uint8_t*)malloc(cap);
    uint32_t node = 0, parent = 0; uint8_t c = 0;
    for (size_t i = 0; i < n; ++i) {                                         /* :97-121 */
        c = in[i];
        int is_new; uint32_t child = trie_find_or_insert(&t, node, c, &is_new);
        if (is_new) {
            if (z == cap) { cap *= 2; I = (uint32_t*)realloc(I, cap * 4); Cc = (uint8_t*)realloc(Cc, cap); }
            I[z] = node; Cc[z] = c; ++z;
            parent = node = 0;
        } else { parent = node; node = child; }
    }
    if (node != 0) {                                                         /* :124-131 leftover phrase */
        if (z == cap) { cap *= 2; I = (uint32_t*)realloc(I, cap * 4); Cc = (uint8_t*)realloc(Cc, cap); }
        I[z] = parent; Cc[z] = c; ++z;
    }
    free(t.first_child); free(t.next_sibling); free(t.literal);
    *ids = I; *chars = Cc;
    return z;
}

static void bo_write_elias_gamma(bitout* b, uint64_t v) {                    /* io/BitOStream.hpp:105-129 */
    const unsigned k = orc_bits_for(v);
    for (unsigned i = 0; i < k; ++i) bo_write_bit(b, 0);                     /* write_unary(bits_for(v)) */
    bo_write_bit(b, 1);
    bo_write_int(b, v, k);
}

int orc_lz78_gamma_compress(const uint8_t* in, size_t n, uint8_t** out, size_t* out_len) {
    uint32_t* ids; uint8_t* chars;
    size_t z = orc_lz78_factors(in, n, &ids, &chars);
    bitout b; bo_init(&b);
    for (size_t i = 0; i < z; ++i) {
        bo_write_elias_gamma(&b, ids[i]);
        /* NB: the leftover phrase passes a (signed) char (LZ78Compressor.hpp:96,126); bytes >= 0x80 there
         * sign-extend in the reference.  Not reproduced: inputs of configs 1-5 are ASCII. */
        bo_write_elias_gamma(&b, chars[i]);
    }
    bo_finish(&b);
    free(ids); free(chars);
    *out = b.buf; *out_len = b.len;
    return 0;
}
