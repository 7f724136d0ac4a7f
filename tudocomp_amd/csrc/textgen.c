/*
 * textgen.c -- deterministic, integer-only synthetic corpora for the benchmark configurations
 * (BASELINE.json configs 2-5; generator specification: SURVEY.md section 8d).
 *
 * Host-side counterpart of the reference's string generators (include/tudocomp/generators/,
 * used by test/test/util.hpp:180-207): it only produces benchmark/test input, never output bits.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t xs_next(uint64_t* s) {          /* xorshift64* */
    uint64_t x = *s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    *s = x;
    return x * 0x2545F4914F6CDD1Dull;
}

/* English-like text: Zipf-ish words over a 65,536-word random vocabulary, " " / ". " separators. */
int tdc_gen_english(uint8_t* out, size_t N, uint64_t seed) {
    enum { V = 65536 };
    uint64_t s = 0x9E3779B97F4A7C15ull ^ seed;
    uint8_t (*word)[12] = malloc((size_t)V * 12);
    uint8_t* wlen = malloc(V);
    if (!word || !wlen) { free(word); free(wlen); return -1; }
    for (int w = 0; w < V; ++w) {
        int len = 2 + (int)(xs_next(&s) % 9);
        wlen[w] = (uint8_t)len;
        for (int j = 0; j < len; ++j) word[w][j] = (uint8_t)('a' + xs_next(&s) % 26);
    }
    size_t o = 0;
    while (o < N) {
        unsigned b = (unsigned)(xs_next(&s) % 16);
        uint64_t r = ((uint64_t)1 << b) - 1 + xs_next(&s) % ((uint64_t)1 << b);
        const uint8_t* w = word[r];
        for (int j = 0; j < wlen[r] && o < N; ++j) out[o++] = w[j];
        if (xs_next(&s) % 12 == 0) { if (o < N) out[o++] = '.'; if (o < N) out[o++] = ' '; }
        else if (o < N) out[o++] = ' ';
    }
    free(word); free(wlen);
    return 0;
}

/* DNA: 4096-base random blocks; every 4th block (in expectation) is a copy of an earlier window. */
int tdc_gen_dna(uint8_t* out, size_t N, uint64_t seed) {
    static const char ACGT[4] = { 'A', 'C', 'G', 'T' };
    uint64_t s = 0x9E3779B97F4A7C15ull ^ seed;
    size_t len = 0;
    while (len < N) {
        uint64_t x = xs_next(&s);
        if (x % 4 == 0 && len >= 4096) {
            size_t src = (size_t)(xs_next(&s) % len);
            for (int j = 0; j < 4096; ++j) { uint8_t c = out[src + j]; if (len < N) out[len] = c; ++len; if (len >= N) break; }
        } else {
            for (int j = 0; j < 4096; ++j) { uint8_t c = (uint8_t)ACGT[xs_next(&s) >> 62]; if (len < N) out[len] = c; ++len; }
        }
    }
    return 0;
}
