// pairs statistic: how many table reads does a state-only walker need when one read may cover two (three) AC symbols
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef struct { uint8_t len[65536]; uint8_t sym[65536]; } Lut;
static Lut lut[2][2];
static uint8_t *bits; static size_t nbits;
static int bpm, kcomp[10], td[3], ta[3];
static void build(Lut *L, const uint8_t *cnt, const uint8_t *vals) {
    memset(L, 0, sizeof *L); uint32_t code = 0; int k = 0;
    for (int l = 1; l <= 16; ++l) { for (int i = 0; i < cnt[l - 1]; ++i, ++k, ++code) { uint32_t first = code << (16 - l);
        for (uint32_t f = 0; f < (1u << (16 - l)); ++f) { L->len[first + f] = (uint8_t)l; L->sym[first + f] = vals[k]; } } code <<= 1; }
}
static inline uint32_t peek16(uint64_t p) { uint32_t v = 0; size_t b = p >> 3; for (int i = 0; i < 4; ++i) v = (v << 8) | (b + i < (nbits + 7) / 8 ? bits[b + i] : 0); return (v << (p & 7)) >> 16; }
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    uint8_t *d = malloc(n); if (fread(d, 1, n, f) != (size_t)n) return 1;
    int W = argc > 2 ? atoi(argv[2]) : 9;
    int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}; long i = 2, scan = 0; int ncomp = 3;
    while (i < n) { int m = d[i + 1], len = (d[i + 2] << 8) | d[i + 3];
        if (m == 0xC0) { ncomp = d[i + 9]; for (int c = 0; c < ncomp; ++c) { hs[c] = d[i + 11 + 3 * c] >> 4; vs[c] = d[i + 11 + 3 * c] & 15; } }
        if (m == 0xC4) { long q = i + 4; while (q < i + 2 + len) { int tc = d[q] >> 4, th = d[q] & 15; int tot = 0; for (int k = 0; k < 16; ++k) tot += d[q + 1 + k]; build(&lut[tc][th], d + q + 1, d + q + 17); q += 17 + tot; } }
        if (m == 0xDA) { for (int c = 0; c < ncomp; ++c) { td[c] = d[i + 6 + 2 * c] >> 4; ta[c] = d[i + 6 + 2 * c] & 15; } scan = i + 2 + len; break; }
        i += 2 + len; }
    bpm = 0; for (int c = 0; c < ncomp; ++c) for (int k = 0; k < hs[c] * vs[c]; ++k) kcomp[bpm++] = c;
    bits = malloc(n); size_t o = 0; for (long q = scan; q < n - 2; ++q) { bits[o++] = d[q]; if (d[q] == 0xFF && d[q + 1] == 0) ++q; } nbits = o * 8;
    uint64_t p = 0; int c = 0, z = 0; uint64_t syms = 0, reads1 = 0, reads2 = 0, reads3 = 0; int cover2 = 0, cover3 = 0; int used_in2 = 0, used_in3 = 0; uint64_t longs = 0;
    // simulate: reads2 = reads with pair entries, reads3 with triples
    int skip2 = 0, skip3 = 0;   // symbols still covered by the previous read
    while (p + 32 < nbits) {
        int ac = z != 0, comp = kcomp[c]; const Lut *L = &lut[ac][ac ? ta[comp] : td[comp]];
        uint32_t v = peek16(p); int len = L->len[v], sym = L->sym[v]; if (!len) break;
        int sz = sym & 15, r = ac ? sym >> 4 : 0; int used = len + sz;
        ++syms; ++reads1; if (len > 9) ++longs;
        int adv = ac ? (sz ? r + 1 : (r == 15 ? 16 : 64)) : 1;
        int endblk = z + adv >= 64;
        if (skip2) --skip2; else { ++reads2;
            if (ac && !endblk && len <= 9) { // look at next symbol (AC for sure)
                uint32_t v2 = peek16(p + used); int len2 = L->len[v2]; if (len2 && used + len2 <= W) skip2 = 1; } }
        if (skip3) --skip3; else { ++reads3;
            if (ac && !endblk && len <= 9) { uint32_t v2 = peek16(p + used); int len2 = L->len[v2], sym2 = L->sym[v2];
                if (len2 && used + len2 <= W) { skip3 = 1; int sz2 = sym2 & 15, r2 = sym2 >> 4; int adv2 = sz2 ? r2 + 1 : (r2 == 15 ? 16 : 64);
                    if (z + adv + adv2 < 64) { uint32_t v3 = peek16(p + used + len2 + sz2); int len3 = L->len[v3]; if (len3 && used + len2 + sz2 + len3 <= W) skip3 = 2; } } } }
        p += used; z += adv; if (z >= 64) { z = 0; c = c + 1 == bpm ? 0 : c + 1; }
    }
    printf("window %d bits: symbols %llu, long codes %.2f%%, reads with pairs %.3f per symbol, with triples %.3f\n", W, (unsigned long long)syms, 100.0 * longs / syms, (double)reads2 / syms, (double)reads3 / syms);
    return 0;
}
