// Research tool (CPU, not part of the product): simulates the self-synchronising decode of csrc/jpeg_entropy.hip on one
// baseline 4:2:0 / 4:4:4 file without restart markers and reports, per fixpoint iteration, how many sub-sequences are
// decoded again and HOW a failed re-synchronisation looks: same (bit position, zigzag index) as the previous exit but
// another block-in-MCU phase ("c-shift"), or a different bit position.     gcc -O2 -o sim tools/sim_entropy_sync.c
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct { uint8_t len[65536]; uint8_t sym[65536]; } Lut;
static Lut lut[2][2];                                  // [class dc/ac][id]
static uint8_t *bits; static size_t nbits;
static int bpm, kcomp[10], td[3], ta[3];

static void build(Lut *L, const uint8_t *cnt, const uint8_t *vals) {
    memset(L, 0, sizeof *L);
    uint32_t code = 0; int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < cnt[l - 1]; ++i, ++k, ++code) {
            uint32_t first = code << (16 - l);
            for (uint32_t f = 0; f < (1u << (16 - l)); ++f) { L->len[first + f] = (uint8_t)l; L->sym[first + f] = vals[k]; }
        }
        code <<= 1;
    }
}
static inline uint32_t peek16(uint64_t p) {
    uint32_t v = 0;
    size_t b = p >> 3;
    for (int i = 0; i < 4; ++i) v = (v << 8) | (b + i < (nbits + 7) / 8 ? bits[b + i] : 0);
    return (v << (p & 7)) >> 16;
}
typedef struct { uint64_t p; int c, z; } St;
static St walk(St s, uint64_t end) {
    while (s.p < end) {
        int ac = s.z != 0, comp = kcomp[s.c];
        const Lut *L = &lut[ac][ac ? ta[comp] : td[comp]];
        uint32_t v = peek16(s.p);
        int len = L->len[v] ? L->len[v] : 16, sym = L->len[v] ? L->sym[v] : 0;
        int sz = sym & 15, r = ac ? sym >> 4 : 0;
        s.p += len + sz;
        if (ac && sz == 0) s.z = r == 15 ? s.z + 16 : 64; else s.z = s.z + r + 1;
        if (s.z >= 64) { s.z = 0; s.c = s.c + 1 == bpm ? 0 : s.c + 1; }
    }
    return s;
}
int main(int argc, char **argv) {
    FILE *f = fopen(argv[1], "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); rewind(f);
    uint8_t *d = malloc(n); if (fread(d, 1, n, f) != (size_t)n) return 1;
    int subbits = argc > 2 ? atoi(argv[2]) : 1024;
    int hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}; long i = 2, scan = 0; int ncomp = 3;
    while (i < n) {
        int m = d[i + 1], len = (d[i + 2] << 8) | d[i + 3];
        if (m == 0xC0) { ncomp = d[i + 9]; } if (m == 0xC0) for (int c = 0; c < ncomp; ++c) { hs[c] = d[i + 11 + 3 * c] >> 4; vs[c] = d[i + 11 + 3 * c] & 15; }
        if (m == 0xC4) { long q = i + 4; while (q < i + 2 + len) { int tc = d[q] >> 4, th = d[q] & 15; int tot = 0; for (int k = 0; k < 16; ++k) tot += d[q + 1 + k]; build(&lut[tc][th], d + q + 1, d + q + 17); q += 17 + tot; } }
        if (m == 0xDA) { for (int c = 0; c < ncomp; ++c) { td[c] = d[i + 6 + 2 * c] >> 4; ta[c] = d[i + 6 + 2 * c] & 15; } scan = i + 2 + len; break; }
        i += 2 + len;
    }
    bpm = 0; for (int c = 0; c < ncomp; ++c) for (int k = 0; k < hs[c] * vs[c]; ++k) kcomp[bpm++] = c;
    bits = malloc(n); size_t o = 0;
    for (long q = scan; q < n - 2; ++q) { bits[o++] = d[q]; if (d[q] == 0xFF && d[q + 1] == 0) ++q; }
    nbits = o * 8;
    size_t ns = (nbits + subbits - 1) / subbits;
    St *ex = calloc(ns, sizeof(St)), *used = calloc(ns, sizeof(St)), *nex = calloc(ns, sizeof(St));
    for (size_t j = 0; j < ns; ++j) { St s = {(uint64_t)j * subbits, 0, 0}; used[j] = s; uint64_t e = (j + 1) * (uint64_t)subbits; if (e > nbits) e = nbits; ex[j] = walk(s, e); }
    printf("%zu sub-sequences of %d bits, %d blocks per MCU\n", ns, subbits, bpm);
    for (int it = 1; it < 64; ++it) {
        size_t need = 0, moved = 0, cshift = 0, pz_diff = 0, hyp_ok = 0, hyp_n = 0;
        memcpy(nex, ex, ns * sizeof(St));
        for (size_t j = 1; j < ns; ++j) {
            St e = ex[j - 1];
            if (e.p == used[j].p && e.c == used[j].c && e.z == used[j].z) continue;
            ++need;
            uint64_t end = (j + 1) * (uint64_t)subbits; if (end > nbits) end = nbits;
            St r = walk(e, end); used[j] = e; nex[j] = r;
            if (r.p != ex[j].p || r.c != ex[j].c || r.z != ex[j].z) {
                ++moved;
                if (r.p == ex[j].p && r.z == ex[j].z) ++cshift; else ++pz_diff;
            }
        }
        printf("iteration %2d: decoded again %6zu, exit moved %6zu  (same bit position and zigzag index, other MCU phase: %zu; other position: %zu)\n", it, need, moved, cshift, pz_diff);
        (void)hyp_ok; (void)hyp_n;
        memcpy(ex, nex, ns * sizeof(St));
        if (!need) break;
    }
    return 0;
}
