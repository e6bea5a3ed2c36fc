/*
 * jpeg_oracle.c -- CPU ORACLE for the JPEG pixel stage (dequant + IDCT + chroma upsample + YCbCr->BGRA).
 * TEST INFRASTRUCTURE ONLY -- never linked or called by the product (see oracle/if_oracle.c header).
 *
 * What the reference does on this path: codecs/mozjpeg_decoder.rs:295-420 drives libjpeg (mozjpeg-sys 2.2.3,
 * Cargo.lock:1940-1948, C sources NOT under /root/reference) with every decompress parameter left at its default
 * except out_color_space = JCS_EXT_BGRA (:320) and scale_num/8 (:610-611): dct_method = JDCT_ISLOW,
 * do_fancy_upsampling = TRUE.  The arithmetic therefore is the public IJG algorithm family, restated here from its
 * published description (ITU T.81 Annex A/F + the IJG "islow" 13-bit fixed-point factorisation, the triangle
 * ("fancy") chroma up-sampler and the 16-bit fixed-point YCbCr tables):
 *   jidctint.c  jpeg_idct_islow      -> idct_islow_block()
 *   jdsample.c  h2v1/h2v2_fancy_upsample, jdmainct.c context-row duplication -> upsample_*()
 *   jdcolor.c   build_ycc_rgb_table / ycc_rgb_convert -> ycc_to_bgra()
 * plus a baseline (SOF0, 8-bit, Huffman, interleaved scan, optional DRI) entropy decoder that only exists to get
 * quantised coefficient planes out of real .jpg files for the tests (the product receives coefficients from
 * libjpeg's jpeg_read_coefficients on the host; entropy decoding is out of the GPU stage's scope).
 *
 * PARITY PIN: tests/test_oracle_jpeg.py decodes Pillow-encoded files with this oracle and requires byte equality
 * with Pillow's own decode (libjpeg-turbo 3.1.4.1: the same islow / fancy / fixed-point family, and libjpeg-turbo's
 * SIMD paths are bit-exact with its C paths).  mozjpeg itself is not available: parity with it is by algorithm family.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define JO_OK 0
#define JO_ERR_FORMAT 1
#define JO_ERR_UNSUPPORTED 2
#define JO_ERR_ALLOC 4

/* ------------------------------------------------------------------------------------------------ */
/* Baseline JPEG parser + Huffman decoder -> coefficient planes                                      */
/* ------------------------------------------------------------------------------------------------ */
static const uint8_t ZIGZAG[64] = {
     0,  1,  8, 16,  9,  2,  3, 10, 17, 24, 32, 25, 18, 11,  4,  5,
    12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13,  6,  7, 14, 21, 28,
    35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51,
    58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

typedef struct {
    int32_t mincode[17], maxcode[18], valptr[17];
    uint8_t vals[256];
    int present;
} jo_huff;

typedef struct {
    uint32_t width, height;
    int ncomp;
    uint8_t id[3], h[3], v[3], tq[3], td[3], ta[3];
    uint16_t qt[4][64];            /* natural order */
    int qt_present[4];
    jo_huff dc[4], ac[4];
    int restart_interval;
    int hmax, vmax;
    uint32_t mcus_w, mcus_h;
    uint32_t bw[3], bh[3];         /* blocks per row / column per component, MCU padded */
} jo_info;

typedef struct {
    const uint8_t* p;
    const uint8_t* end;
    uint32_t bits;
    int nbits;
    int hit_marker;
} jo_bits;

static void huff_build(jo_huff* h, const uint8_t* counts, const uint8_t* vals, int nvals) {
    int code = 0, k = 0;
    memcpy(h->vals, vals, (size_t)nvals);
    for (int l = 1; l <= 16; l++) {
        h->valptr[l] = k;
        h->mincode[l] = code;
        code += counts[l - 1];
        k += counts[l - 1];
        h->maxcode[l] = counts[l - 1] ? code - 1 : -1;
        code <<= 1;
    }
    h->maxcode[17] = 0x7fffffff;
    h->present = 1;
}

static int get_bit(jo_bits* b) {
    if (b->nbits == 0) {
        uint32_t c = 0;
        if (b->p < b->end && !b->hit_marker) {
            c = *b->p++;
            if (c == 0xFF) {
                if (b->p < b->end && *b->p == 0x00) b->p++;
                else { b->hit_marker = 1; b->p--; c = 0; }
            }
        }
        b->bits = c;
        b->nbits = 8;
    }
    b->nbits--;
    return (int)((b->bits >> b->nbits) & 1u);
}
static int get_bits(jo_bits* b, int n) { int v = 0; while (n--) v = (v << 1) | get_bit(b); return v; }
static int huff_decode(jo_bits* b, const jo_huff* h) {
    int code = 0;
    for (int l = 1; l <= 16; l++) {
        code = (code << 1) | get_bit(b);
        if (h->maxcode[l] >= 0 && code <= h->maxcode[l] && code >= h->mincode[l])
            return h->vals[h->valptr[l] + code - h->mincode[l]];
    }
    return -1;
}
static int extend(int v, int t) { return (t && v < (1 << (t - 1))) ? v - (1 << t) + 1 : v; }

static int parse_headers(const uint8_t* d, size_t len, jo_info* I, size_t* scan_off) {
    memset(I, 0, sizeof *I);
    if (len < 4 || d[0] != 0xFF || d[1] != 0xD8) return JO_ERR_FORMAT;
    size_t p = 2;
    while (p + 4 <= len) {
        if (d[p] != 0xFF) return JO_ERR_FORMAT;
        uint8_t m = d[p + 1];
        if (m == 0xFF) { p++; continue; }
        uint32_t seglen = ((uint32_t)d[p + 2] << 8) | d[p + 3];
        if (p + 2 + seglen > len) return JO_ERR_FORMAT;
        const uint8_t* s = d + p + 4;
        uint32_t n = seglen - 2;
        if (m == 0xDB) {                                          /* DQT */
            uint32_t q = 0;
            while (q < n) {
                int pq = s[q] >> 4, tq = s[q] & 15;
                if (pq != 0 || tq > 3) return JO_ERR_UNSUPPORTED;
                for (int i = 0; i < 64; i++) I->qt[tq][ZIGZAG[i]] = s[q + 1 + i];
                I->qt_present[tq] = 1;
                q += 65;
            }
        } else if (m == 0xC0 || m == 0xC1) {                      /* SOF0 / SOF1 (8-bit) */
            if (s[0] != 8) return JO_ERR_UNSUPPORTED;
            I->height = ((uint32_t)s[1] << 8) | s[2];
            I->width = ((uint32_t)s[3] << 8) | s[4];
            I->ncomp = s[5];
            if (I->ncomp != 1 && I->ncomp != 3) return JO_ERR_UNSUPPORTED;
            for (int c = 0; c < I->ncomp; c++) {
                I->id[c] = s[6 + 3 * c];
                I->h[c] = s[7 + 3 * c] >> 4;
                I->v[c] = s[7 + 3 * c] & 15;
                I->tq[c] = s[8 + 3 * c];
            }
        } else if (m == 0xC2 || (m >= 0xC5 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC)) {
            return JO_ERR_UNSUPPORTED;                            /* progressive / arithmetic / lossless */
        } else if (m == 0xC4) {                                   /* DHT */
            uint32_t q = 0;
            while (q < n) {
                int tc = s[q] >> 4, th = s[q] & 15;
                if (th > 3) return JO_ERR_FORMAT;
                int total = 0;
                for (int i = 0; i < 16; i++) total += s[q + 1 + i];
                if (total > 256) return JO_ERR_FORMAT;
                huff_build(tc ? &I->ac[th] : &I->dc[th], s + q + 1, s + q + 17, total);
                q += 17 + (uint32_t)total;
            }
        } else if (m == 0xDD) {
            I->restart_interval = (s[0] << 8) | s[1];
        } else if (m == 0xDA) {                                   /* SOS */
            int ns = s[0];
            if (ns != I->ncomp) return JO_ERR_UNSUPPORTED;        /* single interleaved scan only */
            for (int k = 0; k < ns; k++) {
                int cid = s[1 + 2 * k], c = -1;
                for (int j = 0; j < I->ncomp; j++) if (I->id[j] == cid) c = j;
                if (c < 0) return JO_ERR_FORMAT;
                I->td[c] = s[2 + 2 * k] >> 4;
                I->ta[c] = s[2 + 2 * k] & 15;
            }
            *scan_off = p + 2 + seglen;
            break;
        }
        p += 2 + seglen;
    }
    if (!I->width || !I->height || !*scan_off) return JO_ERR_FORMAT;
    if (I->ncomp == 1) { I->h[0] = I->v[0] = 1; }
    I->hmax = I->vmax = 1;
    for (int c = 0; c < I->ncomp; c++) {
        if (I->h[c] < 1 || I->h[c] > 2 || I->v[c] < 1 || I->v[c] > 2) return JO_ERR_UNSUPPORTED;
        if (I->h[c] > I->hmax) I->hmax = I->h[c];
        if (I->v[c] > I->vmax) I->vmax = I->v[c];
    }
    I->mcus_w = (I->width + 8 * I->hmax - 1) / (8 * I->hmax);
    I->mcus_h = (I->height + 8 * I->vmax - 1) / (8 * I->vmax);
    for (int c = 0; c < I->ncomp; c++) { I->bw[c] = I->mcus_w * I->h[c]; I->bh[c] = I->mcus_h * I->v[c]; }
    return JO_OK;
}

/* Header query: out9 = {width, height, ncomp, h0, h1, h2, v0, v1, v2}. */
int jo_jpeg_info(const uint8_t* d, size_t len, uint32_t* out9) {
    jo_info I; size_t so = 0;
    int rc = parse_headers(d, len, &I, &so);
    if (rc) return rc;
    out9[0] = I.width; out9[1] = I.height; out9[2] = (uint32_t)I.ncomp;
    for (int c = 0; c < 3; c++) { out9[3 + c] = c < I.ncomp ? I.h[c] : 0; out9[6 + c] = c < I.ncomp ? I.v[c] : 0; }
    return JO_OK;
}
int jo_jpeg_block_dims(const uint8_t* d, size_t len, uint32_t* bw3, uint32_t* bh3) {
    jo_info I; size_t so = 0;
    int rc = parse_headers(d, len, &I, &so);
    if (rc) return rc;
    for (int c = 0; c < 3; c++) { bw3[c] = c < I.ncomp ? I.bw[c] : 0; bh3[c] = c < I.ncomp ? I.bh[c] : 0; }
    return JO_OK;
}

/* Entropy-decode into quantised coefficient planes coef[c][bh][bw][64] (natural order, as jpeg_read_coefficients)
 * and the quantisation tables qt[c][64] (natural order). Buffers are caller-allocated from jo_jpeg_block_dims. */
int jo_jpeg_read_coefficients(const uint8_t* d, size_t len, int16_t* coef0, int16_t* coef1, int16_t* coef2,
                              uint16_t* qt3x64) {
    jo_info I; size_t so = 0;
    int rc = parse_headers(d, len, &I, &so);
    if (rc) return rc;
    int16_t* coef[3] = {coef0, coef1, coef2};
    for (int c = 0; c < I.ncomp; c++) {
        if (!I.qt_present[I.tq[c]] || !I.dc[I.td[c]].present || !I.ac[I.ta[c]].present) return JO_ERR_FORMAT;
        memcpy(qt3x64 + 64 * c, I.qt[I.tq[c]], 128);
        memset(coef[c], 0, sizeof(int16_t) * 64 * (size_t)I.bw[c] * I.bh[c]);
    }
    jo_bits b = {d + so, d + len, 0, 0, 0};
    int pred[3] = {0, 0, 0};
    uint32_t mcu_count = 0, total = I.mcus_w * I.mcus_h;
    for (uint32_t my = 0; my < I.mcus_h; my++) {
        for (uint32_t mx = 0; mx < I.mcus_w; mx++) {
            if (I.restart_interval && mcu_count && (mcu_count % (uint32_t)I.restart_interval) == 0) {
                /* RSTn: byte align, skip marker, reset predictors */
                b.nbits = 0; b.hit_marker = 0;
                while (b.p + 1 < b.end && !(b.p[0] == 0xFF && b.p[1] >= 0xD0 && b.p[1] <= 0xD7)) b.p++;
                if (b.p + 1 < b.end) b.p += 2;
                pred[0] = pred[1] = pred[2] = 0;
            }
            for (int c = 0; c < I.ncomp; c++) {
                for (int by = 0; by < I.v[c]; by++) for (int bx = 0; bx < I.h[c]; bx++) {
                    uint32_t brow = my * I.v[c] + (uint32_t)by, bcol = mx * I.h[c] + (uint32_t)bx;
                    int16_t* blk = coef[c] + 64 * ((size_t)brow * I.bw[c] + bcol);
                    int t = huff_decode(&b, &I.dc[I.td[c]]);
                    if (t < 0 || t > 11) return JO_ERR_FORMAT;
                    int diff = extend(get_bits(&b, t), t);
                    pred[c] += diff;
                    blk[0] = (int16_t)pred[c];
                    for (int k = 1; k < 64;) {
                        int rs = huff_decode(&b, &I.ac[I.ta[c]]);
                        if (rs < 0) return JO_ERR_FORMAT;
                        int r = rs >> 4, sz = rs & 15;
                        if (sz == 0) { if (r == 15) { k += 16; continue; } break; }
                        k += r;
                        if (k > 63) return JO_ERR_FORMAT;
                        blk[ZIGZAG[k]] = (int16_t)extend(get_bits(&b, sz), sz);
                        k++;
                    }
                }
            }
            mcu_count++;
        }
    }
    (void)total;
    return JO_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* IDCT: the "islow" factorisation (Loeffler-Ligtenberg-Moshytz, 12 multiplies), 13-bit constants     */
/* ------------------------------------------------------------------------------------------------ */
#define CONST_BITS 13
#define PASS1_BITS 2
#define FIX_0_298631336 2446
#define FIX_0_390180644 3196
#define FIX_0_541196100 4433
#define FIX_0_765366865 6270
#define FIX_0_899976223 7373
#define FIX_1_175875602 9633
#define FIX_1_501321110 12299
#define FIX_1_847759065 15137
#define FIX_1_961570560 16069
#define FIX_2_053119869 16819
#define FIX_2_562915447 20995
#define FIX_3_072711026 25172
#define DESCALE(x, n) (((x) + ((int32_t)1 << ((n) - 1))) >> (n))

static inline uint8_t range_limit(int32_t v) {
    /* libjpeg's post-IDCT table indexed with (v & RANGE_MASK), RANGE_MASK = 1023: [0,128) -> v+128, [128,512) -> 255,
       [512,896) -> 0, [896,1024) -> v-896.  A clamp for |v| < 512, and the same wrap-around beyond (only reachable
       with coefficients no real encoder emits, but tests feed random blocks). */
    uint32_t i = (uint32_t)v & 1023u;
    if (i < 128u) return (uint8_t)(i + 128u);
    if (i < 512u) return 255;
    if (i < 896u) return 0;
    return (uint8_t)(i - 896u);
}

void jo_idct_islow_block(const int16_t* coef, const uint16_t* quant, uint8_t* out, int out_stride) {
    int32_t ws[64];
    for (int col = 0; col < 8; col++) {
        const int16_t* in = coef + col;
        const uint16_t* q = quant + col;
        int32_t* w = ws + col;
        if (in[8] == 0 && in[16] == 0 && in[24] == 0 && in[32] == 0 && in[40] == 0 && in[48] == 0 && in[56] == 0) {
            int32_t dc = (int32_t)((uint32_t)((int32_t)in[0] * (int32_t)q[0]) << PASS1_BITS);
            for (int r = 0; r < 8; r++) w[8 * r] = dc;
            continue;
        }
        int32_t z2 = in[16] * q[16], z3 = in[48] * q[48];
        int32_t z1 = (z2 + z3) * FIX_0_541196100;
        int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int32_t tmp3 = z1 + z2 * FIX_0_765366865;
        z2 = in[0] * q[0]; z3 = in[32] * q[32];
        int32_t tmp0 = (int32_t)((uint32_t)(z2 + z3) << CONST_BITS);
        int32_t tmp1 = (int32_t)((uint32_t)(z2 - z3) << CONST_BITS);
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = in[56] * q[56]; tmp1 = in[40] * q[40]; tmp2 = in[24] * q[24]; tmp3 = in[8] * q[8];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        w[0]  = DESCALE(tmp10 + tmp3, CONST_BITS - PASS1_BITS);
        w[56] = DESCALE(tmp10 - tmp3, CONST_BITS - PASS1_BITS);
        w[8]  = DESCALE(tmp11 + tmp2, CONST_BITS - PASS1_BITS);
        w[48] = DESCALE(tmp11 - tmp2, CONST_BITS - PASS1_BITS);
        w[16] = DESCALE(tmp12 + tmp1, CONST_BITS - PASS1_BITS);
        w[40] = DESCALE(tmp12 - tmp1, CONST_BITS - PASS1_BITS);
        w[24] = DESCALE(tmp13 + tmp0, CONST_BITS - PASS1_BITS);
        w[32] = DESCALE(tmp13 - tmp0, CONST_BITS - PASS1_BITS);
    }
    for (int row = 0; row < 8; row++) {
        const int32_t* w = ws + 8 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t z2 = w[2], z3 = w[6];
        int32_t z1 = (z2 + z3) * FIX_0_541196100;
        int32_t tmp2 = z1 + z3 * (-FIX_1_847759065);
        int32_t tmp3 = z1 + z2 * FIX_0_765366865;
        int32_t tmp0 = (int32_t)((uint32_t)(w[0] + w[4]) << CONST_BITS);
        int32_t tmp1 = (int32_t)((uint32_t)(w[0] - w[4]) << CONST_BITS);
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        tmp0 = w[7]; tmp1 = w[5]; tmp2 = w[3]; tmp3 = w[1];
        z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
        int32_t z4 = tmp1 + tmp3;
        int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp0 *= FIX_0_298631336; tmp1 *= FIX_2_053119869; tmp2 *= FIX_3_072711026; tmp3 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
        const int S = CONST_BITS + PASS1_BITS + 3;
        o[0] = range_limit(DESCALE(tmp10 + tmp3, S));
        o[7] = range_limit(DESCALE(tmp10 - tmp3, S));
        o[1] = range_limit(DESCALE(tmp11 + tmp2, S));
        o[6] = range_limit(DESCALE(tmp11 - tmp2, S));
        o[2] = range_limit(DESCALE(tmp12 + tmp1, S));
        o[5] = range_limit(DESCALE(tmp12 - tmp1, S));
        o[3] = range_limit(DESCALE(tmp13 + tmp0, S));
        o[4] = range_limit(DESCALE(tmp13 - tmp0, S));
    }
}

/* ------------------------------------------------------------------------------------------------ */
/* Full-size pixel stage: coefficient planes -> BGRA8                                                */
/* ------------------------------------------------------------------------------------------------ */
static int32_t cr_r[256], cb_b[256], cr_g[256], cb_g[256];
static int ycc_ready = 0;
static void build_ycc(void) {
    if (ycc_ready) return;
    for (int i = 0; i < 256; i++) {
        int32_t x = i - 128;
        cr_r[i] = (int32_t)((91881 * x + 32768) >> 16);          /* FIX(1.40200) */
        cb_b[i] = (int32_t)((116130 * x + 32768) >> 16);         /* FIX(1.77200) */
        cr_g[i] = -46802 * x;                                    /* FIX(0.71414) */
        cb_g[i] = -22554 * x + 32768;                            /* FIX(0.34414) + ONE_HALF */
    }
    ycc_ready = 1;
}
static inline uint8_t clamp255(int32_t v) { return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v)); }

/* chroma sample with libjpeg's edge rules: rows/cols outside the component's downsampled size duplicate the edge */
static inline int32_t cs(const uint8_t* plane, uint32_t pw, uint32_t dw, uint32_t dh, int32_t x, int32_t y) {
    if (x < 0) x = 0;
    if (x >= (int32_t)dw) x = (int32_t)dw - 1;
    if (y < 0) y = 0;
    if (y >= (int32_t)dh) y = (int32_t)dh - 1;
    return plane[(size_t)y * pw + (size_t)x];
}

/*
 * coef[c]: [bh_c][bw_c][64] natural order; qt: [ncomp][64]; hs/vs: sampling factors.
 * Output BGRA8 with alpha = 255 (what JCS_EXT_BGRA writes), rows `stride` bytes.
 */
int jo_jpeg_idct_color_scaled(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                              int ncomp, const uint8_t* hs, const uint8_t* vs, uint32_t width, uint32_t height,
                              int scale_num, int luma_mode,
                              const int8_t* w7x8, const uint8_t* log2div7, const uint16_t* s2l12, const uint8_t* l2s12,
                              uint8_t* bgra, uint32_t stride);
int jo_jpeg_idct_color(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                       int ncomp, const uint8_t* hs, const uint8_t* vs, uint32_t width, uint32_t height,
                       uint8_t* bgra, uint32_t stride) {
    /* full size = the scaled stage at 8/8 (no block scaler tables are touched there) */
    return jo_jpeg_idct_color_scaled(coef0, coef1, coef2, qt, ncomp, hs, vs, width, height, 8, 0, 0, 0, 0, 0, bgra, stride);
}

/* ------------------------------------------------------------------------------------------------ */
/* Reduced-size decode (scale_num/8, scale_num in {1,2,4})                                           */
/*                                                                                                  */
/* libjpeg's reduced-size inverse DCTs (jidctred.c: jpeg_idct_4x4 / 2x2 / 1x1) and its rule for      */
/* picking each component's IDCT size (jdmaster.c: a sub-sampled component takes an IDCT twice as    */
/* large instead of being up-sampled).  With imageflow's selector installed the luma component uses  */
/* islow + flow_scale_spatial[_srgb]_NxN instead (codec_jpeg_wrapper.c:274-343).                     */
/* Pin: luma_mode 0 against Pillow's draft-mode decode; the spatial scalers against the reference's   */
/* own compiled functions.                                                                          */
/* ------------------------------------------------------------------------------------------------ */
#define FIX_0_211164243 1730
#define FIX_0_509795579 4176
#define FIX_0_601344887 4926
#define FIX_0_720959822 5906
#define FIX_0_850430095 6967
#define FIX_1_061594337 8697
#define FIX_1_272758580 10426
#define FIX_1_451774981 11893
#define FIX_2_172734803 17799
#define FIX_3_624509785 29692

void jo_idct_4x4_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[8 * 4];
    for (int col = 0; col < 8; col++) {
        if (col == 4) continue;                              /* column 4 is not used by the second pass */
        const int16_t* c = in + col;
        const uint16_t* qq = q + col;
        int32_t* w = ws + col;
        if (c[8] == 0 && c[16] == 0 && c[24] == 0 && c[40] == 0 && c[48] == 0 && c[56] == 0) {
            int32_t dc = (int32_t)((uint32_t)(c[0] * qq[0]) << PASS1_BITS);
            w[0] = w[8] = w[16] = w[24] = dc;
            continue;
        }
        int32_t tmp0 = (int32_t)((uint32_t)(c[0] * qq[0]) << (CONST_BITS + 1));
        int32_t z2 = c[16] * qq[16], z3 = c[48] * qq[48];
        int32_t tmp2 = z2 * FIX_1_847759065 + z3 * (-FIX_0_765366865);
        int32_t tmp10 = tmp0 + tmp2, tmp12 = tmp0 - tmp2;
        int32_t z1 = c[56] * qq[56];
        z2 = c[40] * qq[40]; z3 = c[24] * qq[24];
        int32_t z4 = c[8] * qq[8];
        tmp0 = z1 * (-FIX_0_211164243) + z2 * FIX_1_451774981 + z3 * (-FIX_2_172734803) + z4 * FIX_1_061594337;
        tmp2 = z1 * (-FIX_0_509795579) + z2 * (-FIX_0_601344887) + z3 * FIX_0_899976223 + z4 * FIX_2_562915447;
        w[0]  = DESCALE(tmp10 + tmp2, CONST_BITS - PASS1_BITS + 1);
        w[24] = DESCALE(tmp10 - tmp2, CONST_BITS - PASS1_BITS + 1);
        w[8]  = DESCALE(tmp12 + tmp0, CONST_BITS - PASS1_BITS + 1);
        w[16] = DESCALE(tmp12 - tmp0, CONST_BITS - PASS1_BITS + 1);
    }
    for (int row = 0; row < 4; row++) {
        const int32_t* w = ws + 8 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t tmp0 = (int32_t)((uint32_t)w[0] << (CONST_BITS + 1));
        int32_t tmp2 = w[2] * FIX_1_847759065 + w[6] * (-FIX_0_765366865);
        int32_t tmp10 = tmp0 + tmp2, tmp12 = tmp0 - tmp2;
        int32_t z1 = w[7], z2 = w[5], z3 = w[3], z4 = w[1];
        tmp0 = z1 * (-FIX_0_211164243) + z2 * FIX_1_451774981 + z3 * (-FIX_2_172734803) + z4 * FIX_1_061594337;
        tmp2 = z1 * (-FIX_0_509795579) + z2 * (-FIX_0_601344887) + z3 * FIX_0_899976223 + z4 * FIX_2_562915447;
        const int S = CONST_BITS + PASS1_BITS + 3 + 1;
        o[0] = range_limit(DESCALE(tmp10 + tmp2, S));
        o[3] = range_limit(DESCALE(tmp10 - tmp2, S));
        o[1] = range_limit(DESCALE(tmp12 + tmp0, S));
        o[2] = range_limit(DESCALE(tmp12 - tmp0, S));
    }
}

void jo_idct_2x2_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[8 * 2];
    for (int col = 0; col < 8; col++) {
        if (col == 2 || col == 4 || col == 6) continue;      /* even columns other than 0 are not used */
        const int16_t* c = in + col;
        const uint16_t* qq = q + col;
        int32_t* w = ws + col;
        if (c[8] == 0 && c[24] == 0 && c[40] == 0 && c[56] == 0) {
            int32_t dc = (int32_t)((uint32_t)(c[0] * qq[0]) << PASS1_BITS);
            w[0] = w[8] = dc;
            continue;
        }
        int32_t tmp10 = (int32_t)((uint32_t)(c[0] * qq[0]) << (CONST_BITS + 2));
        int32_t tmp0 = (c[56] * qq[56]) * (-FIX_0_720959822) + (c[40] * qq[40]) * FIX_0_850430095
                       + (c[24] * qq[24]) * (-FIX_1_272758580) + (c[8] * qq[8]) * FIX_3_624509785;
        w[0] = DESCALE(tmp10 + tmp0, CONST_BITS - PASS1_BITS + 2);
        w[8] = DESCALE(tmp10 - tmp0, CONST_BITS - PASS1_BITS + 2);
    }
    for (int row = 0; row < 2; row++) {
        const int32_t* w = ws + 8 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t tmp10 = (int32_t)((uint32_t)w[0] << (CONST_BITS + 2));
        int32_t tmp0 = w[7] * (-FIX_0_720959822) + w[5] * FIX_0_850430095 + w[3] * (-FIX_1_272758580) + w[1] * FIX_3_624509785;
        const int S = CONST_BITS + PASS1_BITS + 3 + 2;
        o[0] = range_limit(DESCALE(tmp10 + tmp0, S));
        o[1] = range_limit(DESCALE(tmp10 - tmp0, S));
    }
}

void jo_idct_1x1_block(const int16_t* in, const uint16_t* q, uint8_t* out) {
    out[0] = range_limit(DESCALE((int32_t)in[0] * (int32_t)q[0], 3));
}

/* ------------------------------------------------------------------------------------------------ */
/* libjpeg's other scaled inverse DCTs (jidctint.c: jpeg_idct_3x3 / 5x5 / 6x6 / 10x10 / 12x12), the  */
/* block routines jddctmgr.c installs for DCT_scaled_size 3, 5, 6, 10, 12: what scale_num 3, 5, 6    */
/* need (luma NxN; a 2x2 sub-sampled chroma component takes 2N x 2N, jdmaster.c).  No zero-column     */
/* shortcut in these; plain arithmetic shifts, the rounding constant is added to the DC term.         */
/* Pin: tests/golden/jpeg_scaled_cases.npz, recorded from the system libjpeg-turbo 2.1.2 by           */
/* tests/golden/make_jpeg_scaled_golden.py (mozjpeg-sys derives from the same jidctint.c).            */
/* ------------------------------------------------------------------------------------------------ */
#define FIXC(x) ((int32_t)((x) * 8192.0 + 0.5))
#define DQ(k) ((int32_t)c[8 * (k)] * (int32_t)qq[8 * (k)])
#define RS(x, n) ((x) >> (n))
#define P1 (CONST_BITS - PASS1_BITS)
#define P2 (CONST_BITS + PASS1_BITS + 3)

void jo_idct_3x3_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[3 * 3];
    for (int col = 0; col < 3; col++) {
        const int16_t* c = in + col; const uint16_t* qq = q + col;
        int32_t tmp0 = (int32_t)((uint32_t)DQ(0) << CONST_BITS) + (1 << (P1 - 1));
        int32_t tmp2 = DQ(2);
        int32_t tmp12 = tmp2 * FIXC(0.707106781);
        int32_t tmp10 = tmp0 + tmp12;
        tmp2 = tmp0 - tmp12 - tmp12;
        tmp12 = DQ(1);
        tmp0 = tmp12 * FIXC(1.224744871);
        ws[3 * 0 + col] = RS(tmp10 + tmp0, P1);
        ws[3 * 2 + col] = RS(tmp10 - tmp0, P1);
        ws[3 * 1 + col] = RS(tmp2, P1);
    }
    for (int row = 0; row < 3; row++) {
        const int32_t* w = ws + 3 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t tmp0 = (int32_t)((uint32_t)(w[0] + (1 << (PASS1_BITS + 2))) << CONST_BITS);
        int32_t tmp2 = w[2];
        int32_t tmp12 = tmp2 * FIXC(0.707106781);
        int32_t tmp10 = tmp0 + tmp12;
        tmp2 = tmp0 - tmp12 - tmp12;
        tmp12 = w[1];
        tmp0 = tmp12 * FIXC(1.224744871);
        o[0] = range_limit(RS(tmp10 + tmp0, P2));
        o[2] = range_limit(RS(tmp10 - tmp0, P2));
        o[1] = range_limit(RS(tmp2, P2));
    }
}

static inline void idct5_core(int32_t tmp12, int32_t e2, int32_t e4, int32_t o1, int32_t o3, int32_t r[5]) {
    int32_t z1 = (e2 + e4) * FIXC(0.790569415);
    int32_t z2 = (e2 - e4) * FIXC(0.353553391);
    int32_t z3 = tmp12 + z2;
    int32_t tmp10 = z3 + z1, tmp11 = z3 - z1;
    tmp12 -= (int32_t)((uint32_t)z2 << 2);
    z1 = (o1 + o3) * FIXC(0.831253876);
    int32_t tmp0 = z1 + o1 * FIXC(0.513743148);
    int32_t tmp1 = z1 - o3 * FIXC(2.176250899);
    r[0] = tmp10 + tmp0; r[4] = tmp10 - tmp0; r[1] = tmp11 + tmp1; r[3] = tmp11 - tmp1; r[2] = tmp12;
}
void jo_idct_5x5_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[5 * 5], r[5];
    for (int col = 0; col < 5; col++) {
        const int16_t* c = in + col; const uint16_t* qq = q + col;
        idct5_core((int32_t)((uint32_t)DQ(0) << CONST_BITS) + (1 << (P1 - 1)), DQ(2), DQ(4), DQ(1), DQ(3), r);
        for (int k = 0; k < 5; k++) ws[5 * k + col] = RS(r[k], P1);
    }
    for (int row = 0; row < 5; row++) {
        const int32_t* w = ws + 5 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        idct5_core((int32_t)((uint32_t)(w[0] + (1 << (PASS1_BITS + 2))) << CONST_BITS), w[2], w[4], w[1], w[3], r);
        for (int k = 0; k < 5; k++) o[k] = range_limit(RS(r[k], P2));
    }
}

void jo_idct_6x6_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[6 * 6];
    for (int col = 0; col < 6; col++) {
        const int16_t* c = in + col; const uint16_t* qq = q + col;
        int32_t tmp0 = (int32_t)((uint32_t)DQ(0) << CONST_BITS) + (1 << (P1 - 1));
        int32_t tmp2 = DQ(4);
        int32_t tmp10 = tmp2 * FIXC(0.707106781);
        int32_t tmp1 = tmp0 + tmp10;
        int32_t tmp11 = RS(tmp0 - tmp10 - tmp10, P1);
        tmp10 = DQ(2);
        tmp0 = tmp10 * FIXC(1.224744871);
        tmp10 = tmp1 + tmp0;
        int32_t tmp12 = tmp1 - tmp0;
        int32_t z1 = DQ(1), z2 = DQ(3), z3 = DQ(5);
        tmp1 = (z1 + z3) * FIXC(0.366025404);
        tmp0 = tmp1 + (int32_t)((uint32_t)(z1 + z2) << CONST_BITS);
        tmp2 = tmp1 + (int32_t)((uint32_t)(z3 - z2) << CONST_BITS);
        tmp1 = (int32_t)((uint32_t)(z1 - z2 - z3) << PASS1_BITS);
        ws[6 * 0 + col] = RS(tmp10 + tmp0, P1);
        ws[6 * 5 + col] = RS(tmp10 - tmp0, P1);
        ws[6 * 1 + col] = tmp11 + tmp1;
        ws[6 * 4 + col] = tmp11 - tmp1;
        ws[6 * 2 + col] = RS(tmp12 + tmp2, P1);
        ws[6 * 3 + col] = RS(tmp12 - tmp2, P1);
    }
    for (int row = 0; row < 6; row++) {
        const int32_t* w = ws + 6 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t tmp0 = (int32_t)((uint32_t)(w[0] + (1 << (PASS1_BITS + 2))) << CONST_BITS);
        int32_t tmp2 = w[4];
        int32_t tmp10 = tmp2 * FIXC(0.707106781);
        int32_t tmp1 = tmp0 + tmp10;
        int32_t tmp11 = tmp0 - tmp10 - tmp10;
        tmp10 = w[2];
        tmp0 = tmp10 * FIXC(1.224744871);
        tmp10 = tmp1 + tmp0;
        int32_t tmp12 = tmp1 - tmp0;
        int32_t z1 = w[1], z2 = w[3], z3 = w[5];
        tmp1 = (z1 + z3) * FIXC(0.366025404);
        tmp0 = tmp1 + (int32_t)((uint32_t)(z1 + z2) << CONST_BITS);
        tmp2 = tmp1 + (int32_t)((uint32_t)(z3 - z2) << CONST_BITS);
        tmp1 = (int32_t)((uint32_t)(z1 - z2 - z3) << CONST_BITS);
        o[0] = range_limit(RS(tmp10 + tmp0, P2));
        o[5] = range_limit(RS(tmp10 - tmp0, P2));
        o[1] = range_limit(RS(tmp11 + tmp1, P2));
        o[4] = range_limit(RS(tmp11 - tmp1, P2));
        o[2] = range_limit(RS(tmp12 + tmp2, P2));
        o[3] = range_limit(RS(tmp12 - tmp2, P2));
    }
}

/* 10-point kernel: z3 = DC term (already shifted, rounding added), e2/e4/e6 even inputs, o1/o3/o7 odd inputs,
 * o5s = input 5 shifted left by CONST_BITS; mid = the exact middle pair's odd term; r[] = the 10 unshifted sums
 * except r[2]/r[7], which pass 1 forms from pre-shifted halves (returned through t22/t12). */
typedef struct { int32_t s[10]; int32_t t22, t12; } idct10_out;
static inline void idct10_core(int32_t z3, int32_t e2, int32_t e4, int32_t e6, int32_t o1, int32_t o3, int32_t o5s,
                               int32_t o7, idct10_out* R) {
    int32_t z4 = e4;
    int32_t z1 = z4 * FIXC(1.144122806);
    int32_t z2 = z4 * FIXC(0.437016024);
    int32_t tmp10 = z3 + z1, tmp11 = z3 - z2;
    R->t22 = z3 - (int32_t)((uint32_t)(z1 - z2) << 1);
    z2 = e2; int32_t z3b = e6;
    z1 = (z2 + z3b) * FIXC(0.831253876);
    int32_t tmp12 = z1 + z2 * FIXC(0.513743148);
    int32_t tmp13 = z1 - z3b * FIXC(2.176250899);
    int32_t tmp20 = tmp10 + tmp12, tmp24 = tmp10 - tmp12, tmp21 = tmp11 + tmp13, tmp23 = tmp11 - tmp13;
    z1 = o1; z2 = o3; z4 = o7;
    tmp11 = z2 + z4;
    tmp13 = z2 - z4;
    tmp12 = tmp13 * FIXC(0.309016994);
    z2 = tmp11 * FIXC(0.951056516);
    z4 = o5s + tmp12;
    tmp10 = z1 * FIXC(1.396802247) + z2 + z4;
    int32_t tmp14 = z1 * FIXC(0.221231742) - z2 + z4;
    z2 = tmp11 * FIXC(0.587785252);
    z4 = o5s - tmp12 - (int32_t)((uint32_t)tmp13 << (CONST_BITS - 1));
    R->t12 = z1 - tmp13;                                      /* caller finishes: pass 1 (.. - z3) << PASS1_BITS, pass 2 (<< CONST_BITS) - z3s */
    tmp11 = z1 * FIXC(1.260073511) - z2 - z4;
    tmp13 = z1 * FIXC(0.642039522) - z2 + z4;
    R->s[0] = tmp20 + tmp10; R->s[9] = tmp20 - tmp10;
    R->s[1] = tmp21 + tmp11; R->s[8] = tmp21 - tmp11;
    R->s[3] = tmp23 + tmp13; R->s[6] = tmp23 - tmp13;
    R->s[4] = tmp24 + tmp14; R->s[5] = tmp24 - tmp14;
}
void jo_idct_10x10_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[8 * 10];
    idct10_out R;
    for (int col = 0; col < 8; col++) {
        const int16_t* c = in + col; const uint16_t* qq = q + col;
        int32_t z5 = DQ(5);
        idct10_core((int32_t)((uint32_t)DQ(0) << CONST_BITS) + (1 << (P1 - 1)), DQ(2), DQ(4), DQ(6), DQ(1), DQ(3),
                    (int32_t)((uint32_t)z5 << CONST_BITS), DQ(7), &R);
        int32_t tmp22 = RS(R.t22, P1);
        int32_t tmp12 = (int32_t)((uint32_t)(R.t12 - z5) << PASS1_BITS);
        ws[8 * 0 + col] = RS(R.s[0], P1); ws[8 * 9 + col] = RS(R.s[9], P1);
        ws[8 * 1 + col] = RS(R.s[1], P1); ws[8 * 8 + col] = RS(R.s[8], P1);
        ws[8 * 2 + col] = tmp22 + tmp12;  ws[8 * 7 + col] = tmp22 - tmp12;
        ws[8 * 3 + col] = RS(R.s[3], P1); ws[8 * 6 + col] = RS(R.s[6], P1);
        ws[8 * 4 + col] = RS(R.s[4], P1); ws[8 * 5 + col] = RS(R.s[5], P1);
    }
    for (int row = 0; row < 10; row++) {
        const int32_t* w = ws + 8 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        int32_t z5s = (int32_t)((uint32_t)w[5] << CONST_BITS);
        idct10_core((int32_t)((uint32_t)(w[0] + (1 << (PASS1_BITS + 2))) << CONST_BITS), w[2], w[4], w[6], w[1], w[3], z5s, w[7], &R);
        int32_t tmp12 = (int32_t)((uint32_t)R.t12 << CONST_BITS) - z5s;
        o[0] = range_limit(RS(R.s[0], P2)); o[9] = range_limit(RS(R.s[9], P2));
        o[1] = range_limit(RS(R.s[1], P2)); o[8] = range_limit(RS(R.s[8], P2));
        o[2] = range_limit(RS(R.t22 + tmp12, P2)); o[7] = range_limit(RS(R.t22 - tmp12, P2));
        o[3] = range_limit(RS(R.s[3], P2)); o[6] = range_limit(RS(R.s[6], P2));
        o[4] = range_limit(RS(R.s[4], P2)); o[5] = range_limit(RS(R.s[5], P2));
    }
}

/* 12-point kernel; e2s/e6s are inputs 2 and 6 shifted left by CONST_BITS, e2 the unshifted input 2. */
static inline void idct12_core(int32_t z3, int32_t e2, int32_t e4, int32_t e6, int32_t o1, int32_t o3, int32_t o5,
                               int32_t o7, int32_t r[12]) {
    int32_t z4 = e4 * FIXC(1.224744871);
    int32_t tmp10 = z3 + z4, tmp11 = z3 - z4;
    int32_t z1 = e2;
    z4 = z1 * FIXC(1.366025404);
    z1 = (int32_t)((uint32_t)z1 << CONST_BITS);
    int32_t z2 = (int32_t)((uint32_t)e6 << CONST_BITS);
    int32_t tmp12 = z1 - z2;
    int32_t tmp21 = z3 + tmp12, tmp24 = z3 - tmp12;
    tmp12 = z4 + z2;
    int32_t tmp20 = tmp10 + tmp12, tmp25 = tmp10 - tmp12;
    tmp12 = z4 - z1 - z2;
    int32_t tmp22 = tmp11 + tmp12, tmp23 = tmp11 - tmp12;
    z1 = o1; z2 = o3; int32_t z3o = o5; z4 = o7;
    tmp11 = z2 * FIXC(1.306562965);
    int32_t tmp14 = z2 * (-FIX_0_541196100);
    tmp10 = z1 + z3o;
    int32_t tmp15 = (tmp10 + z4) * FIXC(0.860918669);
    tmp12 = tmp15 + tmp10 * FIXC(0.261052384);
    tmp10 = tmp12 + tmp11 + z1 * FIXC(0.280143716);
    int32_t tmp13 = (z3o + z4) * (-FIXC(1.045510580));
    tmp12 += tmp13 + tmp14 - z3o * FIXC(1.478575242);
    tmp13 += tmp15 - tmp11 + z4 * FIXC(1.586706681);
    tmp15 += tmp14 - z1 * FIXC(0.676326758) - z4 * FIXC(1.982889723);
    z1 -= z4;
    z2 -= z3o;
    z3o = (z1 + z2) * FIX_0_541196100;
    tmp11 = z3o + z1 * FIX_0_765366865;
    tmp14 = z3o - z2 * FIX_1_847759065;
    r[0] = tmp20 + tmp10; r[11] = tmp20 - tmp10;
    r[1] = tmp21 + tmp11; r[10] = tmp21 - tmp11;
    r[2] = tmp22 + tmp12; r[9] = tmp22 - tmp12;
    r[3] = tmp23 + tmp13; r[8] = tmp23 - tmp13;
    r[4] = tmp24 + tmp14; r[7] = tmp24 - tmp14;
    r[5] = tmp25 + tmp15; r[6] = tmp25 - tmp15;
}
void jo_idct_12x12_block(const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    int32_t ws[8 * 12], r[12];
    for (int col = 0; col < 8; col++) {
        const int16_t* c = in + col; const uint16_t* qq = q + col;
        idct12_core((int32_t)((uint32_t)DQ(0) << CONST_BITS) + (1 << (P1 - 1)), DQ(2), DQ(4), DQ(6), DQ(1), DQ(3), DQ(5), DQ(7), r);
        for (int k = 0; k < 12; k++) ws[8 * k + col] = RS(r[k], P1);
    }
    for (int row = 0; row < 12; row++) {
        const int32_t* w = ws + 8 * row;
        uint8_t* o = out + (size_t)row * out_stride;
        idct12_core((int32_t)((uint32_t)(w[0] + (1 << (PASS1_BITS + 2))) << CONST_BITS), w[2], w[4], w[6], w[1], w[3], w[5], w[7], r);
        for (int k = 0; k < 12; k++) o[k] = range_limit(RS(r[k], P2));
    }
}

/* jddctmgr.c: the block routine for a DCT_scaled_size (the sizes this path can ask for) */
int jo_idct_scaled_block(int n, const int16_t* in, const uint16_t* q, uint8_t* out, int out_stride) {
    switch (n) {
    case 1: jo_idct_1x1_block(in, q, out); return 0;
    case 2: jo_idct_2x2_block(in, q, out, out_stride); return 0;
    case 3: jo_idct_3x3_block(in, q, out, out_stride); return 0;
    case 4: jo_idct_4x4_block(in, q, out, out_stride); return 0;
    case 5: jo_idct_5x5_block(in, q, out, out_stride); return 0;
    case 6: jo_idct_6x6_block(in, q, out, out_stride); return 0;
    case 8: jo_idct_islow_block(in, q, out, out_stride); return 0;
    case 10: jo_idct_10x10_block(in, q, out, out_stride); return 0;
    case 12: jo_idct_12x12_block(in, q, out, out_stride); return 0;
    default: return -1;
    }
}

/* flow_scale_spatial[_srgb]_NxN semantics over caller-provided tables (tests load them from
 * tests/golden/block_scaler_tables.npz, i.e. from the reference's own file). */
void jo_scale_spatial_block(const uint8_t* in /* 8x8, stride in_stride */, int in_stride, int n, int srgb,
                            const int8_t* w7x8, const uint8_t* log2div7, const uint16_t* s2l, const uint8_t* l2s,
                            uint8_t* out, int out_stride) {
    for (int r = 0; r < n; r++) {
        int32_t v[8];
        for (int j = 0; j < 8; j++) {
            int32_t s = 0;
            for (int i = 0; i < 8; i++) {
                int32_t p = in[i * in_stride + j];
                s += (int32_t)w7x8[r * 8 + i] * (srgb ? (int32_t)s2l[p] : p);
            }
            v[j] = s;
        }
        for (int c = 0; c < n; c++) {
            int sh = log2div7[r] + log2div7[c];
            int32_t sum = (int32_t)1 << (sh - 1);
            for (int j = 0; j < 8; j++) sum += v[j] * (int32_t)w7x8[c * 8 + j];
            uint8_t o;
            if (sum < 0) o = 0;
            else if ((uint32_t)sum >= ((uint32_t)4096 << sh)) o = 255;
            else o = srgb ? l2s[sum >> sh] : (uint8_t)(sum >> sh);
            out[r * out_stride + c] = o;
        }
    }
}

/*
 * Scaled pixel stage: scale_num in {1..6, 8}; luma_mode: 0 libjpeg's own scaled IDCT, 1 flow_scale_spatial,
 * 2 flow_scale_spatial_srgb (the reference's default when scaled; luma only, codec_jpeg_wrapper.c:277-278).
 * Supported: grayscale, 4:4:4, 4:2:2 (h2v1), 4:4:0 (h1v2), 4:2:0.
 *   jdmaster.c (jpeg_calc_output_dimensions): output = ceil(dim * scale_num / 8); a component's IDCT size starts at
 *   scale_num and doubles while it stays < 8 before doubling and both sampling ratios allow it -- so 4:2:0 chroma
 *   decodes at 2*scale_num with no up-sampling, 4:2:2 / 4:4:0 chroma stays at scale_num and is up-sampled.
 *   jdsample.c (jinit_upsampler): fancy (triangle) up-sampling needs do_fancy_upsampling (libjpeg's default, the
 *   reference never changes it) AND min_DCT_scaled_size > 1; the h2v1 / h2v2 fancy forms also need
 *   downsampled_width > 2, otherwise samples are replicated.
 */
static int comp_idct_size(int scale_num, int hs_c, int vs_c, int hmax, int vmax) {
    int ssize = scale_num;
    while (ssize < 8 && (hmax * scale_num) % (hs_c * ssize * 2) == 0 && (vmax * scale_num) % (vs_c * ssize * 2) == 0) ssize *= 2;
    return ssize;
}

int jo_jpeg_idct_color_scaled(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                              int ncomp, const uint8_t* hs, const uint8_t* vs, uint32_t width, uint32_t height,
                              int scale_num, int luma_mode,
                              const int8_t* w7x8, const uint8_t* log2div7, const uint16_t* s2l12, const uint8_t* l2s12,
                              uint8_t* bgra, uint32_t stride) {
    build_ycc();
    if (scale_num < 1 || scale_num > 8 || scale_num == 7) return JO_ERR_UNSUPPORTED;
    const int16_t* coef[3] = {coef0, coef1, coef2};
    int hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; c++) { if (hs[c] > hmax) hmax = hs[c]; if (vs[c] > vmax) vmax = vs[c]; }
    if (ncomp == 3 && !(hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1 && hs[0] == hmax && vs[0] == vmax && hmax <= 2 && vmax <= 2))
        return JO_ERR_UNSUPPORTED;
    if (ncomp != 1 && ncomp != 3) return JO_ERR_UNSUPPORTED;
    if (luma_mode != 0 && scale_num == 8) luma_mode = 0;                     /* the selector only acts when scaled < 8 */
    const uint32_t N = (uint32_t)scale_num;
    uint32_t ow = (width * N + 7) / 8, oh = (height * N + 7) / 8;
    uint32_t mw = (width + 8u * hmax - 1) / (8u * hmax), mh = (height + 8u * vmax - 1) / (8u * vmax);
    const int fancy = scale_num > 1;
    uint8_t* plane[3] = {0, 0, 0};
    uint32_t pw[3], dw[3], dh[3];
    int ux[3], uy[3];                                                        /* up-sampling factors left to do (1 or 2) */
    for (int c = 0; c < ncomp; c++) {
        int n = comp_idct_size(scale_num, hs[c], vs[c], hmax, vmax);
        uint32_t bw = mw * hs[c], bh = mh * vs[c];
        pw[c] = bw * (uint32_t)n;
        ux[c] = (hmax * scale_num) / (hs[c] * n);
        uy[c] = (vmax * scale_num) / (vs[c] * n);
        dw[c] = (width * hs[c] * (uint32_t)n + (uint32_t)hmax * 8 - 1) / ((uint32_t)hmax * 8);      /* downsampled_width  */
        dh[c] = (height * vs[c] * (uint32_t)n + (uint32_t)vmax * 8 - 1) / ((uint32_t)vmax * 8);     /* downsampled_height */
        plane[c] = (uint8_t*)malloc((size_t)pw[c] * bh * n);
        if (!plane[c]) { for (int k = 0; k < c; k++) free(plane[k]); return JO_ERR_ALLOC; }
        for (uint32_t by = 0; by < bh; by++)
            for (uint32_t bx = 0; bx < bw; bx++) {
                const int16_t* blk = coef[c] + 64 * ((size_t)by * bw + bx);
                uint8_t* dst = plane[c] + (size_t)by * n * pw[c] + (size_t)bx * n;
                if (c == 0 && luma_mode != 0) {
                    uint8_t full[64];
                    jo_idct_islow_block(blk, qt, full, 8);
                    jo_scale_spatial_block(full, 8, n, luma_mode == 2, w7x8, log2div7, s2l12, l2s12, dst, (int)pw[c]);
                } else if (jo_idct_scaled_block(n, blk, qt + 64 * c, dst, (int)pw[c])) {
                    for (int k = 0; k <= c; k++) free(plane[k]);
                    return JO_ERR_UNSUPPORTED;
                }
            }
    }
    for (uint32_t y = 0; y < oh; y++) {
        uint8_t* o = bgra + (size_t)y * stride;
        for (uint32_t x = 0; x < ow; x++) {
            int32_t Y = plane[0][(size_t)y * pw[0] + x];
            if (ncomp == 1) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = (uint8_t)Y; o[4 * x + 3] = 255; continue; }
            int32_t v[2];
            for (int k = 0; k < 2; k++) {
                const int c = 1 + k;
                const uint8_t* P = plane[c];
                const uint32_t W = pw[c], DW = dw[c], DH = dh[c];
                if (ux[c] == 1 && uy[c] == 1) v[k] = P[(size_t)y * W + x];
                else if (ux[c] == 2 && uy[c] == 1) {                          /* h2v1 */
                    int32_t cx = (int32_t)(x >> 1);
                    if (!(fancy && DW > 2)) v[k] = cs(P, W, DW, DH, cx, (int32_t)y);
                    else if (x == 0 || (x == 2 * DW - 1)) v[k] = cs(P, W, DW, DH, cx, (int32_t)y);
                    else if (x & 1) v[k] = (3 * cs(P, W, DW, DH, cx, (int32_t)y) + cs(P, W, DW, DH, cx + 1, (int32_t)y) + 2) >> 2;
                    else v[k] = (3 * cs(P, W, DW, DH, cx, (int32_t)y) + cs(P, W, DW, DH, cx - 1, (int32_t)y) + 1) >> 2;
                } else if (ux[c] == 1 && uy[c] == 2) {                        /* h1v2 */
                    int32_t cy = (int32_t)(y >> 1);
                    if (!fancy) v[k] = cs(P, W, DW, DH, (int32_t)x, cy);
                    else if (y & 1) v[k] = (3 * cs(P, W, DW, DH, (int32_t)x, cy) + cs(P, W, DW, DH, (int32_t)x, cy + 1) + 2) >> 2;
                    else v[k] = (3 * cs(P, W, DW, DH, (int32_t)x, cy) + cs(P, W, DW, DH, (int32_t)x, cy - 1) + 1) >> 2;
                } else {                                                       /* h2v2 */
                    int32_t cx = (int32_t)(x >> 1), cy = (int32_t)(y >> 1);
                    if (!(fancy && DW > 2)) { v[k] = cs(P, W, DW, DH, cx, cy); continue; }
                    int32_t ny = (y & 1) ? cy + 1 : cy - 1;
                    int32_t thiscol = 3 * cs(P, W, DW, DH, cx, cy) + cs(P, W, DW, DH, cx, ny);
                    if ((x & 1) == 0) {
                        if (cx == 0) v[k] = (thiscol * 4 + 8) >> 4;
                        else v[k] = (thiscol * 3 + 3 * cs(P, W, DW, DH, cx - 1, cy) + cs(P, W, DW, DH, cx - 1, ny) + 8) >> 4;
                    } else {
                        if (cx == (int32_t)DW - 1) v[k] = (thiscol * 4 + 7) >> 4;
                        else v[k] = (thiscol * 3 + 3 * cs(P, W, DW, DH, cx + 1, cy) + cs(P, W, DW, DH, cx + 1, ny) + 7) >> 4;
                    }
                }
            }
            o[4 * x + 2] = clamp255(Y + cr_r[v[1]]);
            o[4 * x + 1] = clamp255(Y + ((cb_g[v[0]] + cr_g[v[1]]) >> 16));
            o[4 * x + 0] = clamp255(Y + cb_b[v[0]]);
            o[4 * x + 3] = 255;
        }
    }
    for (int c = 0; c < ncomp; c++) free(plane[c]);
    return JO_OK;
}

/* ------------------------------------------------------------------------------------------------ */
/* Encode-side pixel stage: BGRA -> YCbCr -> chroma down-sampling -> forward   */
/* DCT (islow) -> quantisation -> coefficient planes.  What libjpeg runs before entropy coding when  */
/* imageflow encodes with its classic preset (codecs/mozjpeg.rs:78-160: set_fastest_defaults, i.e.   */
/* no trellis; input colour space JCS_EXT_BGRA/BGRX).  Restated from the published IJG algorithms:    */
/*   jccolor.c rgb_ycc_convert, jcsample.c fullsize/h2v1/h2v2_downsample (+ edge expansion),          */
/*   jfdctint.c jpeg_fdct_islow, jcdctmgr.c quantize (round-half-up division by 8*Q),                 */
/*   jccoefct.c compress_data dummy blocks at the right/bottom MCU edges (AC = 0, DC = previous).     */
/* PIN: tests encode with Pillow/libjpeg-turbo (optimize=False), entropy-decode the file with the     */
/* decoder above and require these coefficient planes to be identical.                                */
/* ------------------------------------------------------------------------------------------------ */
static void fdct_islow_block(const uint8_t* s, int stride, int32_t* out) {
    int32_t ws[64];
    for (int r = 0; r < 8; r++) {
        const uint8_t* d = s + (size_t)r * stride;
        int32_t e[8];
        for (int k = 0; k < 8; k++) e[k] = (int32_t)d[k] - 128;
        int32_t tmp0 = e[0] + e[7], tmp7 = e[0] - e[7], tmp1 = e[1] + e[6], tmp6 = e[1] - e[6];
        int32_t tmp2 = e[2] + e[5], tmp5 = e[2] - e[5], tmp3 = e[3] + e[4], tmp4 = e[3] - e[4];
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        int32_t* w = ws + 8 * r;
        w[0] = (int32_t)((uint32_t)(tmp10 + tmp11) << PASS1_BITS);
        w[4] = (int32_t)((uint32_t)(tmp10 - tmp11) << PASS1_BITS);
        int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
        w[2] = DESCALE(z1 + tmp13 * FIX_0_765366865, CONST_BITS - PASS1_BITS);
        w[6] = DESCALE(z1 + tmp12 * (-FIX_1_847759065), CONST_BITS - PASS1_BITS);
        z1 = tmp4 + tmp7;
        int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
        int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp4 *= FIX_0_298631336; tmp5 *= FIX_2_053119869; tmp6 *= FIX_3_072711026; tmp7 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        w[7] = DESCALE(tmp4 + z1 + z3, CONST_BITS - PASS1_BITS);
        w[5] = DESCALE(tmp5 + z2 + z4, CONST_BITS - PASS1_BITS);
        w[3] = DESCALE(tmp6 + z2 + z3, CONST_BITS - PASS1_BITS);
        w[1] = DESCALE(tmp7 + z1 + z4, CONST_BITS - PASS1_BITS);
    }
    for (int c = 0; c < 8; c++) {
        const int32_t* w = ws + c;
        int32_t tmp0 = w[0] + w[56], tmp7 = w[0] - w[56], tmp1 = w[8] + w[48], tmp6 = w[8] - w[48];
        int32_t tmp2 = w[16] + w[40], tmp5 = w[16] - w[40], tmp3 = w[24] + w[32], tmp4 = w[24] - w[32];
        int32_t tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
        int32_t* o = out + c;
        o[0] = DESCALE(tmp10 + tmp11, PASS1_BITS);
        o[32] = DESCALE(tmp10 - tmp11, PASS1_BITS);
        int32_t z1 = (tmp12 + tmp13) * FIX_0_541196100;
        o[16] = DESCALE(z1 + tmp13 * FIX_0_765366865, CONST_BITS + PASS1_BITS);
        o[48] = DESCALE(z1 + tmp12 * (-FIX_1_847759065), CONST_BITS + PASS1_BITS);
        z1 = tmp4 + tmp7;
        int32_t z2 = tmp5 + tmp6, z3 = tmp4 + tmp6, z4 = tmp5 + tmp7;
        int32_t z5 = (z3 + z4) * FIX_1_175875602;
        tmp4 *= FIX_0_298631336; tmp5 *= FIX_2_053119869; tmp6 *= FIX_3_072711026; tmp7 *= FIX_1_501321110;
        z1 *= -FIX_0_899976223; z2 *= -FIX_2_562915447; z3 *= -FIX_1_961570560; z4 *= -FIX_0_390180644;
        z3 += z5; z4 += z5;
        o[56] = DESCALE(tmp4 + z1 + z3, CONST_BITS + PASS1_BITS);
        o[40] = DESCALE(tmp5 + z2 + z4, CONST_BITS + PASS1_BITS);
        o[24] = DESCALE(tmp6 + z2 + z3, CONST_BITS + PASS1_BITS);
        o[8]  = DESCALE(tmp7 + z1 + z4, CONST_BITS + PASS1_BITS);
    }
}

/*
 * bgra: height rows of `stride` bytes (alpha ignored, as JCS_EXT_BGRX).  hs/vs: sampling (4:4:4, 4:2:2 h2v1, 4:2:0).
 * qt: [3][64] natural order.  coef[c]: int16 [bh_c][bw_c][64] natural order, MCU padded (as jo_jpeg_read_coefficients).
 */
int jo_jpeg_forward(const uint8_t* bgra, uint32_t width, uint32_t height, uint32_t stride, int ncomp,
                    const uint8_t* hs, const uint8_t* vs, const uint16_t* qt,
                    int16_t* coef0, int16_t* coef1, int16_t* coef2) {
    if (ncomp != 3) return JO_ERR_UNSUPPORTED;
    int hmax = hs[0], vmax = vs[0];
    if (!(hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1 && (hmax == 1 || hmax == 2) && (vmax == 1 || vmax == 2)) ||
        (hmax == 1 && vmax == 2))
        return JO_ERR_UNSUPPORTED;
    int16_t* coef[3] = {coef0, coef1, coef2};
    uint32_t mw = (width + 8u * hmax - 1) / (8u * hmax), mh = (height + 8u * vmax - 1) / (8u * vmax);
    for (int c = 0; c < 3; c++) {
        uint32_t bw = mw * hs[c], bh = mh * vs[c];
        uint32_t dw = (width * hs[c] + hmax - 1) / hmax, dh = (height * vs[c] + vmax - 1) / vmax;   /* downsampled size */
        uint32_t rbw = (dw + 7) / 8, rbh = (dh + 7) / 8;                                             /* real blocks */
        uint32_t pw = rbw * 8, ph = rbh * 8;
        uint8_t* plane = (uint8_t*)malloc((size_t)pw * ph);
        if (!plane) return JO_ERR_ALLOC;
        int fx = hmax / hs[c], fy = vmax / vs[c];                     /* source pixels per sample: 1 or 2 */
        for (uint32_t y = 0; y < ph; y++) {
            if (y >= dh) {          /* jcprepct.c: rows below the last down-sampled row repeat that row */
                memcpy(plane + (size_t)y * pw, plane + (size_t)(dh - 1) * pw, pw);
                continue;
            }
            for (uint32_t x = 0; x < pw; x++) {
                int32_t sum = 0;
                for (int dy = 0; dy < fy; dy++)
                    for (int dx = 0; dx < fx; dx++) {
                        uint32_t sx = x * (uint32_t)fx + (uint32_t)dx, sy = y * (uint32_t)fy + (uint32_t)dy;
                        if (sx >= width) sx = width - 1;               /* expand_right_edge / expand_bottom_edge */
                        if (sy >= height) sy = height - 1;
                        const uint8_t* p = bgra + (size_t)sy * stride + (size_t)sx * 4;
                        int32_t b = p[0], g = p[1], r = p[2], v;
                        if (c == 0) v = (19595 * r + 38470 * g + 7471 * b + 32768) >> 16;
                        else if (c == 1) v = (-11059 * r - 21709 * g + 32768 * b + (128 << 16) + 32767) >> 16;
                        else v = (32768 * r - 27439 * g - 5329 * b + (128 << 16) + 32767) >> 16;
                        sum += v;
                    }
                int32_t v;
                if (fx == 1 && fy == 1) v = sum;
                else if (fy == 1) v = (sum + (int32_t)(x & 1)) >> 1;          /* h2v1: bias 0,1,0,1 */
                else v = (sum + 1 + (int32_t)(x & 1)) >> 2;                    /* h2v2: bias 1,2,1,2 */
                plane[(size_t)y * pw + x] = (uint8_t)v;
            }
        }
        const uint16_t* q = qt + 64 * c;
        memset(coef[c], 0, sizeof(int16_t) * 64 * (size_t)bw * bh);
        for (uint32_t by = 0; by < rbh; by++)
            for (uint32_t bx = 0; bx < rbw; bx++) {
                int32_t d[64];
                fdct_islow_block(plane + (size_t)by * 8 * pw + bx * 8, (int)pw, d);
                int16_t* o = coef[c] + 64 * ((size_t)by * bw + bx);
                for (int i = 0; i < 64; i++) {
                    int32_t qv = (int32_t)q[i] << 3, t = d[i];
                    if (t < 0) { t = -t; t += qv >> 1; t = (t >= qv) ? t / qv : 0; t = -t; }
                    else { t += qv >> 1; t = (t >= qv) ? t / qv : 0; }
                    o[i] = (int16_t)t;
                }
            }
        free(plane);
        /* dummy blocks (jccoefct.c compress_data): right edge: DC of the block to the left; bottom edge rows of an MCU:
           DC of the previous block in MCU order = the last block of the MCU's previous block row */
        for (uint32_t by = 0; by < rbh; by++)
            for (uint32_t bx = rbw; bx < bw; bx++)
                coef[c][64 * ((size_t)by * bw + bx)] = coef[c][64 * ((size_t)by * bw + bx - 1)];
        for (uint32_t by = rbh; by < bh; by++)
            for (uint32_t bx = 0; bx < bw; bx++) {
                uint32_t mcu_x = bx / hs[c];
                uint32_t prev_bx = mcu_x * hs[c] + hs[c] - 1;          /* last block of the MCU's previous block row */
                coef[c][64 * ((size_t)by * bw + bx)] = coef[c][64 * ((size_t)(by - 1) * bw + prev_bx)];
            }
    }
    return JO_OK;
}
