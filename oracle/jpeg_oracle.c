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
int jo_jpeg_idct_color(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                       int ncomp, const uint8_t* hs, const uint8_t* vs, uint32_t width, uint32_t height,
                       uint8_t* bgra, uint32_t stride) {
    build_ycc();
    const int16_t* coef[3] = {coef0, coef1, coef2};
    int hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; c++) { if (hs[c] > hmax) hmax = hs[c]; if (vs[c] > vmax) vmax = vs[c]; }
    if (ncomp == 3 && !(hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1 && hs[0] == hmax && vs[0] == vmax))
        return JO_ERR_UNSUPPORTED;
    uint32_t mw = (width + 8u * hmax - 1) / (8u * hmax), mh = (height + 8u * vmax - 1) / (8u * vmax);
    uint8_t* plane[3] = {0, 0, 0};
    uint32_t pw[3], ph[3], dw[3], dh[3];
    for (int c = 0; c < ncomp; c++) {
        pw[c] = mw * hs[c] * 8; ph[c] = mh * vs[c] * 8;
        dw[c] = (width * hs[c] + hmax - 1) / hmax;              /* downsampled_width  */
        dh[c] = (height * vs[c] + vmax - 1) / vmax;             /* downsampled_height */
        plane[c] = (uint8_t*)malloc((size_t)pw[c] * ph[c]);
        if (!plane[c]) { for (int k = 0; k < c; k++) free(plane[k]); return JO_ERR_ALLOC; }
        uint32_t bw = mw * hs[c], bh = mh * vs[c];
        for (uint32_t by = 0; by < bh; by++)
            for (uint32_t bx = 0; bx < bw; bx++)
                jo_idct_islow_block(coef[c] + 64 * ((size_t)by * bw + bx), qt + 64 * c,
                                    plane[c] + (size_t)by * 8 * pw[c] + bx * 8, (int)pw[c]);
    }
    for (uint32_t y = 0; y < height; y++) {
        uint8_t* o = bgra + (size_t)y * stride;
        for (uint32_t x = 0; x < width; x++) {
            int32_t Y = plane[0][(size_t)y * pw[0] + x];
            if (ncomp == 1) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = (uint8_t)Y; o[4 * x + 3] = 255; continue; }
            int32_t cbv, crv;
            if (hmax == 1 && vmax == 1) {
                cbv = plane[1][(size_t)y * pw[1] + x]; crv = plane[2][(size_t)y * pw[2] + x];
            } else if (hmax == 2 && vmax == 1) {                  /* h2v1 fancy: 3/4 near + 1/4 far */
                int32_t cx = (int32_t)(x >> 1), far = (x & 1) ? cx + 1 : cx - 1, bias = (x & 1) ? 2 : 1;
                int32_t v[2];
                for (int k = 0; k < 2; k++) {
                    const uint8_t* P = plane[1 + k];
                    if ((x == 0) || (x == 2 * dw[1 + k] - 1 && (x & 1)))
                        v[k] = cs(P, pw[1 + k], dw[1 + k], dh[1 + k], cx, (int32_t)y);
                    else
                        v[k] = (3 * cs(P, pw[1 + k], dw[1 + k], dh[1 + k], cx, (int32_t)y)
                                + cs(P, pw[1 + k], dw[1 + k], dh[1 + k], far, (int32_t)y) + bias) >> 2;
                }
                cbv = v[0]; crv = v[1];
            } else if (hmax == 2 && vmax == 2) {                  /* h2v2 fancy: triangle in both directions */
                int32_t cx = (int32_t)(x >> 1), cy = (int32_t)(y >> 1);
                int32_t ny = (y & 1) ? cy + 1 : cy - 1;           /* the "other" chroma row */
                int32_t v[2];
                for (int k = 0; k < 2; k++) {
                    const uint8_t* P = plane[1 + k];
                    uint32_t W = pw[1 + k], DW = dw[1 + k], DH = dh[1 + k];
                    int32_t thiscol = 3 * cs(P, W, DW, DH, cx, cy) + cs(P, W, DW, DH, cx, ny);
                    if ((x & 1) == 0) {
                        if (cx == 0) v[k] = (thiscol * 4 + 8) >> 4;
                        else {
                            int32_t last = 3 * cs(P, W, DW, DH, cx - 1, cy) + cs(P, W, DW, DH, cx - 1, ny);
                            v[k] = (thiscol * 3 + last + 8) >> 4;
                        }
                    } else {
                        if (cx == (int32_t)DW - 1) v[k] = (thiscol * 4 + 7) >> 4;
                        else {
                            int32_t next = 3 * cs(P, W, DW, DH, cx + 1, cy) + cs(P, W, DW, DH, cx + 1, ny);
                            v[k] = (thiscol * 3 + next + 7) >> 4;
                        }
                    }
                }
                cbv = v[0]; crv = v[1];
            } else { for (int k = 0; k < ncomp; k++) free(plane[k]); return JO_ERR_UNSUPPORTED; }
            o[4 * x + 2] = clamp255(Y + cr_r[crv]);
            o[4 * x + 1] = clamp255(Y + ((cb_g[cbv] + cr_g[crv]) >> 16));
            o[4 * x + 0] = clamp255(Y + cb_b[cbv]);
            o[4 * x + 3] = 255;
        }
    }
    for (int c = 0; c < ncomp; c++) free(plane[c]);
    return JO_OK;
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
 * Reduced-size pixel stage.  scale_num in {1,2,4}; luma_mode: 0 libjpeg's own reduced IDCT, 1 flow_scale_spatial,
 * 2 flow_scale_spatial_srgb (the reference's default when scaled).  Supported: grayscale, 4:4:4, 4:2:0.
 * Output size: ceil(width*scale_num/8) x ceil(height*scale_num/8) (jdmaster.c jpeg_calc_output_dimensions).
 */
int jo_jpeg_idct_color_scaled(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint16_t* qt,
                              int ncomp, const uint8_t* hs, const uint8_t* vs, uint32_t width, uint32_t height,
                              int scale_num, int luma_mode,
                              const int8_t* w7x8, const uint8_t* log2div7, const uint16_t* s2l12, const uint8_t* l2s12,
                              uint8_t* bgra, uint32_t stride) {
    build_ycc();
    if (scale_num != 1 && scale_num != 2 && scale_num != 4) return JO_ERR_UNSUPPORTED;
    const int16_t* coef[3] = {coef0, coef1, coef2};
    int hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; c++) { if (hs[c] > hmax) hmax = hs[c]; if (vs[c] > vmax) vmax = vs[c]; }
    int is420 = ncomp == 3 && hmax == 2 && vmax == 2 && hs[0] == 2 && vs[0] == 2 && hs[1] == 1 && vs[1] == 1 && hs[2] == 1 && vs[2] == 1;
    int is444 = ncomp == 3 && hmax == 1 && vmax == 1;
    if (!(ncomp == 1 || is420 || is444)) return JO_ERR_UNSUPPORTED;
    uint32_t ow = (width * (uint32_t)scale_num + 7) / 8, oh = (height * (uint32_t)scale_num + 7) / 8;
    uint32_t mw = (width + 8u * hmax - 1) / (8u * hmax), mh = (height + 8u * vmax - 1) / (8u * vmax);
    uint8_t* plane[3] = {0, 0, 0};
    uint32_t pw[3];
    for (int c = 0; c < ncomp; c++) {
        /* jdmaster.c: a component sub-sampled 2x in both directions takes an IDCT twice as large, no up-sampling */
        int n = (c > 0 && is420) ? scale_num * 2 : scale_num;
        uint32_t bw = mw * hs[c], bh = mh * vs[c];
        pw[c] = bw * (uint32_t)n;
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
                } else if (n == 8) jo_idct_islow_block(blk, qt + 64 * c, dst, (int)pw[c]);
                else if (n == 4) jo_idct_4x4_block(blk, qt + 64 * c, dst, (int)pw[c]);
                else if (n == 2) jo_idct_2x2_block(blk, qt + 64 * c, dst, (int)pw[c]);
                else jo_idct_1x1_block(blk, qt + 64 * c, dst);
            }
    }
    for (uint32_t y = 0; y < oh; y++) {
        uint8_t* o = bgra + (size_t)y * stride;
        for (uint32_t x = 0; x < ow; x++) {
            int32_t Y = plane[0][(size_t)y * pw[0] + x];
            if (ncomp == 1) { o[4 * x] = o[4 * x + 1] = o[4 * x + 2] = (uint8_t)Y; o[4 * x + 3] = 255; continue; }
            int32_t cbv = plane[1][(size_t)y * pw[1] + x], crv = plane[2][(size_t)y * pw[2] + x];
            o[4 * x + 2] = clamp255(Y + cr_r[crv]);
            o[4 * x + 1] = clamp255(Y + ((cb_g[cbv] + cr_g[crv]) >> 16));
            o[4 * x + 0] = clamp255(Y + cb_b[cbv]);
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
