/*
 * if_oracle.c -- CPU ORACLE for the imageflow pixel hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Nothing in the product (imageflow_amd/, include/, the C-ABI library) may include, link
 * or call this file.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
 * use it, and only as the checker / the timed CPU baseline.
 *
 * What it restates (paths relative to /root/reference/imageflow_core/src):
 *   weights      graphics/weights.rs:176-331 (filter catalogue), :333-350 (negative-lobe ratio),
 *                :352-492 (kernels, bessj1), :681-788 (populate_weights)
 *   colour       graphics/color.rs:22-72 (ColorContext), :85-91 (srgb_to_linear),
 *                :101-108 (uchar_clamp_ff); graphics/lut.rs:4-8 (linear_to_srgb_lut);
 *                LUT generator formula tests/integration/color_conversion.rs:381-388
 *   dispatch     graphics/scaling.rs:19-90 (scale_and_render), :211-251 (alpha forced to 255),
 *                :254-287 (composite_premul_f32_over_srgb_u8), :294-302 (pixel_desc)
 *   flatten      graphics/blend.rs:6-59 (apply_matte)
 *   layout       graphics/bitmaps.rs:712-740 (64-byte stride rule)
 *
 * PARITY STATUS.  Pinned: weights (golden tables tests/integration/weights.txt and
 * weights_params.txt, all rows), the 16384-entry linear->sRGB LUT (vs graphics/lut.rs table),
 * 256-value sRGB round trip, matte KATs.  UNPINNED: the convolution arithmetic itself.  The
 * reference delegates it to the un-vendored crate zenresize 0.3.1 (Cargo.lock:4103-4116) whose
 * source is absent, so pass order / accumulation order / un-premultiply rounding are DEFINED
 * HERE (see "Arithmetic contract" below) and the reference is expected to agree only to the
 * +-1 LSB class its own tests accept (Tolerance::off_by_one, tests/integration/visuals/scaling.rs).
 *
 * Arithmetic contract (the HIP kernels reproduce it bit for bit):
 *   1. sample -> working float:  linear: f = s2l[byte] (f32 table built with powf as color.rs:31-45);
 *      srgb: f = byte * (1/255f).  alpha a = byte * (1/255f).  If the input's alpha is meaningful
 *      colour channels are premultiplied f = f * a (one f32 multiply) and channel 3 carries a;
 *      otherwise channel 3 of the working buffer is the constant 1.0f.
 *   2. vertical pass first ("Scale vertically, then horizontally", tests/integration/variation.rs:240):
 *      v[j][x][c] = chain over taps y = left_j..right_j ascending of acc = fmaf(w, f, acc), acc0 = +0.
 *   3. horizontal pass: o[j][u][c] = chain over taps x = left_u..right_u ascending of fmaf(w, v, acc).
 *   4. output stage per compositing mode, every f32 operation written out separately (no contraction).
 *
 * Build: gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp -shared -fPIC (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define IFO_OK 0
#define IFO_ERR_INVALID_ARGUMENT 1
#define IFO_ERR_NOT_IMPLEMENTED 2
#define IFO_ERR_INVALID_STATE 3
#define IFO_ERR_ALLOC 4

/* ------------------------------------------------------------------------------------------
 * Filter catalogue  (weights.rs:43-78 discriminants, :176-331 parameters)
 * ---------------------------------------------------------------------------------------- */
enum { K_FLEX_CUBIC, K_CUBIC_FAST, K_SINC, K_BOX, K_TRIANGLE, K_SINC_WINDOWED, K_JINC, K_GINSENG };
enum { LOBE_NATURAL = 0, LOBE_EXACT = 1, LOBE_SHARPEN_PERCENT = 2 };

typedef struct {
    double window, p1, p2, p3, q1, q2, q3, q4, blur;
    int kernel;
    int lobe_mode;
    float lobe_value;
} ifo_details;

static const double PI_ = 3.14159265358979323846264338327950288;

static void det_plain(ifo_details* d, double window, double blur, int kernel) {
    /* Default::default() then override window/blur/filter  (weights.rs:126-142) */
    d->window = window; d->blur = blur; d->kernel = kernel;
    d->p1 = 0.0; d->p2 = 1.0; d->p3 = 1.0; d->q1 = 0.0; d->q2 = 1.0; d->q3 = 1.0; d->q4 = 1.0;
    d->lobe_mode = LOBE_NATURAL; d->lobe_value = 0.f;
}
static void det_bicubic(ifo_details* d, double window, double blur, double b, double c) {
    /* weights.rs:159-174 */
    double bx2 = b + b;
    d->window = window; d->blur = blur; d->kernel = K_FLEX_CUBIC;
    d->p1 = 1.0 - (1.0 / 3.0) * b;
    d->p2 = -3.0 + bx2 + c;
    d->p3 = 2.0 - 1.5 * b - c;
    d->q1 = (4.0 / 3.0) * b + 4.0 * c;
    d->q2 = -8.0 * c - bx2;
    d->q3 = b + 5.0 * c;
    d->q4 = (-1.0 / 6.0) * b - c;
    d->lobe_mode = LOBE_NATURAL; d->lobe_value = 0.f;
}

int ifo_details_create(int filter, ifo_details* d) {
    switch (filter) {
    case 22: case 23: det_plain(d, 1.0, 1.0, K_TRIANGLE); break;                  /* Triangle | Linear */
    case 20: det_plain(d, 2.0, 1.0, K_SINC); break;                                /* RawLanczos2 */
    case 18: det_plain(d, 3.0, 1.0, K_SINC); break;                                /* RawLanczos3 */
    case 21: det_plain(d, 2.0, 0.9549963639785485, K_SINC); break;                 /* RawLanczos2Sharp */
    case 19: det_plain(d, 3.0, 0.9812505644269356, K_SINC); break;                 /* RawLanczos3Sharp */
    case 8:  det_plain(d, 2.0, 1.0, K_SINC_WINDOWED); break;                       /* Lanczos2 */
    case 6:  det_plain(d, 3.0, 1.0, K_SINC_WINDOWED); break;                       /* Lanczos */
    case 9:  det_plain(d, 2.0, 0.9549963639785485, K_SINC_WINDOWED); break;        /* Lanczos2Sharp */
    case 7:  det_plain(d, 3.0, 0.9812505644269356, K_SINC_WINDOWED); break;        /* LanczosSharp */
    case 10: det_plain(d, 2.0, 1.0, K_CUBIC_FAST); break;                          /* CubicFast */
    case 24: det_plain(d, 0.5, 1.0, K_BOX); break;                                 /* Box */
    case 4:  det_plain(d, 3.0, 1.0, K_GINSENG); break;                             /* Ginseng */
    case 5:  det_plain(d, 3.0, 0.9812505644269356, K_GINSENG); break;              /* GinsengSharp */
    case 17: det_plain(d, 6.0, 1.0, K_JINC); break;                                /* Jinc */
    case 15: det_bicubic(d, 2.0, 1.0, 1.0, 0.0); break;                            /* CubicBSpline */
    case 11: det_bicubic(d, 2.0, 1.0, 0.0, 1.0); break;                            /* Cubic */
    case 12: det_bicubic(d, 2.0, 0.9549963639785485, 0.0, 1.0); break;             /* CubicSharp */
    case 13: det_bicubic(d, 2.0, 1.0, 0.0, 0.5); break;                            /* CatmullRom */
    case 25: det_bicubic(d, 1.0, 1.0, 0.0, 0.5); break;                            /* CatmullRomFast */
    case 26: det_bicubic(d, 1.0, 13.0 / 16.0, 0.0, 0.5); break;                    /* CatmullRomFastSharp */
    case 14: det_bicubic(d, 2.0, 1.0, 1.0 / 3.0, 1.0 / 3.0); break;                /* Mitchell */
    case 28: det_bicubic(d, 1.0, 1.0, 1.0 / 3.0, 1.0 / 3.0); break;                /* MitchellFast */
    case 29: det_bicubic(d, 2.5, 1.0 / 1.1685777620836933, 0.3782157550939987, 0.3108921224530007); break; /* NCubic */
    case 30: det_bicubic(d, 2.5, 1.0 / 1.105822933719019, 0.2620145123990142, 0.3689927438004929); break;  /* NCubicSharp */
    case 2:  det_bicubic(d, 2.0, 1.0, 0.3782157550939987, 0.3108921224530007); break;                      /* Robidoux */
    case 31: det_bicubic(d, 2.0, 1. / 1.1685777620836932, 0.3782157550939987, 0.3108921224530007); break;  /* LegacyIDCTFilter */
    case 27: det_bicubic(d, 0.74, 0.74, 0.3782157550939987, 0.3108921224530007); break;                    /* Fastest */
    case 1:  det_bicubic(d, 1.05, 1.0, 0.3782157550939987, 0.3108921224530007); break;                     /* RobidouxFast */
    case 3:  det_bicubic(d, 2.0, 1.0, 0.2620145123990142, 0.3689927438004929); break;                      /* RobidouxSharp */
    case 16: det_bicubic(d, 1.0, 1.0, 0.0, 0.0); break;                            /* Hermite */
    default: return IFO_ERR_INVALID_ARGUMENT;
    }
    return IFO_OK;
}

/* weights.rs:460-492 */
static double bessj1(double x) {
    double ax = fabs(x), ans;
    if (ax < 8.0) {
        double y = x * x;
        double ans1 = x * (72362614232.0 + y * (-7895059235.0 + y * (242396853.1
                      + y * (-2972611.439 + y * (15704.48260 + y * (-30.16036606))))));
        double ans2 = 144725228442.0 + y * (2300535178.0 + y * (18583304.74
                      + y * (99447.43394 + y * (376.9991397 + y * 1.0))));
        ans = ans1 / ans2;
    } else {
        double z = 8.0 / ax;
        double y = z * z;
        double xx = ax - 2.356194491;
        double ans1 = 1.0 + y * (0.183105e-2 + y * (-0.3516396496e-4 + y * (0.2457520174e-5 + y * (-0.240337019e-6))));
        double ans2 = 0.04687499995 + y * (-0.2002690873e-3 + y * (0.8449199096e-5
                      + y * (-0.88228987e-6 + y * 0.105787412e-6)));
        ans = sqrt(0.63661977236758134307553505349005744 /* FRAC_2_PI */ / ax) * (cos(xx) * ans1 - z * sin(xx) * ans2);
    }
    return x < 0.0 ? -ans : ans;
}

/* weights.rs:352-458 */
static double kernel_eval(const ifo_details* d, double x) {
    switch (d->kernel) {
    case K_FLEX_CUBIC: {
        double t = fabs(x) / d->blur;
        if (t < 1.0) return d->p1 + t * (t * (d->p2 + t * d->p3));
        if (t < 2.0) return d->q1 + t * (d->q2 + t * (d->q3 + t * d->q4));
        return 0.0;
    }
    case K_CUBIC_FAST: {
        double a = fabs(x) / d->blur, a2 = a * a;
        if (a < 1.0) return 1.0 - 2.0 * a2 + a2 * a;
        if (a < 2.0) return 4.0 - 8.0 * a + 5.0 * a2 - a2 * a;
        return 0.0;
    }
    case K_SINC: {
        double a = fabs(x) / d->blur;
        if (a == 0.0) return 1.0;
        if (a > d->window) return 0.0;
        a = a * PI_;
        return sin(a) / a;
    }
    case K_BOX: {
        double t = x / d->blur;
        return (t >= -d->window && t < d->window) ? 1.0 : 0.0;
    }
    case K_TRIANGLE: {
        double t = fabs(x) / d->blur;
        return t < 1.0 ? 1.0 - t : 0.0;
    }
    case K_SINC_WINDOWED: {
        double t = x / d->blur, a = fabs(t);
        if (a == 0.0) return 1.0;
        if (a > d->window) return 0.0;
        return d->window * sin(PI_ * t / d->window) * sin(t * PI_) / (PI_ * PI_ * t * t);
    }
    case K_JINC: {
        double t = fabs(x) / d->blur;
        if (t == 0.0) return 0.5 * PI_;
        return bessj1(PI_ * t) / t;
    }
    case K_GINSENG: {
        double a = fabs(x) / d->blur, t_pi = a * PI_;
        if (a == 0.0) return 1.0;
        if (a > 3.0) return 0.0;
        double jin = 1.2196698912665046 * t_pi / d->window;
        double jout = bessj1(jin) / (jin * 0.5);
        return jout * sin(t_pi) / t_pi;
    }
    }
    return 0.0;
}

/* weights.rs:333-350 */
double ifo_percent_negative_weight(const ifo_details* d) {
    const int samples = 50;
    double step = d->window / (double)samples;
    double last = kernel_eval(d, -step);
    double pos = 0.0, neg = 0.0;
    for (int i = 0; i < samples + 3; i++) {
        double h = kernel_eval(d, (double)i * step);
        double area = (h + last) / 2.0 * step;
        last = h;
        if (area > 0.0) pos += area; else neg -= area;
    }
    return neg / pos;
}

/* Rust `f64 as i32`: truncate toward zero, saturate, NaN -> 0 */
static int32_t sat_i32(double v) {
    if (v != v) return 0;
    if (v >= 2147483647.0) return 2147483647;
    if (v <= -2147483648.0) return (int32_t)(-2147483647 - 1);
    return (int32_t)v;
}

typedef struct {
    uint32_t n_out;
    uint32_t* left;    /* first source index per output            */
    uint32_t* count;   /* taps per output                          */
    uint32_t* offset;  /* index of first weight in `w` per output  */
    float* w;
    uint32_t n_w;
    uint32_t max_taps;
} ifo_weights;

void ifo_weights_free(ifo_weights* W) {
    if (!W) return;
    free(W->left); free(W->count); free(W->offset); free(W->w);
    memset(W, 0, sizeof *W);
}

/* weights.rs:681-788 */
int ifo_populate_weights(ifo_weights* W, uint32_t out_size, uint32_t in_size, const ifo_details* d) {
    memset(W, 0, sizeof *W);
    if (out_size == 0 || in_size == 0) return IFO_ERR_INVALID_ARGUMENT;
    double sharpen_ratio = ifo_percent_negative_weight(d);
    double desired;
    if (d->lobe_mode == LOBE_EXACT) {
        double r = (double)d->lobe_value;
        desired = r < 0.0 ? 0.0 : (r > 1.0 ? 1.0 : r);
    } else if (d->lobe_mode == LOBE_SHARPEN_PERCENT) {
        double p = (double)d->lobe_value / 100.0;
        double m = sharpen_ratio > p ? sharpen_ratio : p;   /* natural_ratio.max(pct/100) */
        desired = m < 1.0 ? m : 1.0;                        /* 1.0.min(..) */
    } else desired = sharpen_ratio;

    double scale = (double)out_size / (double)in_size;
    double down = scale < 1.0 ? scale : 1.0;
    double half_window = (d->window + 0.5) / down;
    uint32_t alloc_window = (uint32_t)(sat_i32(ceil(2.0 * (half_window - 0.00001))) + 1);

    W->n_out = out_size;
    W->left = (uint32_t*)malloc(sizeof(uint32_t) * out_size);
    W->count = (uint32_t*)malloc(sizeof(uint32_t) * out_size);
    W->offset = (uint32_t*)malloc(sizeof(uint32_t) * out_size);
    W->w = (float*)malloc(sizeof(float) * (size_t)out_size * alloc_window);
    float* tmp = (float*)malloc(sizeof(float) * (size_t)alloc_window);
    if (!W->left || !W->count || !W->offset || !W->w || !tmp) { free(tmp); ifo_weights_free(W); return IFO_ERR_ALLOC; }

    int rc = IFO_OK;
    for (uint32_t u = 0; u < out_size; u++) {
        double center = ((double)u + 0.5) / scale - 0.5;
        int32_t left_edge = sat_i32(ceil(center - d->window / down - 0.0001));
        int32_t right_edge = sat_i32(floor(center + d->window / down + 0.0001));
        uint32_t left = (uint32_t)(left_edge > 0 ? left_edge : 0);
        int32_t rmax = (int32_t)in_size - 1;
        uint32_t right = (uint32_t)(right_edge < rmax ? right_edge : rmax);
        uint32_t count = right - left + 1u;          /* wrapping, as the reference */
        if (count > alloc_window) { rc = IFO_ERR_INVALID_STATE; break; }
        double total = 0.0, tneg = 0.0, tpos = 0.0;
        for (uint32_t ix = left; ix <= right; ix++) {
            double add = kernel_eval(d, down * ((double)ix - center));
            if (fabs(add) <= 2e-8) add = 0.0;
            tmp[ix - left] = (float)add;
            total += add;
            tneg += add < 0.0 ? add : 0.0;
            tpos += add > 0.0 ? add : 0.0;
        }
        float neg_factor = (float)(1.0 / total);
        float pos_factor = neg_factor;
        if (total <= 0.0 || fabs(desired - sharpen_ratio) > 1e-10) {
            if (tneg < 0.0) {
                if (desired < 1.0) {
                    double target_pos = 1.0 / (1.0 - desired);
                    double target_neg = desired * -target_pos;
                    pos_factor = (float)(target_pos / tpos);
                    neg_factor = (float)(target_neg / tneg);
                    if (tneg == 0.0) neg_factor = 1.0f;
                }
            } else if (total == 0.0) { rc = IFO_ERR_INVALID_STATE; break; }
        }
        for (uint32_t i = 0; i < count; i++) {
            if (tmp[i] < 0.f) tmp[i] *= neg_factor; else tmp[i] *= pos_factor;
        }
        /* trim zero ends (weights.rs:771-782) */
        uint32_t lo = 0, hi = count;
        while (hi > lo && tmp[hi - 1] == 0.f) hi--;
        while (lo < hi && tmp[lo] == 0.f) lo++;
        if (hi == lo) { rc = IFO_ERR_INVALID_STATE; break; }   /* NoPixelInputs */
        W->left[u] = left + lo;
        W->count[u] = hi - lo;
        W->offset[u] = W->n_w;
        memcpy(W->w + W->n_w, tmp + lo, sizeof(float) * (hi - lo));
        W->n_w += hi - lo;
        if (hi - lo > W->max_taps) W->max_taps = hi - lo;
    }
    free(tmp);
    if (rc != IFO_OK) ifo_weights_free(W);
    return rc;
}

/* Flat export for tests: returns number of weights; arrays sized by caller (n_out, n_out, n_out*alloc). */
int ifo_weights_flat(int filter, int lobe_mode, float lobe_value, double kernel_width_scale,
                     uint32_t out_size, uint32_t in_size,
                     uint32_t* left, uint32_t* count, float* w, uint32_t w_cap, uint32_t* n_w) {
    ifo_details d;
    int rc = ifo_details_create(filter, &d);
    if (rc) return rc;
    d.blur *= kernel_width_scale;            /* set_kernel_width_scale, weights.rs:155-157 */
    d.lobe_mode = lobe_mode; d.lobe_value = lobe_value;
    ifo_weights W;
    rc = ifo_populate_weights(&W, out_size, in_size, &d);
    if (rc) return rc;
    if (W.n_w > w_cap) { ifo_weights_free(&W); return IFO_ERR_INVALID_ARGUMENT; }
    memcpy(left, W.left, sizeof(uint32_t) * out_size);
    memcpy(count, W.count, sizeof(uint32_t) * out_size);
    memcpy(w, W.w, sizeof(float) * W.n_w);
    *n_w = W.n_w;
    ifo_weights_free(&W);
    return IFO_OK;
}

double ifo_natural_negative_ratio(int filter) {
    ifo_details d;
    if (ifo_details_create(filter, &d)) return -1.0;
    return ifo_percent_negative_weight(&d);
}

/* ------------------------------------------------------------------------------------------
 * Colour  (color.rs, lut.rs)
 * ---------------------------------------------------------------------------------------- */
static float g_s2l[256];      /* sRGB byte -> linear f32, ColorContext(LinearRGB).byte_to_float */
static float g_s2f[256];      /* sRGB byte -> byte/255 f32, ColorContext(StandardRGB)           */
static uint8_t g_l2s[16384];  /* LINEAR_TO_SRGB_LUT                                             */
static int g_tables_ready = 0;

static float srgb_to_linear_f(float s) {               /* color.rs:85-91 */
    if (s <= 0.04045f) return s / 12.92f;
    return powf((s + 0.055f) / (1.0f + 0.055f), 2.4f);
}

void ifo_init_tables(void) {
    if (g_tables_ready) return;
    for (int n = 0; n < 256; n++) {
        float v = (float)n * (1.0f / 255.0f);           /* color.rs:38 */
        g_s2f[n] = v;
        g_s2l[n] = srgb_to_linear_f(v);
    }
    for (int i = 0; i < 16384; i++) {                   /* color_conversion.rs:381-388 */
        double linear = (double)i / 16383.0;
        double srgb = linear <= 0.0031308 ? 12.92 * linear : 1.055 * pow(linear, 1.0 / 2.4) - 0.055;
        double e = srgb * 255.0 + 0.5;
        e = e < 0.0 ? 0.0 : (e > 255.0 ? 255.0 : e);
        g_l2s[i] = (uint8_t)e;
    }
    g_tables_ready = 1;
}

const float* ifo_table_s2l(void) { ifo_init_tables(); return g_s2l; }
const float* ifo_table_s2f(void) { ifo_init_tables(); return g_s2f; }
const uint8_t* ifo_table_l2s(void) { ifo_init_tables(); return g_l2s; }

/* color.rs:101-108.  `(clr as f64 + 0.5) as i16 as u16` with Rust's saturating float casts. */
uint8_t ifo_uchar_clamp_ff(float clr) {
    double t = (double)clr + 0.5;
    int32_t i;
    if (t != t) i = 0;
    else if (t >= 32767.0) i = 32767;
    else if (t <= -32768.0) i = -32768;
    else i = (int32_t)t;
    uint16_t r = (uint16_t)(int16_t)i;
    if (r > 255) r = (clr < 0.0f) ? 0 : 255;
    return (uint8_t)r;
}

/* lut.rs:4-8 : (linear * 16383.0).clamp(0.0, 16383.0) as usize  (NaN -> 0) */
uint8_t ifo_linear_to_srgb_lut(float linear) {
    float s = linear * 16383.0f;
    if (s != s) return g_l2s[0];
    if (s < 0.0f) s = 0.0f;
    if (s > 16383.0f) s = 16383.0f;
    return g_l2s[(uint32_t)s];
}

/* ColorContext::floatspace_to_srgb (color.rs:61-71); space: 0 = StandardRGB, 1 = LinearRGB */
static inline uint8_t float_to_srgb(int linear, float v) {
    return linear ? ifo_linear_to_srgb_lut(v) : ifo_uchar_clamp_ff(255.0f * v);
}

/* ------------------------------------------------------------------------------------------
 * Resample core
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    ifo_weights wv, wh;      /* vertical (in_h -> out_h), horizontal (in_w -> out_w) */
} ifo_plan;

static int make_plan(ifo_plan* P, uint32_t in_w, uint32_t in_h, uint32_t out_w, uint32_t out_h,
                     int filter, float sharpen_percent) {
    ifo_details d;
    int rc = ifo_details_create(filter, &d);
    if (rc) return rc;
    if (sharpen_percent > 0.0f) { d.lobe_mode = LOBE_SHARPEN_PERCENT; d.lobe_value = sharpen_percent; }  /* scaling.rs:103-105 */
    rc = ifo_populate_weights(&P->wv, out_h, in_h, &d);
    if (rc) return rc;
    rc = ifo_populate_weights(&P->wh, out_w, in_w, &d);
    if (rc) { ifo_weights_free(&P->wv); return rc; }
    return IFO_OK;
}

/* convert one BGRA8 row to working floats, 4 per pixel (contract step 1) */
static void row_to_float(const uint8_t* row, uint32_t w, int alpha_meaningful, const float* lut, float* dst) {
    const float a255 = 1.0f / 255.0f;
    if (!alpha_meaningful) {
        for (uint32_t x = 0; x < w; x++) {
            dst[4 * x + 0] = lut[row[4 * x + 0]];
            dst[4 * x + 1] = lut[row[4 * x + 1]];
            dst[4 * x + 2] = lut[row[4 * x + 2]];
            dst[4 * x + 3] = 1.0f;
        }
    } else {
        for (uint32_t x = 0; x < w; x++) {
            float a = (float)row[4 * x + 3] * a255;
            dst[4 * x + 0] = lut[row[4 * x + 0]] * a;
            dst[4 * x + 1] = lut[row[4 * x + 1]] * a;
            dst[4 * x + 2] = lut[row[4 * x + 2]] * a;
            dst[4 * x + 3] = a;
        }
    }
}

/* Per-thread scratch, kept between calls: the batch leg (one frame per OpenMP thread, bench.py's cpu_baseline) used to
 * malloc / free a few megabytes per frame, i.e. mmap / munmap and a thousand page faults per frame under the process's
 * one address-space lock -- with a hundred threads that lock, not the arithmetic, set the rate. */
static __thread struct { void* p; size_t cap; } tls_scratch[3];
static void* scratch(int slot, size_t bytes) {
    if (tls_scratch[slot].cap < bytes) {
        free(tls_scratch[slot].p);
        tls_scratch[slot].p = malloc(bytes);
        tls_scratch[slot].cap = tls_scratch[slot].p ? bytes : 0;
    }
    return tls_scratch[slot].p;
}

/*
 * Resample to the f32 working buffer: out_f32[out_h][out_w][4], premultiplied working-space floats.
 */
static int resample_to_f32(const uint8_t* in, uint32_t in_w, uint32_t in_h, uint32_t in_stride,
                           int alpha_meaningful, int linear, const ifo_plan* P,
                           uint32_t out_w, uint32_t out_h, float* out_f32) {
    const float* lut = linear ? g_s2l : g_s2f;
    uint32_t ring_n = P->wv.max_taps + 1;
    size_t rowf = (size_t)in_w * 4;
    float* ring = (float*)scratch(0, sizeof(float) * rowf * ring_n);
    int64_t* tag = (int64_t*)malloc(sizeof(int64_t) * ring_n);
    float* vrow = (float*)scratch(1, sizeof(float) * rowf);
    if (!ring || !tag || !vrow) { free(tag); return IFO_ERR_ALLOC; }
    for (uint32_t i = 0; i < ring_n; i++) tag[i] = -1;

    for (uint32_t j = 0; j < out_h; j++) {
        uint32_t left = P->wv.left[j], n = P->wv.count[j];
        const float* w = P->wv.w + P->wv.offset[j];
        for (size_t i = 0; i < rowf; i++) vrow[i] = 0.0f;
        for (uint32_t k = 0; k < n; k++) {
            uint32_t y = left + k;
            uint32_t slot = y % ring_n;
            float* f = ring + rowf * slot;
            if (tag[slot] != (int64_t)y) {
                row_to_float(in + (size_t)y * in_stride, in_w, alpha_meaningful, lut, f);
                tag[slot] = (int64_t)y;
            }
            const float wk = w[k];
            for (size_t i = 0; i < rowf; i++) vrow[i] = fmaf(wk, f[i], vrow[i]);
        }
        float* orow = out_f32 + (size_t)j * out_w * 4;
        for (uint32_t u = 0; u < out_w; u++) {
            uint32_t hl = P->wh.left[u], hn = P->wh.count[u];
            const float* hw = P->wh.w + P->wh.offset[u];
            float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
            const float* src = vrow + (size_t)hl * 4;
            for (uint32_t k = 0; k < hn; k++) {
                float wk = hw[k];
                a0 = fmaf(wk, src[4 * k + 0], a0);
                a1 = fmaf(wk, src[4 * k + 1], a1);
                a2 = fmaf(wk, src[4 * k + 2], a2);
                a3 = fmaf(wk, src[4 * k + 3], a3);
            }
            orow[4 * u + 0] = a0; orow[4 * u + 1] = a1; orow[4 * u + 2] = a2;
            orow[4 * u + 3] = alpha_meaningful ? a3 : 1.0f;
        }
    }
    free(tag);
    return IFO_OK;
}

/* scaling.rs:254-287, verbatim semantics */
static void composite_premul_f32_over_srgb_u8(int linear, const float* src, uint8_t* canvas, uint32_t w,
                                              int alpha_meaningful) {
    const float* s2 = linear ? g_s2l : g_s2f;
    float dest_alpha_coeff = alpha_meaningful ? 1.0f / 255.0f : 0.0f;
    float dest_alpha_offset = alpha_meaningful ? 0.0f : 1.0f;
    for (uint32_t x = 0; x < w; x++) {
        const float* sp = src + 4 * x;
        uint8_t* cp = canvas + 4 * x;
        float src_a = sp[3];
        if (src_a > 0.994f || !alpha_meaningful) {
            cp[0] = float_to_srgb(linear, sp[0]);
            cp[1] = float_to_srgb(linear, sp[1]);
            cp[2] = float_to_srgb(linear, sp[2]);
            cp[3] = 255;
        } else {
            uint8_t dest_a = cp[3];
            float dest_coeff = (1.0f - src_a) * (dest_alpha_coeff * (float)(int32_t)dest_a + dest_alpha_offset);
            float final_alpha = src_a + dest_coeff;
            cp[0] = float_to_srgb(linear, (sp[0] + dest_coeff * s2[cp[0]]) / final_alpha);
            cp[1] = float_to_srgb(linear, (sp[1] + dest_coeff * s2[cp[1]]) / final_alpha);
            cp[2] = float_to_srgb(linear, (sp[2] + dest_coeff * s2[cp[2]]) / final_alpha);
            cp[3] = ifo_uchar_clamp_ff(final_alpha * 255.0f);
        }
    }
}

/* ReplaceSelf u8 output (contract step 4a): un-premultiply, encode, alpha=255 if not meaningful */
static void store_replace(int linear, const float* src, uint8_t* canvas, uint32_t w, int alpha_meaningful) {
    for (uint32_t x = 0; x < w; x++) {
        const float* sp = src + 4 * x;
        uint8_t* cp = canvas + 4 * x;
        if (!alpha_meaningful) {
            cp[0] = float_to_srgb(linear, sp[0]);
            cp[1] = float_to_srgb(linear, sp[1]);
            cp[2] = float_to_srgb(linear, sp[2]);
            cp[3] = 255;                                         /* scaling.rs:227-232 */
        } else {
            float a = sp[3];
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (a > 0.0f) { c0 = sp[0] / a; c1 = sp[1] / a; c2 = sp[2] / a; }
            cp[0] = float_to_srgb(linear, c0);
            cp[1] = float_to_srgb(linear, c1);
            cp[2] = float_to_srgb(linear, c2);
            cp[3] = ifo_uchar_clamp_ff(a * 255.0f);
        }
    }
}

/* BlendWithMatte u8 output (contract step 4b): source-over a solid matte, blend.rs:21-52 shape */
static void store_matte(int linear, const float* src, uint8_t* canvas, uint32_t w, int alpha_meaningful,
                        uint32_t matte_bgra) {
    const float* s2 = linear ? g_s2l : g_s2f;
    uint8_t mb = (uint8_t)(matte_bgra & 255), mg = (uint8_t)((matte_bgra >> 8) & 255),
            mr = (uint8_t)((matte_bgra >> 16) & 255), ma = (uint8_t)(matte_bgra >> 24);
    float matte_a = (float)ma * (1.0f / 255.0f);
    float m0 = s2[mb], m1 = s2[mg], m2 = s2[mr];
    for (uint32_t x = 0; x < w; x++) {
        const float* sp = src + 4 * x;
        uint8_t* cp = canvas + 4 * x;
        if (!alpha_meaningful) {
            cp[0] = float_to_srgb(linear, sp[0]);
            cp[1] = float_to_srgb(linear, sp[1]);
            cp[2] = float_to_srgb(linear, sp[2]);
            cp[3] = 255;
        } else {
            float a = sp[3];
            a = a < 0.0f ? 0.0f : (a > 1.0f ? 1.0f : a);
            float ia = (1.0f - a) * matte_a;
            float final_a = ia + a;
            float c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (final_a > 0.0f) {
                c0 = (sp[0] + m0 * ia) / final_a;
                c1 = (sp[1] + m1 * ia) / final_a;
                c2 = (sp[2] + m2 * ia) / final_a;
            }
            cp[0] = float_to_srgb(linear, c0);
            cp[1] = float_to_srgb(linear, c1);
            cp[2] = float_to_srgb(linear, c2);
            cp[3] = ifo_uchar_clamp_ff(255.0f * final_a);
        }
    }
}

/*
 * scale_and_render (scaling.rs:19-90).
 * compositing: 0 ReplaceSelf, 1 BlendWithSelf, 2 BlendWithMatte  (ffi/mod.rs:41-47)
 * working_space: 0 StandardRGB, 1 LinearRGB
 * f32_dump: optional [h][w][4] working buffer (premultiplied, working space) for the ULP check.
 */
int ifo_scale_and_render(const uint8_t* in, uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful,
                         uint8_t* canvas, uint32_t cw, uint32_t ch, uint32_t c_stride,
                         uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                         int filter, float sharpen_percent_goal, int working_space, int compositing,
                         uint32_t matte_bgra, float* f32_dump) {
    ifo_init_tables();
    if ((uint64_t)h + y > ch || (uint64_t)w + x > cw) return IFO_ERR_INVALID_ARGUMENT;   /* scaling.rs:24-29 */
    if (w == 0 || h == 0 || in_w == 0 || in_h == 0) return IFO_ERR_INVALID_ARGUMENT;
    if (compositing < 0 || compositing > 2) return IFO_ERR_INVALID_ARGUMENT;
    int linear = working_space == 1;
    ifo_plan P;
    int rc = make_plan(&P, in_w, in_h, w, h, filter, sharpen_percent_goal);
    if (rc) return rc;
    float* buf = f32_dump ? f32_dump : (float*)scratch(2, sizeof(float) * (size_t)w * h * 4);
    if (!buf) { ifo_weights_free(&P.wv); ifo_weights_free(&P.wh); return IFO_ERR_ALLOC; }
    rc = resample_to_f32(in, in_w, in_h, in_stride, in_alpha_meaningful, linear, &P, w, h, buf);
    if (rc == IFO_OK) {
        for (uint32_t j = 0; j < h; j++) {
            uint8_t* crow = canvas + (size_t)(y + j) * c_stride + (size_t)x * 4;
            const float* srow = buf + (size_t)j * w * 4;
            if (compositing == 0) store_replace(linear, srow, crow, w, in_alpha_meaningful);
            else if (compositing == 2) store_matte(linear, srow, crow, w, in_alpha_meaningful, matte_bgra);
            else composite_premul_f32_over_srgb_u8(linear, srow, crow, w, in_alpha_meaningful);
        }
    }
    ifo_weights_free(&P.wv); ifo_weights_free(&P.wh);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Contract variants.  Used ONLY by tests/test_oracle_reference_checksums.py to measure how far
 * the choices of the arithmetic contract (header, steps 1-4) can move the BGRA8 result, and
 * which of them still reproduce the reference's stored checksums
 * (tests/integration/visuals/canvas.checksums, trim.checksums).  flags == 0 is the contract.
 * ---------------------------------------------------------------------------------------- */
enum {
    IFO_VAR_HFIRST       = 1u << 0,   /* horizontal pass first                                   */
    IFO_VAR_ACC_MASK     = 3u << 1,   /* 0 ascending fmaf | 1 ascending mul,add | 2 f64 sum, one rounding | 3 descending fmaf */
    IFO_VAR_UNPREMUL_RCP = 1u << 3,   /* c * (1/a) instead of c / a                              */
    IFO_VAR_ENCODE_EXACT = 1u << 4,   /* linear->sRGB by the f64 formula, round half up, no LUT  */
    IFO_VAR_S2L_F64      = 1u << 5,   /* sRGB->linear table from f64 pow, rounded to f32         */
    IFO_VAR_ALPHA_RNE    = 1u << 6,   /* alpha byte by round-half-even instead of uchar_clamp_ff */
    IFO_VAR_NO_PREMUL    = 1u << 7,   /* negative control: straight alpha, four plain channels   */
    IFO_VAR_SRGB_SPACE   = 1u << 8    /* negative control: filter in sRGB space                  */
};

static inline float var_acc(uint32_t mode, const float* w, const float* v, size_t vstep, uint32_t n) {
    if (mode == 0) { float a = 0.f; for (uint32_t k = 0; k < n; k++) a = fmaf(w[k], v[k * vstep], a); return a; }
    if (mode == 1) { float a = 0.f; for (uint32_t k = 0; k < n; k++) { float p = w[k] * v[k * vstep]; a = a + p; } return a; }
    if (mode == 2) { double a = 0.0; for (uint32_t k = 0; k < n; k++) a += (double)w[k] * (double)v[k * vstep]; return (float)a; }
    float a = 0.f; for (uint32_t k = n; k-- > 0;) a = fmaf(w[k], v[k * vstep], a); return a;
}

static uint8_t var_encode(uint32_t flags, int linear, float v) {
    if (!linear || !(flags & IFO_VAR_ENCODE_EXACT)) return float_to_srgb(linear, v);
    double l = (double)v;
    if (!(l == l) || l <= 0.0) return 0;
    if (l >= 1.0) return 255;
    double s = l <= 0.0031308 ? 12.92 * l : 1.055 * pow(l, 1.0 / 2.4) - 0.055;
    double e = s * 255.0 + 0.5;
    return (uint8_t)(e < 0.0 ? 0.0 : (e > 255.0 ? 255.0 : e));
}

int ifo_scale_and_render_variant(const uint8_t* in, uint32_t in_w, uint32_t in_h, uint32_t in_stride,
                                 int alpha_meaningful, uint8_t* canvas, uint32_t c_stride, uint32_t w, uint32_t h,
                                 int filter, float sharpen, int working_space, uint32_t flags) {
    ifo_init_tables();
    int linear = working_space == 1 && !(flags & IFO_VAR_SRGB_SPACE);
    uint32_t mode = (flags & IFO_VAR_ACC_MASK) >> 1;
    ifo_plan P;
    int rc = make_plan(&P, in_w, in_h, w, h, filter, sharpen);
    if (rc) return rc;
    float lut[256];
    for (int i = 0; i < 256; i++) {
        lut[i] = linear ? g_s2l[i] : g_s2f[i];
        if (linear && (flags & IFO_VAR_S2L_F64)) {
            double s = (double)i / 255.0;
            lut[i] = (float)(s <= 0.04045 ? s / 12.92 : pow((s + 0.055) / 1.055, 2.4));
        }
    }
    size_t npx = (size_t)in_w * in_h;
    float* src = (float*)malloc(sizeof(float) * 4 * npx);
    size_t mid_w = (flags & IFO_VAR_HFIRST) ? w : in_w, mid_h = (flags & IFO_VAR_HFIRST) ? in_h : h;
    float* mid = (float*)malloc(sizeof(float) * 4 * mid_w * mid_h);
    float* out = (float*)malloc(sizeof(float) * 4 * (size_t)w * h);
    if (!src || !mid || !out) { free(src); free(mid); free(out); ifo_weights_free(&P.wv); ifo_weights_free(&P.wh); return IFO_ERR_ALLOC; }
    for (uint32_t y = 0; y < in_h; y++) {
        const uint8_t* row = in + (size_t)y * in_stride;
        float* d = src + (size_t)y * in_w * 4;
        if (alpha_meaningful && (flags & IFO_VAR_NO_PREMUL)) {
            for (uint32_t x = 0; x < in_w; x++) {
                d[4 * x] = lut[row[4 * x]]; d[4 * x + 1] = lut[row[4 * x + 1]]; d[4 * x + 2] = lut[row[4 * x + 2]];
                d[4 * x + 3] = (float)row[4 * x + 3] * (1.0f / 255.0f);
            }
        } else row_to_float(row, in_w, alpha_meaningful, lut, d);
    }
    if (flags & IFO_VAR_HFIRST) {
        for (uint32_t y = 0; y < in_h; y++)
            for (uint32_t u = 0; u < w; u++)
                for (int c = 0; c < 4; c++)
                    mid[((size_t)y * w + u) * 4 + c] = var_acc(mode, P.wh.w + P.wh.offset[u],
                        src + ((size_t)y * in_w + P.wh.left[u]) * 4 + c, 4, P.wh.count[u]);
        for (uint32_t j = 0; j < h; j++)
            for (uint32_t u = 0; u < w; u++)
                for (int c = 0; c < 4; c++)
                    out[((size_t)j * w + u) * 4 + c] = var_acc(mode, P.wv.w + P.wv.offset[j],
                        mid + ((size_t)P.wv.left[j] * w + u) * 4 + c, (size_t)w * 4, P.wv.count[j]);
    } else {
        for (uint32_t j = 0; j < h; j++)
            for (uint32_t x = 0; x < in_w; x++)
                for (int c = 0; c < 4; c++)
                    mid[((size_t)j * in_w + x) * 4 + c] = var_acc(mode, P.wv.w + P.wv.offset[j],
                        src + ((size_t)P.wv.left[j] * in_w + x) * 4 + c, (size_t)in_w * 4, P.wv.count[j]);
        for (uint32_t j = 0; j < h; j++)
            for (uint32_t u = 0; u < w; u++)
                for (int c = 0; c < 4; c++)
                    out[((size_t)j * w + u) * 4 + c] = var_acc(mode, P.wh.w + P.wh.offset[u],
                        mid + ((size_t)j * in_w + P.wh.left[u]) * 4 + c, 4, P.wh.count[u]);
    }
    for (uint32_t j = 0; j < h; j++) {
        uint8_t* cp = canvas + (size_t)j * c_stride;
        const float* sp = out + (size_t)j * w * 4;
        for (uint32_t u = 0; u < w; u++, cp += 4, sp += 4) {
            if (!alpha_meaningful) {
                cp[0] = var_encode(flags, linear, sp[0]); cp[1] = var_encode(flags, linear, sp[1]);
                cp[2] = var_encode(flags, linear, sp[2]); cp[3] = 255;
                continue;
            }
            float a = sp[3], c0 = 0.f, c1 = 0.f, c2 = 0.f;
            if (flags & IFO_VAR_NO_PREMUL) { c0 = sp[0]; c1 = sp[1]; c2 = sp[2]; }
            else if (a > 0.0f) {
                if (flags & IFO_VAR_UNPREMUL_RCP) { float r = 1.0f / a; c0 = sp[0] * r; c1 = sp[1] * r; c2 = sp[2] * r; }
                else { c0 = sp[0] / a; c1 = sp[1] / a; c2 = sp[2] / a; }
            }
            cp[0] = var_encode(flags, linear, c0); cp[1] = var_encode(flags, linear, c1); cp[2] = var_encode(flags, linear, c2);
            if (flags & IFO_VAR_ALPHA_RNE) {
                float t = a * 255.0f;
                t = t < 0.f ? 0.f : (t > 255.f ? 255.f : t);
                cp[3] = (uint8_t)lrintf(t);
            } else cp[3] = ifo_uchar_clamp_ff(a * 255.0f);
        }
    }
    free(src); free(mid); free(out);
    ifo_weights_free(&P.wv); ifo_weights_free(&P.wh);
    return IFO_OK;
}

/* Batch over independent images, one image per OpenMP thread (the CPU baseline leg). */
int ifo_scale_and_render_batch(const uint8_t* in, size_t in_image_bytes, uint32_t n_images,
                               uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful,
                               uint8_t* canvas, size_t canvas_image_bytes, uint32_t cw, uint32_t ch, uint32_t c_stride,
                               uint32_t x, uint32_t y, uint32_t w, uint32_t h,
                               int filter, float sharpen, int working_space, int compositing, uint32_t matte_bgra,
                               int n_threads) {
    int rc_all = 0;
    ifo_init_tables();
    if (n_threads < 1) n_threads = 1;
#pragma omp parallel for num_threads(n_threads) schedule(dynamic, 1)
    for (int64_t i = 0; i < (int64_t)n_images; i++) {
        int rc = ifo_scale_and_render(in + (size_t)i * in_image_bytes, in_w, in_h, in_stride, in_alpha_meaningful,
                                      canvas + (size_t)i * canvas_image_bytes, cw, ch, c_stride, x, y, w, h,
                                      filter, sharpen, working_space, compositing, matte_bgra, NULL);
        if (rc) {
#pragma omp critical
            rc_all = rc;
        }
    }
    return rc_all;
}

/* blend.rs:6-59 */
int ifo_apply_matte(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int alpha_meaningful, uint32_t matte_bgra) {
    ifo_init_tables();
    if (!alpha_meaningful) return IFO_OK;                                  /* blend.rs:11-13 */
    uint8_t mb = (uint8_t)(matte_bgra & 255), mg = (uint8_t)((matte_bgra >> 8) & 255),
            mr = (uint8_t)((matte_bgra >> 16) & 255), ma = (uint8_t)(matte_bgra >> 24);
    const float alpha_to_float = 1.0f / 255.0f;
    float matte_a0 = (float)ma * alpha_to_float;
    float matte_b = g_s2l[mb], matte_g = g_s2l[mg], matte_r = g_s2l[mr];
    for (uint32_t yy = 0; yy < h; yy++) {
        uint8_t* row = bgra + (size_t)yy * stride;
        for (uint32_t xx = 0; xx < w; xx++) {
            uint8_t* p = row + 4 * xx;
            uint8_t pa = p[3];
            float paf = (float)(int32_t)pa * alpha_to_float;
            if (pa == 0) { p[0] = mb; p[1] = mg; p[2] = mr; p[3] = ma; }
            else if (pa != 255) {
                float matte_a = (1.0f - paf) * matte_a0;
                float final_a = matte_a + paf;
                uint8_t nb = ifo_linear_to_srgb_lut((g_s2l[p[0]] * paf + matte_b * matte_a) / final_a);
                uint8_t ng = ifo_linear_to_srgb_lut((g_s2l[p[1]] * paf + matte_g * matte_a) / final_a);
                uint8_t nr = ifo_linear_to_srgb_lut((g_s2l[p[2]] * paf + matte_r * matte_a) / final_a);
                p[0] = nb; p[1] = ng; p[2] = nr;
                p[3] = ifo_uchar_clamp_ff(255.0f * final_a);
            }
        }
    }
    return IFO_OK;
}

/* bitmaps.rs:712-740 with alignment 64 */
uint32_t ifo_stride_for_width(uint32_t w) {
    uint64_t un = (uint64_t)w * 4;
    uint64_t pad = (un % 64) ? 64 - (un % 64) : 0;
    return (uint32_t)(un + pad);
}
