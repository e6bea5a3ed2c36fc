/* bitmap_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked, imported or called by the product).
 *
 * CPU restatement of the whole-bitmap byte operations that sit next to the resampler in imageflow's graphs
 * (SURVEY.md section 8f rows 2 and 4).  Every function follows in-tree reference source, so these are pinned by
 * construction (pure byte moves / one documented f32 expression):
 *   colour matrix   imageflow_core/src/graphics/color_matrix.rs:5-29 (window_bgra32_apply_color_matrix)
 *                   filter matrices: imageflow_core/src/flow/nodes/color.rs:86-230 (host side, mirrored in Python)
 *   copy_rect       imageflow_core/src/graphics/copy_rect.rs:12-119 (incl. the unused-alpha normalisation :47-66)
 *   fill_rect       imageflow_core/src/graphics/bitmaps.rs:1504-1548 (fill_rectangle)
 *   flips           imageflow_core/src/graphics/flip.rs:10-38
 *   transpose       imageflow_core/src/graphics/transpose.rs:95-121 (bitmap_window_transpose)
 * Rust evaluates `a*r + b*g + c*b + d*a + e` left to right with one rounding per operation and never contracts to
 * FMA; this file is compiled with -ffp-contract=off to match.
 */
#include <stdint.h>
#include <string.h>

uint8_t ifo_uchar_clamp_ff(float clr);

#define BO_OK 0
#define BO_INVALID_ARGUMENT 1

/* color_matrix.rs:5-29.  m is row-major [5][5]; pixels are B,G,R,A bytes. */
int bo_apply_color_matrix(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, const float* m) {
    const float m40 = m[20] * 255.0f, m41 = m[21] * 255.0f, m42 = m[22] * 255.0f, m43 = m[23] * 255.0f;
    for (uint32_t y = 0; y < h; y++) {
        uint8_t* p = bgra + (size_t)y * stride;
        for (uint32_t x = 0; x < w; x++, p += 4) {
            const float b = (float)p[0], g = (float)p[1], r = (float)p[2], a = (float)p[3];
            const uint8_t nr = ifo_uchar_clamp_ff(m[0] * r + m[5] * g + m[10] * b + m[15] * a + m40);
            const uint8_t ng = ifo_uchar_clamp_ff(m[1] * r + m[6] * g + m[11] * b + m[16] * a + m41);
            const uint8_t nb = ifo_uchar_clamp_ff(m[2] * r + m[7] * g + m[12] * b + m[17] * a + m42);
            const uint8_t na = ifo_uchar_clamp_ff(m[3] * r + m[8] * g + m[13] * b + m[18] * a + m43);
            p[0] = nb; p[1] = ng; p[2] = nr; p[3] = na;
        }
    }
    return BO_OK;
}

static void set_alpha_255(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride) {
    for (uint32_t y = 0; y < h; y++)
        for (uint32_t x = 0; x < w; x++) bgra[(size_t)y * stride + 4u * x + 3u] = 255;
}

/* copy_rect.rs:12-119 for 32-bit bitmaps.  *canvas_alpha_meaningful is updated as the reference updates the canvas. */
int bo_copy_rect(uint8_t* input, uint32_t in_w, uint32_t in_h, uint32_t in_stride, int in_alpha_meaningful,
                 uint8_t* canvas, uint32_t cw, uint32_t ch, uint32_t c_stride, int* canvas_alpha_meaningful,
                 uint32_t from_x, uint32_t from_y, uint32_t to_x, uint32_t to_y, uint32_t w, uint32_t h) {
    if (in_w <= from_x || in_h <= from_y || (uint64_t)in_w < (uint64_t)from_x + w || (uint64_t)in_h < (uint64_t)from_y + h ||
        (uint64_t)cw < (uint64_t)to_x + w || (uint64_t)ch < (uint64_t)to_y + h)
        return BO_INVALID_ARGUMENT;
    if (!*canvas_alpha_meaningful && in_alpha_meaningful) {          /* :47-54 Bgr32 canvas, Bgra32 input */
        set_alpha_255(canvas, cw, ch, c_stride);
        *canvas_alpha_meaningful = 1;
    }
    if (!in_alpha_meaningful && *canvas_alpha_meaningful)            /* :64-66 Bgr32 input, Bgra32 canvas */
        set_alpha_255(input, in_w, in_h, in_stride);
    for (uint32_t y = 0; y < h; y++)
        memmove(canvas + (size_t)(to_y + y) * c_stride + 4u * (size_t)to_x,
                input + (size_t)(from_y + y) * in_stride + 4u * (size_t)from_x, 4u * (size_t)w);
    return BO_OK;
}

/* bitmaps.rs:1504-1548; color = Color32 0xAARRGGBB whose little-endian bytes are B,G,R,A. */
int bo_fill_rect(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride, int blend_with_matte,
                 uint32_t x1, uint32_t y1, uint32_t x2, uint32_t y2, uint32_t color) {
    if (blend_with_matte && !(x1 == 0 && y1 == 0 && x2 == w && y2 == h)) return BO_INVALID_ARGUMENT;
    if (y2 == y1 || x2 == x1) return BO_OK;
    if (y2 <= y1 || x2 <= x1 || x2 > w || y2 > h) return BO_INVALID_ARGUMENT;
    for (uint32_t y = y1; y < y2; y++)
        for (uint32_t x = x1; x < x2; x++) memcpy(bgra + (size_t)y * stride + 4u * (size_t)x, &color, 4);
    return BO_OK;
}

/* flip.rs:10-22: only the first 4*w bytes of each row move (row padding stays) */
int bo_flip_vertical(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride) {
    for (uint32_t y = 0; y < h / 2; y++) {
        uint8_t* a = bgra + (size_t)y * stride;
        uint8_t* b = bgra + (size_t)(h - 1 - y) * stride;
        for (size_t i = 0; i < 4u * (size_t)w; i++) { uint8_t t = a[i]; a[i] = b[i]; b[i] = t; }
    }
    return BO_OK;
}

/* flip.rs:26-38 */
int bo_flip_horizontal(uint8_t* bgra, uint32_t w, uint32_t h, uint32_t stride) {
    for (uint32_t y = 0; y < h; y++) {
        uint32_t* row = (uint32_t*)(bgra + (size_t)y * stride);
        for (uint32_t x = 0; x < w / 2; x++) { uint32_t t = row[x]; row[x] = row[w - 1 - x]; row[w - 1 - x] = t; }
    }
    return BO_OK;
}

/* transpose.rs:95-121: to(x = y_from, y = x_from) = from(x_from, y_from); needs from.w == to.h and from.h == to.w */
int bo_transpose(const uint8_t* from, uint32_t from_w, uint32_t from_h, uint32_t from_stride,
                 uint8_t* to, uint32_t to_w, uint32_t to_h, uint32_t to_stride) {
    if (from_w != to_h || from_h != to_w) return BO_INVALID_ARGUMENT;
    for (uint32_t y = 0; y < from_h; y++)
        for (uint32_t x = 0; x < from_w; x++)
            memcpy(to + (size_t)x * to_stride + 4u * (size_t)y, from + (size_t)y * from_stride + 4u * (size_t)x, 4);
    return BO_OK;
}
