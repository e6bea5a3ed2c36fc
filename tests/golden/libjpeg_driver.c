/*
 * libjpeg_driver.c -- drives the SYSTEM libjpeg (/lib/x86_64-linux-gnu/libjpeg.so.8 = libjpeg-turbo 2.1.2, v8 ABI)
 * without its headers (none are installed): the public jpeg_decompress_struct prefix is addressed by offset.
 * Used ONLY by tests/golden/make_jpeg_scaled_golden.py to record what libjpeg decodes at scale_num/8 for
 * scale_num = 1..8 (jdmaster.c / jidctint.c / jdsample.c of the library mozjpeg-sys derives from).  Test infrastructure.
 *
 * Layout facts used (jpeglib.h of libjpeg-turbo 2.1, LP64): err @0, src @40, image_width @48, image_height @52,
 * num_components @56, jpeg_color_space @60, out_color_space @64, scale_num @68, scale_denom @72, dct_method @96,
 * do_fancy_upsampling @100, do_block_smoothing @104, output_width @136, output_height @140, output_components @148,
 * output_scanline @168.  verify() checks them against a decoded header before anything is trusted, and
 * the struct size is found by asking jpeg_CreateDecompress (it refuses a wrong size).
 */
#include <setjmp.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

extern void* jpeg_std_error(void* err);
extern void jpeg_CreateDecompress(void* cinfo, int version, size_t structsize);
extern void jpeg_destroy_decompress(void* cinfo);
extern void jpeg_mem_src(void* cinfo, const unsigned char* buf, unsigned long size);
extern int jpeg_read_header(void* cinfo, int require_image);
extern int jpeg_start_decompress(void* cinfo);
extern unsigned int jpeg_read_scanlines(void* cinfo, unsigned char** rows, unsigned int max_lines);
extern int jpeg_finish_decompress(void* cinfo);

static jmp_buf g_jmp;
static void on_error(void* cinfo) { (void)cinfo; longjmp(g_jmp, 1); }
static void on_message(void* cinfo, int lvl) { (void)cinfo; (void)lvl; }

#define U32(c, off) (*(uint32_t*)((unsigned char*)(c) + (off)))
#define I32(c, off) (*(int32_t*)((unsigned char*)(c) + (off)))

static size_t g_size = 0;
static size_t struct_size(void) {
    if (g_size) return g_size;
    for (size_t s = 400; s <= 1024; s += 8) {
        unsigned char err[1024], cinfo[2048];
        memset(cinfo, 0, sizeof cinfo);
        *(void**)cinfo = jpeg_std_error(err);
        *(void (**)(void*))err = on_error;
        if (setjmp(g_jmp)) continue;
        jpeg_CreateDecompress(cinfo, 80, s);
        jpeg_destroy_decompress(cinfo);
        g_size = s;
        return s;
    }
    return 0;
}

/* Decode `jpg` at scale_num/8 into RGB rows (3 bytes per pixel).  Returns 0 on success; *ow,*oh = output size.
 * out may be NULL to query the size.  fancy = do_fancy_upsampling. */
int ljd_decode_scaled(const unsigned char* jpg, unsigned long len, int scale_num, int fancy, unsigned char* out,
                      size_t out_cap, uint32_t* ow, uint32_t* oh) {
    size_t sz = struct_size();
    if (!sz) return -1;
    unsigned char err[1024];
    unsigned char* cinfo = calloc(1, 4096);
    *(void**)cinfo = jpeg_std_error(err);
    ((void (**)(void*))err)[0] = on_error;                       /* error_exit   */
    ((void (**)(void*, int))err)[1] = on_message;                /* emit_message */
    if (setjmp(g_jmp)) { jpeg_destroy_decompress(cinfo); free(cinfo); return -2; }
    jpeg_CreateDecompress(cinfo, 80, sz);
    jpeg_mem_src(cinfo, jpg, len);
    jpeg_read_header(cinfo, 1);
    /* layout check: defaults after jpeg_read_header (jdapimin.c default_decompress_parms) */
    if (I32(cinfo, 96) != 0 /* JDCT_ISLOW */ || I32(cinfo, 100) != 1 || I32(cinfo, 104) != 1 ||
        U32(cinfo, 68) != 1 || U32(cinfo, 72) != 1 || U32(cinfo, 48) == 0 || U32(cinfo, 52) == 0) {
        jpeg_destroy_decompress(cinfo); free(cinfo); return -3;
    }
    int ncomp = I32(cinfo, 56);
    U32(cinfo, 68) = (uint32_t)scale_num;
    U32(cinfo, 72) = 8;
    I32(cinfo, 100) = fancy;
    I32(cinfo, 64) = ncomp == 1 ? 1 /* JCS_GRAYSCALE */ : 2 /* JCS_RGB */;
    jpeg_start_decompress(cinfo);
    uint32_t w = U32(cinfo, 136), h = U32(cinfo, 140);
    int oc = I32(cinfo, 148);
    *ow = w; *oh = h;
    if (out) {
        if ((size_t)w * h * 3 > out_cap) { jpeg_destroy_decompress(cinfo); free(cinfo); return -4; }
        unsigned char* row = malloc((size_t)w * 4);
        while (U32(cinfo, 168) < h) {
            uint32_t y = U32(cinfo, 168);
            unsigned char* rows[1] = {row};
            if (jpeg_read_scanlines(cinfo, rows, 1) != 1) break;
            for (uint32_t x = 0; x < w; x++)
                for (int c = 0; c < 3; c++) out[((size_t)y * w + x) * 3 + c] = oc == 1 ? row[x] : row[x * 3 + c];
        }
        free(row);
        jpeg_finish_decompress(cinfo);
    }
    jpeg_destroy_decompress(cinfo);
    free(cinfo);
    return 0;
}
int ljd_struct_size(void) { return (int)struct_size(); }

/* CLI: libjpeg_driver in.jpg scale_num fancy out.bin  ->  out.bin = u32le w, u32le h, then RGB rows.
 * (A separate process on purpose: Pillow bundles its own libjpeg with the same symbol names.) */
#ifdef LJD_MAIN
#include <stdio.h>
int main(int argc, char** argv) {
    if (argc != 5) { fprintf(stderr, "usage: %s in.jpg scale_num fancy out.bin\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) return 3;
    fseek(f, 0, SEEK_END);
    long n = ftell(f);
    fseek(f, 0, SEEK_SET);
    unsigned char* jpg = malloc((size_t)n);
    if (fread(jpg, 1, (size_t)n, f) != (size_t)n) return 3;
    fclose(f);
    uint32_t w = 0, h = 0;
    int rc = ljd_decode_scaled(jpg, (unsigned long)n, atoi(argv[2]), atoi(argv[3]), NULL, 0, &w, &h);
    if (rc) return 10 - rc;
    unsigned char* out = malloc((size_t)w * h * 3 + 1);
    rc = ljd_decode_scaled(jpg, (unsigned long)n, atoi(argv[2]), atoi(argv[3]), out, (size_t)w * h * 3, &w, &h);
    if (rc) return 10 - rc;
    FILE* o = fopen(argv[4], "wb");
    fwrite(&w, 4, 1, o); fwrite(&h, 4, 1, o); fwrite(out, 1, (size_t)w * h * 3, o);
    fclose(o);
    return 0;
}
#endif
