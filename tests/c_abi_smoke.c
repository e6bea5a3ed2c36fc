/* C99 consumer of include/imageflow_hip.h: proves the header is plain C and the library links with C linkage.
 * Only host-side entry points are called (no GPU): the filter-weight tables, the JPEG header parser, the file writer and
 * the decode-table report on the writer's own files. */
#include <stdio.h>
#include <string.h>

#include "imageflow_hip.h"

int main(void) {
    uint32_t left[200], count[200], n = 0;
    static float w[200 * 100];
    int rc = ifhip_populate_weights(IFHIP_FILTER_ROBIDOUX, IFHIP_LOBE_NATURAL, 0.0f, 1.0, 200, 3840, left, count, w, 200 * 100, &n);
    if (rc != IFHIP_OK) { printf("populate_weights failed: %s\n", ifhip_last_error_message()); return 1; }
    double sum = 0.0;
    for (uint32_t k = 0; k < count[0]; ++k) sum += w[k];                  /* the first output's weights sum to 1 */
    if (n < 200 * 40 || sum < 0.999 || sum > 1.001) { printf("unexpected weights: n=%u sum=%f\n", n, sum); return 2; }
    uint32_t width = 0, height = 0;
    const uint8_t not_jpeg[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    rc = ifhip_jpeg_parse_headers(not_jpeg, sizeof not_jpeg, &width, &height, 0, 0, 0, 0, 0, 0, 0);
    if (rc == IFHIP_OK || strstr(ifhip_last_error_message(), "not a JPEG") == 0) { printf("parser accepted garbage\n"); return 3; }
    if (ifhip_stride_for_width(200) != 832u) return 4;
    /* host writer -> host parser + table report: a 16x8 gray file of two blocks, baseline and progressive */
    {
        static int16_t coef[2 * 64];
        const uint32_t bw[3] = {2, 0, 0}, bh[3] = {1, 0, 0};
        static uint8_t file[4096];
        size_t len = 0;
        int flags;
        coef[0] = 37; coef[1] = -3; coef[8] = 2; coef[64] = 30; coef[64 + 63] = 1;
        for (flags = 0; flags <= (IFHIP_JPEG_OPTIMIZE_HUFFMAN | IFHIP_JPEG_PROGRESSIVE); ++flags) {
            ifhip_jpeg_scan_report rep;
            rc = ifhip_jpeg_write(coef, 0, 0, bw, bh, 1, 0, 0, 16, 8, 90, flags, file, sizeof file, &len);
            if (rc != IFHIP_OK || len < 100 || file[0] != 0xFF || file[1] != 0xD8 || file[len - 1] != 0xD9) { printf("writer failed (flags %d): %s\n", flags, ifhip_last_error_message()); return 5; }
            rc = ifhip_jpeg_parse_headers(file, len, &width, &height, 0, 0, 0, 0, 0, 0, 0);
            if (flags & IFHIP_JPEG_PROGRESSIVE) {
                if (rc == IFHIP_OK) { printf("progressive file taken for baseline\n"); return 6; }
                continue;
            }
            if (rc != IFHIP_OK || width != 16 || height != 8) { printf("parser refused the writer's file: %s\n", ifhip_last_error_message()); return 7; }
            rc = ifhip_jpeg_debug_scan_report(file, len, &rep);
            if (rc != IFHIP_OK || rep.blocks != 2 || rep.dc_last_segment[0] != 30 || rep.segments_with_wrong_block_count || rep.pair_walk_mismatches ||
                rep.count_walk_mismatches || !rep.scan_complete) { printf("scan report disagrees with what was written\n"); return 8; }
        }
    }
    printf("c abi ok: %s, %u weights\n", ifhip_version(), n);
    return 0;
}
