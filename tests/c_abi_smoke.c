/* C99 consumer of include/imageflow_hip.h: proves the header is plain C and the library links with C linkage.
 * Only host-side entry points are called (no GPU): the filter-weight tables and the JPEG header parser. */
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
    printf("c abi ok: %s, %u weights\n", ifhip_version(), n);
    return 0;
}
