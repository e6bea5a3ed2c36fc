// block_scalers.cpp -- imageflow's 8x8 -> NxN spatial block down-scalers for the JPEG luma path, host tables.
//
// The reference ships these as generated C (c_components/lib/codecs_jpeg_idct_fast.c, 14 functions
// flow_scale_spatial[_srgb]_{1..7}x{1..7}) selected for luma blocks when libjpeg decodes at scale_num/8
// (codec_jpeg_wrapper.c:274-343): int8 weights with a power-of-two divisor per output, applied vertically then
// horizontally in int32 with round-half-up, optionally through 12-bit sRGB<->linear tables.  The generator in the
// tree today (imageflow_core/tests/integration/variation.rs) no longer reproduces the committed numbers (older
// weights; older powf), so the numbers themselves are the specification: 28 weight rows and the two LUTs are carried
// as data (block_scaler_weights.inc, block_scaler_luts.inc, extracted by tests/golden/make_block_scaler_tables.py)
// and tests/test_gpu_block_scalers.py runs the kernel against the reference's own compiled functions.
#include <cstring>
#include <mutex>

#include "common.hpp"
#include "block_scalers.hpp"

namespace ifhip {

namespace {

struct WeightRow { int n, r, log2_div; int8_t w[8]; };
const WeightRow kRows[] = {
#include "block_scaler_weights.inc"
};
#include "block_scaler_luts.inc"

BlockScalerTables g_bs;
std::once_flag g_bs_once;
bool g_bs_ok = false;

void build_tables() {
    std::memset(&g_bs, 0, sizeof g_bs);
    for (int i = 0; i < 256; ++i) g_bs.srgb_to_linear[i] = kScalerS2L[i];
    for (int i = 0, v = 0; i < 4096; ++i) {                 // expand the thresholds back into lut_linear_to_srgb
        while (v < 255 && kScalerL2SThr[v] <= i) ++v;
        g_bs.linear_to_srgb[i] = static_cast<uint8_t>(v);
    }
    int rows = 0;
    for (const WeightRow& row : kRows) {
        if (row.n < 1 || row.n > 7 || row.r < 0 || row.r >= row.n) return;
        BlockScaler& s = g_bs.scaler[row.n];
        s.n = static_cast<uint32_t>(row.n);
        int first = 8, last = -1, sum = 0;
        for (int j = 0; j < 8; ++j) {
            s.w[row.r][j] = row.w[j];
            sum += row.w[j];
            if (row.w[j] != 0) { if (j < first) first = j; last = j; }
        }
        if (sum != (1 << row.log2_div)) return;             // a row's weights sum to its power-of-two divisor
        s.first[row.r] = static_cast<uint8_t>(first);
        s.last[row.r] = static_cast<uint8_t>(last);
        s.log2_div[row.r] = static_cast<uint8_t>(row.log2_div);
        ++rows;
    }
    // (the kernels' output clamp relies on the table's ends: below 0 -> entry 0 = 0, above 4095 -> entry 4095 = 255)
    g_bs_ok = rows == 28 && g_bs.linear_to_srgb[0] == 0 && g_bs.linear_to_srgb[4095] == 255;      // 28 = 1 + 2 + ... + 7
}

}  // namespace

const BlockScalerTables* block_scaler_tables() {
    std::call_once(g_bs_once, build_tables);
    return g_bs_ok ? &g_bs : nullptr;
}

}  // namespace ifhip
