// color.cpp -- host-built colour tables, uploaded once per device.
//
// graphics/color.rs:22-45  ColorContext::new fills byte_to_float[n] = srgb_to_floatspace_uncached(n)
// graphics/color.rs:85-91  srgb_to_linear (f32, libm powf -- evaluated on the HOST, never on the device)
// graphics/lut.rs:14-271   LINEAR_TO_SRGB_LUT; regenerated from the f64 formula its own test uses
//                          (tests/integration/color_conversion.rs:381-388) and checked against the
//                          reference's table in tests/test_host_tables.py.
#include <cmath>
#include <mutex>

#include "common.hpp"

namespace ifhip {

static ColorTables g_tables;
static std::once_flag g_once;

static void fill_tables() {
    const float inv255 = 1.0f / 255.0f;
    for (int n = 0; n < 256; ++n) {
        const float s = static_cast<float>(n) * inv255;
        g_tables.s2f[n] = s;
        g_tables.s2l[n] = (s <= 0.04045f) ? s / 12.92f : powf((s + 0.055f) / (1.0f + 0.055f), 2.4f);
    }
    for (int i = 0; i < 16384; ++i) {
        const double lin = static_cast<double>(i) / 16383.0;
        const double enc = lin <= 0.0031308 ? 12.92 * lin : 1.055 * std::pow(lin, 1.0 / 2.4) - 0.055;
        double q = enc * 255.0 + 0.5;
        q = q < 0.0 ? 0.0 : (q > 255.0 ? 255.0 : q);
        g_tables.l2s[i] = static_cast<uint8_t>(q);
    }
    // The table is monotone, so it is fully described by where each output value first appears; the fused kernel
    // searches these 256 thresholds in LDS instead of holding the 16 KiB table there.
    for (int k = 0; k < 256; ++k) g_tables.l2s_thr[k] = 65535;
    for (int i = 16383; i >= 0; --i) {
        const int v = g_tables.l2s[i];
        for (int k = 0; k < v; ++k)
            if (g_tables.l2s_thr[k] > i) g_tables.l2s_thr[k] = static_cast<uint16_t>(i);
    }
}

const ColorTables& color_tables() {
    std::call_once(g_once, fill_tables);
    return g_tables;
}

}  // namespace ifhip
