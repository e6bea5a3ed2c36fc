// jpeg_write.cpp -- host side of the encode path: what libjpeg does AFTER the pixel stage when imageflow's classic
// encoder preset writes a file (codecs/mozjpeg.rs:78-160 with Defaults::LibJPEGv6 = set_fastest_defaults: baseline
// sequential, Annex K Huffman tables, no optimisation): jpeg_set_quality's table scaling (jcparam.c), the marker
// segments in jcmarker.c's order (SOI, JFIF APP0, DQT x2, SOF0, DHT x4, SOS, EOI) and the sequential Huffman encoder of
// jchuff.c (DC difference categories, AC run/size symbols with ZRL and EOB, byte stuffing, 1-padding of the last byte).
// The quantised coefficients come from the GPU stage (ifhip_jpeg_forward*); entropy coding is serial bit packing and
// stays on the host.  Output is byte-identical to libjpeg-turbo's for the same pixels,
// quality and sampling (tests/test_gpu_abi_shim.py compares with Pillow's encoder).
#include <cstdint>
#include <cstring>
#include <vector>

#include "common.hpp"

namespace {

const uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                             41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                             15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ITU T.81 Annex K.1 / K.2 (natural order) and K.3 (Huffman specifications), as jcparam.c / jstdhuff.c carry them
const uint8_t kStdLumaQ[64] = {16, 11, 10, 16, 24, 40, 51, 61, 12, 12, 14, 19, 26, 58, 60, 55, 14, 13, 16, 24, 40, 57, 69, 56,
                               14, 17, 22, 29, 51, 87, 80, 62, 18, 22, 37, 56, 68, 109, 103, 77, 24, 35, 55, 64, 81, 104, 113, 92,
                               49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 99};
const uint8_t kStdChromaQ[64] = {17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99, 24, 26, 56, 99, 99, 99, 99, 99,
                                 47, 66, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99,
                                 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99, 99};
const uint8_t kDcLumaBits[17] = {0, 0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0};
const uint8_t kDcChromaBits[17] = {0, 0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
const uint8_t kDcVals[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
const uint8_t kAcLumaBits[17] = {0, 0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d};
const uint8_t kAcLumaVals[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1,
    0x08, 0x23, 0x42, 0xb1, 0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26,
    0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56,
    0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85,
    0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa,
    0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4, 0xd5, 0xd6,
    0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};
const uint8_t kAcChromaBits[17] = {0, 0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
const uint8_t kAcChromaVals[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42,
    0x91, 0xa1, 0xb1, 0xc1, 0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19,
    0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36, 0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55,
    0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83,
    0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8,
    0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3, 0xd4,
    0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9,
    0xfa};

struct EncTab { uint16_t code[256]; uint8_t size[256]; };
void build(const uint8_t* bits, const uint8_t* vals, EncTab* t) {              // jchuff.c jpeg_make_c_derived_tbl
    std::memset(t, 0, sizeof *t);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++k, ++code) { t->code[vals[k]] = static_cast<uint16_t>(code); t->size[vals[k]] = static_cast<uint8_t>(l); }
        code <<= 1;
    }
}

struct BitWriter {
    std::vector<uint8_t>& out;
    uint32_t acc = 0;
    int n = 0;
    void put(uint32_t code, int size) {
        acc = (acc << size) | (code & ((1u << size) - 1u));
        n += size;
        while (n >= 8) {
            const uint8_t b = static_cast<uint8_t>(acc >> (n - 8));
            out.push_back(b);
            if (b == 0xFF) out.push_back(0);
            n -= 8;
        }
    }
    void flush() { if (n > 0) put(0x7F, 8 - n); }                                 // pad with 1 bits (jchuff.c flush_bits)
};

void marker(std::vector<uint8_t>& o, uint8_t m, const std::vector<uint8_t>& body) {
    o.push_back(0xFF); o.push_back(m);
    const size_t len = body.size() + 2;
    o.push_back(static_cast<uint8_t>(len >> 8)); o.push_back(static_cast<uint8_t>(len));
    o.insert(o.end(), body.begin(), body.end());
}
int nbits(int v) { int n = 0; while (v) { ++n; v >>= 1; } return n; }

}  // namespace

namespace ifhip {

// jpeg_set_quality(q, force_baseline = TRUE): qt[0] luma, qt[1] chroma, natural order
void jpeg_quality_tables(int quality, uint16_t qt[2][64]) {
    const int q = quality < 1 ? 1 : (quality > 100 ? 100 : quality);
    const int scale = q < 50 ? 5000 / q : 200 - 2 * q;
    for (int i = 0; i < 64; ++i) {
        long a = (static_cast<long>(kStdLumaQ[i]) * scale + 50) / 100, b = (static_cast<long>(kStdChromaQ[i]) * scale + 50) / 100;
        qt[0][i] = static_cast<uint16_t>(a < 1 ? 1 : (a > 255 ? 255 : a));
        qt[1][i] = static_cast<uint16_t>(b < 1 ? 1 : (b > 255 ? 255 : b));
    }
}

// coef[c]: [bh_c][bw_c][64] natural order, MCU padded (the layout of the GPU stage); 1 or 3 components; chroma tables = qt[1]
int jpeg_write_baseline(const int16_t* const coef[3], const uint32_t bw[3], const uint32_t bh[3], int ncomp, const uint8_t hs[3],
                        const uint8_t vs[3], uint32_t width, uint32_t height, const uint16_t qt[2][64], std::vector<uint8_t>* out) {
    if (!out || (ncomp != 1 && ncomp != 3) || width == 0 || height == 0 || width > 65535u || height > 65535u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: jpeg_write_baseline geometry");
    std::vector<uint8_t>& o = *out;
    o.clear();
    o.push_back(0xFF); o.push_back(0xD8);                                                        // SOI
    marker(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});                          // JFIF 1.01, density 1:1, no thumbnail
    for (int t = 0; t < (ncomp == 3 ? 2 : 1); ++t) {                                              // DQT, one marker per table, zigzag order
        std::vector<uint8_t> b(65);
        b[0] = static_cast<uint8_t>(t);
        for (int i = 0; i < 64; ++i) b[1 + i] = static_cast<uint8_t>(qt[t][kZigzag[i]]);
        marker(o, 0xDB, b);
    }
    {
        std::vector<uint8_t> b = {8, static_cast<uint8_t>(height >> 8), static_cast<uint8_t>(height), static_cast<uint8_t>(width >> 8),
                                  static_cast<uint8_t>(width), static_cast<uint8_t>(ncomp)};
        for (int c = 0; c < ncomp; ++c) { b.push_back(static_cast<uint8_t>(c + 1)); b.push_back(static_cast<uint8_t>((hs[c] << 4) | vs[c])); b.push_back(c ? 1 : 0); }
        marker(o, 0xC0, b);                                                                       // SOF0
    }
    EncTab dc[2], ac[2];
    build(kDcLumaBits, kDcVals, &dc[0]); build(kAcLumaBits, kAcLumaVals, &ac[0]);
    build(kDcChromaBits, kDcVals, &dc[1]); build(kAcChromaBits, kAcChromaVals, &ac[1]);
    auto dht = [&](int cls, int id, const uint8_t* bits, const uint8_t* vals, int nvals) {
        std::vector<uint8_t> b;
        b.push_back(static_cast<uint8_t>((cls << 4) | id));
        for (int l = 1; l <= 16; ++l) b.push_back(bits[l]);
        b.insert(b.end(), vals, vals + nvals);
        marker(o, 0xC4, b);
    };
    dht(0, 0, kDcLumaBits, kDcVals, 12); dht(1, 0, kAcLumaBits, kAcLumaVals, 162);
    if (ncomp == 3) { dht(0, 1, kDcChromaBits, kDcVals, 12); dht(1, 1, kAcChromaBits, kAcChromaVals, 162); }
    {
        std::vector<uint8_t> b = {static_cast<uint8_t>(ncomp)};
        for (int c = 0; c < ncomp; ++c) { b.push_back(static_cast<uint8_t>(c + 1)); b.push_back(c ? 0x11 : 0x00); }
        b.push_back(0); b.push_back(63); b.push_back(0);
        marker(o, 0xDA, b);                                                                       // SOS
    }
    // scan: MCUs in raster order, blocks of a component in raster order inside the MCU (a single component is not interleaved)
    const uint32_t hmax = ncomp == 3 ? std::max<uint32_t>(hs[0], std::max<uint32_t>(hs[1], hs[2])) : 1u;
    const uint32_t vmax = ncomp == 3 ? std::max<uint32_t>(vs[0], std::max<uint32_t>(vs[1], vs[2])) : 1u;
    const uint32_t mw = ncomp == 3 ? (width + 8u * hmax - 1u) / (8u * hmax) : (width + 7u) / 8u;
    const uint32_t mh = ncomp == 3 ? (height + 8u * vmax - 1u) / (8u * vmax) : (height + 7u) / 8u;
    BitWriter bwr{o};
    int pred[3] = {0, 0, 0};
    for (uint32_t my = 0; my < mh; ++my)
        for (uint32_t mx = 0; mx < mw; ++mx)
            for (int c = 0; c < ncomp; ++c) {
                const uint32_t H = ncomp == 3 ? hs[c] : 1u, V = ncomp == 3 ? vs[c] : 1u;
                const EncTab &D = dc[c ? 1 : 0], &A = ac[c ? 1 : 0];
                for (uint32_t dy = 0; dy < V; ++dy)
                    for (uint32_t dx = 0; dx < H; ++dx) {
                        const int16_t* blk = coef[c] + (static_cast<size_t>(my * V + dy) * bw[c] + (mx * H + dx)) * 64u;
                        int diff = blk[0] - pred[c];                                              // jchuff.c encode_one_block
                        pred[c] = blk[0];
                        int t = diff < 0 ? -diff : diff, t2 = diff < 0 ? diff - 1 : diff;
                        int nb = nbits(t);
                        bwr.put(D.code[nb], D.size[nb]);
                        if (nb) bwr.put(static_cast<uint32_t>(t2), nb);
                        int r = 0;
                        for (int k = 1; k < 64; ++k) {
                            const int v = blk[kZigzag[k]];
                            if (v == 0) { ++r; continue; }
                            while (r > 15) { bwr.put(A.code[0xF0], A.size[0xF0]); r -= 16; }
                            t = v < 0 ? -v : v; t2 = v < 0 ? v - 1 : v;
                            nb = nbits(t);
                            const int sym = (r << 4) + nb;
                            bwr.put(A.code[sym], A.size[sym]);
                            bwr.put(static_cast<uint32_t>(t2), nb);
                            r = 0;
                        }
                        if (r > 0) bwr.put(A.code[0], A.size[0]);                                 // EOB
                    }
            }
    bwr.flush();
    o.push_back(0xFF); o.push_back(0xD9);                                                        // EOI
    (void)bh;
    return IFHIP_OK;
}

}  // namespace ifhip

extern "C" {
int ifhip_jpeg_quality_tables(int quality, uint16_t* qt2x64) {
    if (!qt2x64) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    ifhip::jpeg_quality_tables(quality, reinterpret_cast<uint16_t(*)[64]>(qt2x64));
    return IFHIP_OK;
}
// Entropy-codes quantised coefficient planes (the output of ifhip_jpeg_forward*) into a baseline JFIF file with the
// Annex K tables: the host half of MozjpegEncoder::write_frame's classic preset.  Two-call pattern: out == NULL or
// capacity too small -> *len receives the size needed and IFHIP_INVALID_ARGUMENT is returned for the short case.
int ifhip_jpeg_write_baseline(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint32_t* blocks_w3,
                              const uint32_t* blocks_h3, int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                              uint32_t height, int quality, uint8_t* out, size_t capacity, size_t* len) {
    if (!coef0 || !blocks_w3 || !blocks_h3 || !len || (n_components == 3 && (!coef1 || !coef2 || !h_samp || !v_samp)))
        return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    uint16_t qt[2][64];
    ifhip::jpeg_quality_tables(quality, qt);
    const int16_t* coef[3] = {coef0, coef1, coef2};
    const uint8_t one[3] = {1, 1, 1};
    std::vector<uint8_t> bytes;
    try {
        const int rc = ifhip::jpeg_write_baseline(coef, blocks_w3, blocks_h3, n_components, n_components == 3 ? h_samp : one,
                                                  n_components == 3 ? v_samp : one, width, height, qt, &bytes);
        if (rc) return rc;
    } catch (const std::bad_alloc&) { return ifhip::fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory"); }
    *len = bytes.size();
    if (!out) return IFHIP_OK;
    if (capacity < bytes.size()) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: output capacity %zu < %zu", capacity, bytes.size());
    std::memcpy(out, bytes.data(), bytes.size());
    return IFHIP_OK;
}
}
