// jpeg_write.cpp -- host side of the encode path: what libjpeg does AFTER the pixel stage when imageflow's classic
// encoder preset writes a file (codecs/mozjpeg.rs:78-160 with Defaults::LibJPEGv6 = set_fastest_defaults: baseline
// sequential, Annex K Huffman tables, no optimisation): jpeg_set_quality's table scaling (jcparam.c), the marker
// segments in jcmarker.c's order (SOI, JFIF APP0, DQT x2, SOF0, DHT x4, SOS, EOI) and the sequential Huffman encoder of
// jchuff.c (DC difference categories, AC run/size symbols with ZRL and EOB, byte stuffing, 1-padding of the last byte).
// The quantised coefficients come from the GPU stage (ifhip_jpeg_forward*); entropy coding is serial bit packing and
// stays on the host.  Output is byte-identical to libjpeg-turbo's for the same pixels,
// quality and sampling (tests/test_gpu_abi_shim.py compares with Pillow's encoder).  The preset's two options are here
// too: optimised Huffman tables (jpeg_gen_optimal_table) and progressive files (jpeg_simple_progression + jcphuff.c).
#include <cstdint>
#include <cstring>
#include <string>
#include <thread>
#include <vector>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif

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

struct EncTab { uint32_t cs[256]; };                                          // code | length << 16: one load per symbol
void build(const uint8_t* bits, const uint8_t* vals, EncTab* t) {              // jchuff.c jpeg_make_c_derived_tbl
    std::memset(t, 0, sizeof *t);
    uint32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        for (int i = 0; i < bits[l]; ++i, ++k, ++code) t->cs[vals[k]] = code | (static_cast<uint32_t>(l) << 16);
        code <<= 1;
    }
}

// Bit packing into the output vector: a 64-bit accumulator, four bytes at a time, byte stuffing only on the (rare)
// words that contain an 0xFF byte.  The vector is grown ahead of the write position and cut to size by finish().
struct BitWriter {
    std::vector<uint8_t>& out;
    uint8_t* p;                                      // write position and end of the vector's storage (raw pointers: byte stores
    uint8_t* lim;                                    // through the vector would make the compiler reload its fields every time)
    uint64_t acc = 0;
    int n = 0;                                       // bits waiting in acc (< 32 between calls)
    BitWriter(std::vector<uint8_t>& o, size_t expected_bytes) : out(o) {
        const size_t at = o.size();
        out.resize(at + expected_bytes + 64);
        p = out.data() + at;
        lim = out.data() + out.size();
    }
    void grow() {
        const size_t at = static_cast<size_t>(p - out.data());
        out.resize(out.size() * 2 + 64);
        p = out.data() + at;
        lim = out.data() + out.size();
    }
    void byte(uint8_t b) {
        *p++ = b;
        if (b == 0xFF) *p++ = 0;
    }
    void put(uint32_t code, int size) {              // size 0..31; bits of `code` above `size` are ignored
        acc = (acc << size) | (code & ((1u << size) - 1u));
        n += size;
        if (n >= 32) {
            if (lim - p < 16) grow();
            const uint32_t w = static_cast<uint32_t>(acc >> (n - 32));
            n -= 32;
            const uint32_t inv = ~w;
            if (((inv - 0x01010101u) & ~inv & 0x80808080u) == 0u) {           // no byte of w is 0xFF
                const uint32_t be = __builtin_bswap32(w);
                std::memcpy(p, &be, 4);
                p += 4;
            } else {
                byte(static_cast<uint8_t>(w >> 24)); byte(static_cast<uint8_t>(w >> 16)); byte(static_cast<uint8_t>(w >> 8)); byte(static_cast<uint8_t>(w));
            }
        }
    }
    void finish() {                                  // pad the last byte with 1 bits (jchuff.c flush_bits), cut the vector
        if (n & 7) put(0x7F, 8 - (n & 7));
        if (lim - p < 16) grow();
        while (n >= 8) { byte(static_cast<uint8_t>(acc >> (n - 8))); n -= 8; }
        out.resize(static_cast<size_t>(p - out.data()));
    }
};

void marker(std::vector<uint8_t>& o, uint8_t m, const std::vector<uint8_t>& body) {
    o.push_back(0xFF); o.push_back(m);
    const size_t len = body.size() + 2;
    o.push_back(static_cast<uint8_t>(len >> 8)); o.push_back(static_cast<uint8_t>(len));
    o.insert(o.end(), body.begin(), body.end());
}
inline int nbits(int v) { return v ? 32 - __builtin_clz(static_cast<unsigned>(v)) : 0; }

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

// ---- optimal Huffman tables and progressive scans (jchuff.c jpeg_gen_optimal_table, jcparam.c jpeg_simple_progression,
// jcphuff.c) -- what the classic preset's optimize_huffman_coding / progressive flags ask libjpeg for -----------------------
namespace {
struct HuffSpecW { uint8_t bits[17]; uint8_t vals[256]; int nvals; };

// jpeg_gen_optimal_table: code lengths by pairwise merging of the least frequent symbols (ties go to the larger symbol
// value), a reserved all-ones code point (pseudo-symbol 256), lengths limited to 16 bits by the Annex K.2 adjustment
void gen_optimal_table(const long counts[256], HuffSpecW* t) {
    long freq[257];
    int codesize[257], others[257];
    uint8_t bits[33];
    std::memset(bits, 0, sizeof bits);
    std::memset(codesize, 0, sizeof codesize);
    for (int i = 0; i < 257; ++i) { others[i] = -1; freq[i] = i < 256 ? counts[i] : 1; }
    for (;;) {
        int c1 = -1, c2 = -1;
        long v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v) { v = freq[i]; c1 = i; }
        v = 1000000000L;
        for (int i = 0; i <= 256; ++i) if (freq[i] && freq[i] <= v && i != c1) { v = freq[i]; c2 = i; }
        if (c2 < 0) break;
        freq[c1] += freq[c2];
        freq[c2] = 0;
        codesize[c1]++;
        while (others[c1] >= 0) { c1 = others[c1]; codesize[c1]++; }
        others[c1] = c2;
        codesize[c2]++;
        while (others[c2] >= 0) { c2 = others[c2]; codesize[c2]++; }
    }
    for (int i = 0; i <= 256; ++i) if (codesize[i]) bits[codesize[i] > 32 ? 32 : codesize[i]]++;
    int i;
    for (i = 32; i > 16; --i)
        while (bits[i] > 0) {
            int j = i - 2;
            while (j > 0 && bits[j] == 0) --j;
            bits[i] -= 2; bits[i - 1]++; bits[j + 1] += 2; bits[j]--;
        }
    while (i > 0 && bits[i] == 0) --i;               // the pseudo-symbol's code point: the longest code loses one
    if (i > 0) bits[i]--;                            // (i == 0: no symbol was counted at all -- an empty table)
    std::memcpy(t->bits, bits, 17);
    t->nvals = 0;
    for (int l = 1; l <= 32; ++l)
        for (int j = 0; j <= 255; ++j) if (codesize[j] == l) t->vals[t->nvals++] = static_cast<uint8_t>(j);
}

// A scan's symbols go either straight into the bit stream (tables known: the Annex K ones) or -- tables to be optimised --
// into statistics plus a token list that is replayed once the tables exist, so that the coefficients are walked once.
// token: value (extra bits) | n extra bits << 16 | symbol << 21 | is_ac << 29 | table << 30 | has symbol << 31
struct Coder {
    BitWriter* w = nullptr;                          // null: gather statistics + tokens
    const EncTab* dc[2] = {nullptr, nullptr};
    const EncTab* ac[2] = {nullptr, nullptr};
    long dc_count[2][256], ac_count[2][256];
    std::vector<uint32_t> tokens;
    bool bad = false;                                // a coefficient with more magnitude bits than 8-bit JPEG has (jchuff.c JERR_BAD_DCT_COEF)
    void range(int nb, int limit) { bad |= nb > limit; }
    void reset() { std::memset(dc_count, 0, sizeof dc_count); std::memset(ac_count, 0, sizeof ac_count); tokens.clear(); }
    void sym(bool is_ac, int tbl, int s, uint32_t extra, int n_extra) {
        if (s > 255) { bad = true; s = 255; }        // run 15 with 16 magnitude bits (a coefficient of -32768): refused by range(), never indexed
        if (w) {                                     // code and extra bits as one field (<= 16 + 11 bits)
            const uint32_t cs = (is_ac ? ac : dc)[tbl]->cs[s];
            w->put(((cs & 0xffffu) << n_extra) | (extra & ((1u << n_extra) - 1u)), static_cast<int>(cs >> 16) + n_extra);
        } else {
            (is_ac ? ac_count : dc_count)[tbl][s]++;
            tokens.push_back((extra & ((1u << n_extra) - 1u)) | (static_cast<uint32_t>(n_extra) << 16) | (static_cast<uint32_t>(s) << 21) |
                             (is_ac ? 1u << 29 : 0u) | (static_cast<uint32_t>(tbl) << 30) | (1u << 31));
        }
    }
    void raw(uint32_t v, int n) {                    // bits without a symbol (DC refinement, correction bits)
        if (w) w->put(v, n);
        else tokens.push_back((v & ((1u << n) - 1u)) | (static_cast<uint32_t>(n) << 16));
    }
    void replay(BitWriter& bw) const {
        for (const uint32_t t : tokens) {
            const int n_extra = static_cast<int>((t >> 16) & 31u);
            if (t >> 31) {
                const uint32_t cs = ((t >> 29) & 1u ? ac : dc)[(t >> 30) & 1u]->cs[(t >> 21) & 255u];
                bw.put(((cs & 0xffffu) << n_extra) | (t & 0xffffu), static_cast<int>(cs >> 16) + n_extra);
            } else {
                bw.put(t & 0xffffu, n_extra);
            }
        }
    }
};

struct Plane { const int16_t* coef; uint32_t pitch_blocks, wb, hb, H, V; int tbl; };   // wb x hb: the component's own size in blocks

struct ScanSpec { int ncomp; int comp[3]; int Ss, Se, Ah, Al; };

// The coefficient loops below never branch on "is this coefficient zero" (on photographic data that branch is a coin
// toss, and a misprediction costs more than the symbol): a bit mask of the block's positions with |coefficient| above a
// threshold is built eight coefficients at a time in natural order, permuted to zigzag order through a table, and its
// set bits are walked with count-trailing-zeros -- only coefficients that produce a symbol are ever loaded singly.
struct ZigzagMaskTable {
    uint64_t t[8][256];                              // natural row r, 8-bit mask of its columns -> the zigzag positions
    ZigzagMaskTable() {
        uint8_t zz_of[64];
        for (int k = 0; k < 64; ++k) zz_of[kZigzag[k]] = static_cast<uint8_t>(k);
        for (int r = 0; r < 8; ++r)
            for (int b = 0; b < 256; ++b) {
                uint64_t m = 0;
                for (int j = 0; j < 8; ++j) if (b & (1 << j)) m |= 1ull << zz_of[r * 8 + j];
                t[r][b] = m;
            }
    }
};
const ZigzagMaskTable kZzMask;

// zigzag positions k with |blk[k]| > above (above = 0: nonzero)
inline uint64_t zigzag_mask(const int16_t* blk, int above) {
    uint64_t nat = 0;                                // natural-order mask, bit i = coefficient i qualifies
#if defined(__SSE2__)
    const __m128i zero = _mm_setzero_si128(), lim = _mm_set1_epi16(static_cast<short>(above));
    for (int r = 0; r < 8; r += 2) {
        __m128i a = _mm_loadu_si128(reinterpret_cast<const __m128i*>(blk + r * 8)), b = _mm_loadu_si128(reinterpret_cast<const __m128i*>(blk + r * 8 + 8));
        a = _mm_max_epi16(a, _mm_subs_epi16(zero, a));                                // |x|, saturating: -32768 counts as 32767 (and is refused later)
        b = _mm_max_epi16(b, _mm_subs_epi16(zero, b));
        const __m128i q = _mm_packs_epi16(_mm_cmpgt_epi16(a, lim), _mm_cmpgt_epi16(b, lim));
        nat |= static_cast<uint64_t>(static_cast<uint32_t>(_mm_movemask_epi8(q))) << (r * 8);
    }
#else
    for (int i = 0; i < 64; ++i) { const int v = blk[i]; nat |= static_cast<uint64_t>((v < 0 ? -v : v) > above) << i; }
#endif
    uint64_t zz = 0;
    for (int r = 0; r < 8; ++r) zz |= kZzMask.t[r][(nat >> (8 * r)) & 255u];
    return zz;
}

// jchuff.c encode_one_block / htest_one_block
void sequential_block(Coder& C, const int16_t* blk, int tbl, int* pred) {
    int diff = blk[0] - *pred;
    *pred = blk[0];
    int t = diff < 0 ? -diff : diff, t2 = diff < 0 ? diff - 1 : diff;
    int nb = nbits(t);
    C.range(nb, 11);
    C.sym(false, tbl, nb, static_cast<uint32_t>(t2), nb);
    uint64_t m = zigzag_mask(blk, 0) & ~1ull;
    int prev = 0;
    while (m) {
        const int k = __builtin_ctzll(m);
        m &= m - 1;
        int r = k - prev - 1;
        prev = k;
        while (r > 15) { C.sym(true, tbl, 0xF0, 0, 0); r -= 16; }
        const int v = blk[kZigzag[k]];
        t = v < 0 ? -v : v; t2 = v < 0 ? v - 1 : v;
        nb = nbits(t);
        C.range(nb, 10);
        C.sym(true, tbl, (r << 4) + nb, static_cast<uint32_t>(t2), nb);
    }
    if (prev < 63) C.sym(true, tbl, 0, 0, 0);
}

// jcphuff.c: the four block coders of a progressive scan, with the end-of-band run and the buffered correction bits
struct Progressive {
    Coder& C;
    int tbl = 0, Ss = 0, Se = 0, Al = 0;
    uint32_t eobrun = 0;
    std::vector<uint8_t> be;                         // correction bits waiting behind the end-of-band run
    explicit Progressive(Coder& c) : C(c) { be.reserve(1024); }
    void buffered(const uint8_t* b, size_t n) {      // up to 16 one-bit values per token / put
        for (size_t i = 0; i < n;) {
            uint32_t v = 0;
            int m = 0;
            for (; m < 16 && i < n; ++m, ++i) v = (v << 1) | b[i];
            C.raw(v, m);
        }
    }
    void emit_eobrun() {
        if (eobrun > 0) {
            int nb = 0;
            for (uint32_t t = eobrun; (t >>= 1);) ++nb;
            C.sym(true, tbl, nb << 4, eobrun, nb);
            eobrun = 0;
            buffered(be.data(), be.size());
            be.clear();
        }
    }
    uint64_t band() const { return (Se == 63 ? ~0ull : (1ull << (Se + 1)) - 1ull) & ~((1ull << Ss) - 1ull); }
    void ac_first(const int16_t* blk) {
        uint64_t m = zigzag_mask(blk, (1 << Al) - 1) & band();        // |v| >> Al != 0
        int prev = Ss - 1;
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            int r = k - prev - 1;
            prev = k;
            const int v = blk[kZigzag[k]], a = (v < 0 ? -v : v) >> Al;
            if (eobrun > 0) emit_eobrun();
            while (r > 15) { C.sym(true, tbl, 0xF0, 0, 0); r -= 16; }
            const int nb = nbits(a);
            C.range(nb, 10);
            C.sym(true, tbl, (r << 4) + nb, static_cast<uint32_t>(v < 0 ? ~a : a), nb);       // (jcphuff.c: ~magnitude when negative)
        }
        if (prev < Se) { if (++eobrun == 0x7FFF) emit_eobrun(); }
    }
    void ac_refine(const int16_t* blk) {
        const uint64_t nz = zigzag_mask(blk, (1 << Al) - 1) & band();  // positions whose magnitude (>> Al) is not zero
        int absv[64], eob = 0;
        for (uint64_t m = nz; m; m &= m - 1) {
            const int k = __builtin_ctzll(m), v = blk[kZigzag[k]];
            absv[k] = (v < 0 ? -v : v) >> Al;
            if (absv[k] == 1) eob = k;               // ascending k: the last newly nonzero coefficient
        }
        int r = 0, prev = Ss - 1;
        uint8_t br[64];                              // correction bits of this block since the last newly nonzero coefficient
        size_t nbr = 0;
        uint64_t m = nz;
        while (m) {
            const int k = __builtin_ctzll(m);
            m &= m - 1;
            r += k - prev - 1;                       // the zero-history coefficients passed on the way
            prev = k;
            const int t = absv[k];
            while (r > 15 && k <= eob) {
                emit_eobrun();
                C.sym(true, tbl, 0xF0, 0, 0);
                r -= 16;
                buffered(br, nbr);
                nbr = 0;
            }
            if (t > 1) { br[nbr++] = static_cast<uint8_t>(t & 1); continue; }
            emit_eobrun();
            C.sym(true, tbl, (r << 4) + 1, blk[kZigzag[k]] < 0 ? 0u : 1u, 1);
            buffered(br, nbr);
            nbr = 0;
            r = 0;
        }
        r += Se - prev;                              // zeros behind the last nonzero position
        if (r > 0 || nbr > 0) {
            ++eobrun;
            be.insert(be.end(), br, br + nbr);
            if (eobrun == 0x7FFF || be.size() > 1000 - 64 + 1) emit_eobrun();
        }
    }
};

// one scan, counted or written; MCU order for interleaved scans, the component's own block raster otherwise
void run_scan(Coder& C, const ScanSpec& sc, const Plane* planes, uint32_t mcus_w, uint32_t mcus_h) {
    Progressive P(C);
    P.Ss = sc.Ss; P.Se = sc.Se; P.Al = sc.Al;
    const bool sequential = sc.Ss == 0 && sc.Se == 63;
    int pred[3] = {0, 0, 0};
    auto block = [&](int ci, const int16_t* blk) {
        const Plane& pl = planes[ci];
        if (sequential) { sequential_block(C, blk, pl.tbl, &pred[ci]); return; }
        if (sc.Ss == 0) {                            // DC scan
            if (sc.Ah == 0) {
                const int t2 = blk[0] >> sc.Al;      // (arithmetic shift, jcphuff.c IRIGHT_SHIFT)
                int diff = t2 - pred[ci];
                pred[ci] = t2;
                int t = diff < 0 ? -diff : diff, tb = diff < 0 ? diff - 1 : diff;
                const int nb = nbits(t);
                C.range(nb, 11);
                C.sym(false, pl.tbl, nb, static_cast<uint32_t>(tb), nb);
            } else {
                C.raw(static_cast<uint32_t>(blk[0] >> sc.Al) & 1u, 1);
            }
            return;
        }
        P.tbl = pl.tbl;
        if (sc.Ah == 0) P.ac_first(blk); else P.ac_refine(blk);
    };
    if (sc.ncomp > 1) {
        for (uint32_t my = 0; my < mcus_h; ++my)
            for (uint32_t mx = 0; mx < mcus_w; ++mx)
                for (int i = 0; i < sc.ncomp; ++i) {
                    const Plane& pl = planes[sc.comp[i]];
                    for (uint32_t dy = 0; dy < pl.V; ++dy)
                        for (uint32_t dx = 0; dx < pl.H; ++dx)
                            block(sc.comp[i], pl.coef + (static_cast<size_t>(my * pl.V + dy) * pl.pitch_blocks + (mx * pl.H + dx)) * 64u);
                }
    } else {
        const Plane& pl = planes[sc.comp[0]];
        for (uint32_t by = 0; by < pl.hb; ++by)
            for (uint32_t bx = 0; bx < pl.wb; ++bx) block(sc.comp[0], pl.coef + (static_cast<size_t>(by) * pl.pitch_blocks + bx) * 64u);
    }
    P.emit_eobrun();
}
}  // namespace

// coef[c]: [bh_c][bw_c][64] natural order, MCU padded (the layout of the GPU stage); 1 or 3 components; chroma tables = qt[1].
// flags: IFHIP_JPEG_OPTIMIZE_HUFFMAN (two passes: symbol statistics, then jpeg_gen_optimal_table's codes),
// IFHIP_JPEG_PROGRESSIVE (SOF2, jpeg_simple_progression's scan script, every scan with its own optimal tables).
int jpeg_write(const int16_t* const coef[3], const uint32_t bw[3], const uint32_t bh[3], int ncomp, const uint8_t hs[3],
               const uint8_t vs[3], uint32_t width, uint32_t height, const uint16_t qt[2][64], int flags, std::vector<uint8_t>* out) {
    if (!out || (ncomp != 1 && ncomp != 3) || width == 0 || height == 0 || width > 65535u || height > 65535u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: jpeg_write geometry");
    const bool progressive = (flags & 2) != 0, optimize = progressive || (flags & 1) != 0;
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
        marker(o, progressive ? 0xC2 : 0xC0, b);                                                  // SOF0 / SOF2
    }
    const uint32_t hmax = ncomp == 3 ? std::max<uint32_t>(hs[0], std::max<uint32_t>(hs[1], hs[2])) : 1u;
    const uint32_t vmax = ncomp == 3 ? std::max<uint32_t>(vs[0], std::max<uint32_t>(vs[1], vs[2])) : 1u;
    const uint32_t mw = (width + 8u * hmax - 1u) / (8u * hmax), mh = (height + 8u * vmax - 1u) / (8u * vmax);
    Plane planes[3];
    for (int c = 0; c < ncomp; ++c) {
        const uint32_t H = ncomp == 3 ? hs[c] : 1u, V = ncomp == 3 ? vs[c] : 1u;
        // jcmaster.c: a component's own size in blocks (non-interleaved scans cover exactly these)
        const uint32_t wb = (width * H + hmax * 8u - 1u) / (hmax * 8u), hb = (height * V + vmax * 8u - 1u) / (vmax * 8u);
        planes[c] = Plane{coef[c], bw[c], wb, hb, H, V, c ? 1 : 0};
        if (bw[c] < mw * H || bh[c] < mh * V) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient plane %d smaller than the MCU grid", c);
    }
    std::vector<ScanSpec> script;
    if (!progressive) {
        ScanSpec s{ncomp, {0, 1, 2}, 0, 63, 0, 0};
        script.push_back(s);
    } else if (ncomp == 3) {                                                                      // jpeg_simple_progression, YCbCr
        script = {{3, {0, 1, 2}, 0, 0, 0, 1}, {1, {0, 0, 0}, 1, 5, 0, 2}, {1, {2, 0, 0}, 1, 63, 0, 1}, {1, {1, 0, 0}, 1, 63, 0, 1},
                  {1, {0, 0, 0}, 6, 63, 0, 2}, {1, {0, 0, 0}, 1, 63, 2, 1}, {3, {0, 1, 2}, 0, 0, 1, 0}, {1, {2, 0, 0}, 1, 63, 1, 0},
                  {1, {1, 0, 0}, 1, 63, 1, 0}, {1, {0, 0, 0}, 1, 63, 1, 0}};
    } else {
        script = {{1, {0, 0, 0}, 0, 0, 0, 1}, {1, {0, 0, 0}, 1, 5, 0, 2}, {1, {0, 0, 0}, 6, 63, 0, 2}, {1, {0, 0, 0}, 1, 63, 2, 1},
                  {1, {0, 0, 0}, 0, 0, 1, 0}, {1, {0, 0, 0}, 1, 63, 1, 0}};
    }
    auto dht = [&](int cls, int id, const uint8_t* bits, const uint8_t* vals, int nvals) {
        std::vector<uint8_t> b;
        b.push_back(static_cast<uint8_t>((cls << 4) | id));
        for (int l = 1; l <= 16; ++l) b.push_back(bits[l]);
        b.insert(b.end(), vals, vals + nvals);
        marker(o, 0xC4, b);
    };
    Coder C;                                                                                      // (its token list keeps its capacity from scan to scan)
    for (const ScanSpec& sc : script) {
        const bool dc_scan = sc.Ss == 0, ac_scan = sc.Se > 0;
        const bool needs_dc = dc_scan && sc.Ah == 0, needs_ac = ac_scan;                          // (a DC refinement scan is raw bits)
        EncTab dct[2], act[2];
        C.w = nullptr;
        C.dc[0] = &dct[0]; C.dc[1] = &dct[1]; C.ac[0] = &act[0]; C.ac[1] = &act[1];
        bool used[2] = {false, false};
        for (int i = 0; i < sc.ncomp; ++i) used[planes[sc.comp[i]].tbl] = true;
        HuffSpecW od[2], oa[2];
        if (optimize) {
            C.reset();
            run_scan(C, sc, planes, mw, mh);                                                      // statistics + tokens
            for (int t = 0; t < 2; ++t) {
                if (!used[t]) continue;
                if (needs_dc) { gen_optimal_table(C.dc_count[t], &od[t]); build(od[t].bits, od[t].vals, &dct[t]); }
                if (needs_ac) { gen_optimal_table(C.ac_count[t], &oa[t]); build(oa[t].bits, oa[t].vals, &act[t]); }
            }
        } else {
            build(kDcLumaBits, kDcVals, &dct[0]); build(kAcLumaBits, kAcLumaVals, &act[0]);
            build(kDcChromaBits, kDcVals, &dct[1]); build(kAcChromaBits, kAcChromaVals, &act[1]);
        }
        // jcmarker.c write_scan_header: the tables of the scan's components in component order, DC then AC, each once
        bool sent_dc[2] = {false, false}, sent_ac[2] = {false, false};
        for (int i = 0; i < sc.ncomp; ++i) {
            const int t = planes[sc.comp[i]].tbl;
            if (needs_dc && !sent_dc[t]) {
                sent_dc[t] = true;
                if (optimize) dht(0, t, od[t].bits, od[t].vals, od[t].nvals);
                else dht(0, t, t ? kDcChromaBits : kDcLumaBits, kDcVals, 12);
            }
            if (needs_ac && !sent_ac[t]) {
                sent_ac[t] = true;
                if (optimize) dht(1, t, oa[t].bits, oa[t].vals, oa[t].nvals);
                else dht(1, t, t ? kAcChromaBits : kAcLumaBits, t ? kAcChromaVals : kAcLumaVals, 162);
            }
        }
        {
            std::vector<uint8_t> b = {static_cast<uint8_t>(sc.ncomp)};
            for (int i = 0; i < sc.ncomp; ++i) {
                const int c = sc.comp[i], t = planes[c].tbl;
                b.push_back(static_cast<uint8_t>(c + 1));
                // jcmarker.c emit_sos: a progressive scan names only the table it uses (DC scans: no AC table; AC scans: no DC table)
                const int td = progressive ? (dc_scan && sc.Ah == 0 ? t : 0) : t, ta = progressive ? (ac_scan ? t : 0) : t;
                b.push_back(static_cast<uint8_t>((td << 4) | ta));
            }
            b.push_back(static_cast<uint8_t>(sc.Ss)); b.push_back(static_cast<uint8_t>(sc.Se));
            b.push_back(static_cast<uint8_t>((sc.Ah << 4) | sc.Al));
            marker(o, 0xDA, b);                                                                   // SOS
        }
        BitWriter bwr(o, optimize ? C.tokens.size() * 2u : static_cast<size_t>(mw) * mh * 24u);
        if (optimize) C.replay(bwr);
        else { C.w = &bwr; run_scan(C, sc, planes, mw, mh); }
        bwr.finish();
        if (C.bad) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient out of range for 8-bit JPEG (more than 11 DC / 10 AC magnitude bits)");
    }
    o.push_back(0xFF); o.push_back(0xD9);                                                        // EOI
    return IFHIP_OK;
}

// For the device coder (csrc/jpeg_encode.hip): the Annex K tables in encode form (dc0, ac0, dc1, ac1) and the marker
// segments jpeg_write puts in front of a baseline file's scan -- SOI, JFIF APP0, DQT per table, SOF0, the DHT segments of the
// scan's components in jcmarker.c's order, SOS (tests compare whole files of the two coders).
void jpeg_std_encode_tables(uint32_t tabs[4][256]) {
    EncTab t[4];
    build(kDcLumaBits, kDcVals, &t[0]); build(kAcLumaBits, kAcLumaVals, &t[1]);
    build(kDcChromaBits, kDcVals, &t[2]); build(kAcChromaBits, kAcChromaVals, &t[3]);
    for (int i = 0; i < 4; ++i) std::memcpy(tabs[i], t[i].cs, sizeof t[i].cs);
}

int jpeg_baseline_header(int ncomp, const uint8_t hs[3], const uint8_t vs[3], uint32_t width, uint32_t height, const uint16_t qt[2][64],
                         std::vector<uint8_t>* out) {
    if (!out || (ncomp != 1 && ncomp != 3) || width == 0 || height == 0 || width > 65535u || height > 65535u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: jpeg header geometry");
    std::vector<uint8_t>& o = *out;
    o.clear();
    o.push_back(0xFF); o.push_back(0xD8);
    marker(o, 0xE0, {'J', 'F', 'I', 'F', 0, 1, 1, 0, 0, 1, 0, 1, 0, 0});
    for (int t = 0; t < (ncomp == 3 ? 2 : 1); ++t) {
        std::vector<uint8_t> b(65);
        b[0] = static_cast<uint8_t>(t);
        for (int i = 0; i < 64; ++i) b[1 + i] = static_cast<uint8_t>(qt[t][kZigzag[i]]);
        marker(o, 0xDB, b);
    }
    {
        std::vector<uint8_t> b = {8, static_cast<uint8_t>(height >> 8), static_cast<uint8_t>(height), static_cast<uint8_t>(width >> 8),
                                  static_cast<uint8_t>(width), static_cast<uint8_t>(ncomp)};
        for (int c = 0; c < ncomp; ++c) { b.push_back(static_cast<uint8_t>(c + 1)); b.push_back(static_cast<uint8_t>((hs[c] << 4) | vs[c])); b.push_back(c ? 1 : 0); }
        marker(o, 0xC0, b);
    }
    for (int t = 0; t < (ncomp == 3 ? 2 : 1); ++t) {
        for (int cls = 0; cls < 2; ++cls) {
            const uint8_t* bits = cls ? (t ? kAcChromaBits : kAcLumaBits) : (t ? kDcChromaBits : kDcLumaBits);
            const uint8_t* vals = cls ? (t ? kAcChromaVals : kAcLumaVals) : kDcVals;
            std::vector<uint8_t> b;
            b.push_back(static_cast<uint8_t>((cls << 4) | t));
            for (int l = 1; l <= 16; ++l) b.push_back(bits[l]);
            b.insert(b.end(), vals, vals + (cls ? 162 : 12));
            marker(o, 0xC4, b);
        }
    }
    std::vector<uint8_t> b = {static_cast<uint8_t>(ncomp)};
    for (int c = 0; c < ncomp; ++c) { b.push_back(static_cast<uint8_t>(c + 1)); b.push_back(static_cast<uint8_t>(c ? 0x11 : 0x00)); }
    b.push_back(0); b.push_back(63); b.push_back(0);
    marker(o, 0xDA, b);
    return IFHIP_OK;
}

int jpeg_write_baseline(const int16_t* const coef[3], const uint32_t bw[3], const uint32_t bh[3], int ncomp, const uint8_t hs[3],
                        const uint8_t vs[3], uint32_t width, uint32_t height, const uint16_t qt[2][64], std::vector<uint8_t>* out) {
    return jpeg_write(coef, bw, bh, ncomp, hs, vs, width, height, qt, 0, out);
}

}  // namespace ifhip

extern "C" {
int ifhip_jpeg_debug_encode_tables(uint32_t* tabs4x256, int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                                   uint32_t height, int quality, uint8_t* header, size_t capacity, size_t* header_len) {
    if (!tabs4x256 || !header_len) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    ifhip::jpeg_std_encode_tables(reinterpret_cast<uint32_t(*)[256]>(tabs4x256));
    uint16_t qt[2][64];
    ifhip::jpeg_quality_tables(quality, qt);
    std::vector<uint8_t> h;
    const uint8_t ones[3] = {1, 1, 1};
    int rc = ifhip::jpeg_baseline_header(n_components, h_samp ? h_samp : ones, v_samp ? v_samp : ones, width, height, qt, &h);
    if (rc) return rc;
    *header_len = h.size();
    if (header && capacity >= h.size()) std::memcpy(header, h.data(), h.size());
    return IFHIP_OK;
}

int ifhip_jpeg_quality_tables(int quality, uint16_t* qt2x64) {
    if (!qt2x64) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    ifhip::jpeg_quality_tables(quality, reinterpret_cast<uint16_t(*)[64]>(qt2x64));
    return IFHIP_OK;
}
// Entropy-codes quantised coefficient planes (the output of ifhip_jpeg_forward*) into a baseline JFIF file with the
// Annex K tables: the host half of MozjpegEncoder::write_frame's classic preset.  Two-call pattern: out == NULL or
// capacity too small -> *len receives the size needed and IFHIP_INVALID_ARGUMENT is returned for the short case.
int ifhip_jpeg_write(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint32_t* blocks_w3,
                     const uint32_t* blocks_h3, int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                     uint32_t height, int quality, int flags, uint8_t* out, size_t capacity, size_t* len) {
    if (!coef0 || !blocks_w3 || !blocks_h3 || !len || (n_components == 3 && (!coef1 || !coef2 || !h_samp || !v_samp)))
        return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    uint16_t qt[2][64];
    ifhip::jpeg_quality_tables(quality, qt);
    const int16_t* coef[3] = {coef0, coef1, coef2};
    const uint8_t one[3] = {1, 1, 1};
    std::vector<uint8_t> bytes;
    try {
        if (flags & ~3) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: unknown flags 0x%x", flags);
        const int rc = ifhip::jpeg_write(coef, blocks_w3, blocks_h3, n_components, n_components == 3 ? h_samp : one,
                                         n_components == 3 ? v_samp : one, width, height, qt, flags, &bytes);
        if (rc) return rc;
    } catch (const std::bad_alloc&) { return ifhip::fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory"); }
    *len = bytes.size();
    if (!out) return IFHIP_OK;
    if (capacity < bytes.size()) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: output capacity %zu < %zu", capacity, bytes.size());
    std::memcpy(out, bytes.data(), bytes.size());
    return IFHIP_OK;
}
// A batch: n images of one geometry, coefficient planes [n][bh_c][bw_c][64] as the device stage leaves them (downloaded
// to the host), coded on up to `threads` host threads (0: one per core, at most n) -- entropy coding is the serial part of
// an encode, but serial per image only.  Files land in `out` at offsets[i] .. offsets[i] + lengths[i], packed in image
// order; out == NULL or a short capacity: *total receives the size needed (and IFHIP_INVALID_ARGUMENT for the short case).
int ifhip_jpeg_write_batch(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint32_t* blocks_w3,
                           const uint32_t* blocks_h3, int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                           uint32_t height, int quality, int flags, uint32_t n_images, uint32_t threads, uint8_t* out, size_t capacity,
                           size_t* offsets, size_t* lengths, size_t* total) {
    if (!coef0 || !blocks_w3 || !blocks_h3 || !total || n_images == 0 || (n_components == 3 && (!coef1 || !coef2 || !h_samp || !v_samp)))
        return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer or empty batch");
    if (flags & ~3) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: unknown flags 0x%x", flags);
    if (n_components != 1 && n_components != 3) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: 1 or 3 components");
    try {
        uint16_t qt[2][64];
        ifhip::jpeg_quality_tables(quality, qt);
        const uint8_t one[3] = {1, 1, 1};
        size_t plane[3] = {0, 0, 0};
        for (int c = 0; c < n_components; ++c) plane[c] = static_cast<size_t>(blocks_w3[c]) * blocks_h3[c] * 64u;
        std::vector<std::vector<uint8_t>> files(n_images);
        std::vector<int> rcs(n_images, IFHIP_OK);
        std::vector<std::string> messages(n_images);
        const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
        const uint32_t n_threads = std::min<uint32_t>(n_images, threads ? threads : hw);
        auto work = [&](uint32_t t) {
            for (uint32_t i = t; i < n_images; i += n_threads) {
                try {
                    const int16_t* coef[3] = {coef0 + plane[0] * i, coef1 ? coef1 + plane[1] * i : nullptr, coef2 ? coef2 + plane[2] * i : nullptr};
                    rcs[i] = ifhip::jpeg_write(coef, blocks_w3, blocks_h3, n_components, n_components == 3 ? h_samp : one,
                                               n_components == 3 ? v_samp : one, width, height, qt, flags, &files[i]);
                    if (rcs[i]) messages[i] = ifhip::last_error();
                } catch (const std::bad_alloc&) { rcs[i] = IFHIP_ALLOCATION_FAILED; messages[i] = "AllocationFailed: host memory"; }
            }
        };
        std::vector<std::thread> pool;
        std::vector<uint32_t> inline_lanes{0u};
        for (uint32_t t = 1; t < n_threads; ++t) {
            try { pool.emplace_back(work, t); } catch (const std::exception&) { inline_lanes.push_back(t); }     // thread limits: the caller's thread takes the share
        }
        for (uint32_t t : inline_lanes) work(t);
        for (auto& th : pool) th.join();
        size_t sum = 0;
        for (uint32_t i = 0; i < n_images; ++i) {
            if (rcs[i]) return ifhip::fail(rcs[i], "%s (image %u of the batch)", messages[i].c_str(), i);
            sum += files[i].size();
        }
        *total = sum;
        if (!out) return IFHIP_OK;
        if (capacity < sum) return ifhip::fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: output capacity %zu < %zu", capacity, sum);
        size_t at = 0;
        for (uint32_t i = 0; i < n_images; ++i) {
            std::memcpy(out + at, files[i].data(), files[i].size());
            if (offsets) offsets[i] = at;
            if (lengths) lengths[i] = files[i].size();
            at += files[i].size();
        }
        return IFHIP_OK;
    } catch (const std::bad_alloc&) { return ifhip::fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory"); }
      catch (const std::exception& ex) { return ifhip::fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what()); }
}
int ifhip_jpeg_write_baseline(const int16_t* coef0, const int16_t* coef1, const int16_t* coef2, const uint32_t* blocks_w3,
                              const uint32_t* blocks_h3, int n_components, const uint8_t* h_samp, const uint8_t* v_samp, uint32_t width,
                              uint32_t height, int quality, uint8_t* out, size_t capacity, size_t* len) {
    return ifhip_jpeg_write(coef0, coef1, coef2, blocks_w3, blocks_h3, n_components, h_samp, v_samp, width, height, quality, 0, out, capacity, len);
}
}
