// jpeg_encode_core.hpp -- the arithmetic of the device entropy coder (csrc/jpeg_encode.hip), written so that it compiles
// for the gfx950 kernels AND for a plain host compiler: tests/enc_emulate.cpp runs the same block routine, the same scan
// order and the same bit / byte placement rules lane by lane on the CPU and compares the file with the host writer's
// (csrc/jpeg_write.cpp, byte-identical to libjpeg-turbo) -- test infrastructure; the product has one path, the kernels.
//
// What is coded: jchuff.c encode_one_block (sequential Huffman, baseline) for the interleaved single scan that
// MozjpegEncoder::write_frame produces with Defaults::LibJPEGv6 (codecs/mozjpeg.rs:108-112: set_fastest_defaults --
// Annex K tables, no optimisation, not progressive, no restart markers).
#pragma once
#include <cstdint>

#if defined(__HIPCC__)
#define IFHIP_HD __host__ __device__ __forceinline__
#else
#define IFHIP_HD inline
#endif

namespace ifhip {

// Geometry of one image's scan.  A block's scan index s = MCU index * blocks per MCU + position inside the MCU
// (jccoefct.c compress_data: components in order, each H x V blocks row-major); planes are [rows][pitch][64] int16,
// natural coefficient order, MCU padded (the layout ifhip_jpeg_forward_batch_device leaves).
struct EncGeom {
    uint32_t ncomp, bpm, mcus_w, mcus_h, nblocks;
    uint32_t H[3], V[3], pitch[3], rows[3];
    uint32_t first[3];          // position inside the MCU of the component's first block
    uint64_t layout;            // four bits per position inside the MCU: component | dx << 2 | dy << 3
    float rcp_bpm, rcp_mcus_w;  // 1 / bpm, 1 / mcus_w (enc_div)
};

// An index that differs from lane to lane must select among REGISTERS: left alone, the compiler turns a select between
// two fields of the argument block into a load from a selected address -- a memory round trip per field, one waiting for
// the other, in front of every workgroup's real work.  The fields are pinned to scalar registers first.
#if defined(__HIP_DEVICE_COMPILE__)
#define IFHIP_UNIFORM32(x) static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(x)))
#else
#define IFHIP_UNIFORM32(x) (x)
#endif
IFHIP_HD uint32_t enc_sel3(const uint32_t (&v)[3], uint32_t c) {
    const uint32_t v0 = IFHIP_UNIFORM32(v[0]), v1 = IFHIP_UNIFORM32(v[1]), v2 = IFHIP_UNIFORM32(v[2]);
    return c == 0u ? v0 : c == 1u ? v1 : v2;
}
IFHIP_HD uint64_t enc_sel3(const uint64_t (&v)[3], uint32_t c) {
    const uint64_t v0 = static_cast<uint64_t>(IFHIP_UNIFORM32(v[0] >> 32)) << 32 | IFHIP_UNIFORM32(v[0] & 0xffffffffu);
    const uint64_t v1 = static_cast<uint64_t>(IFHIP_UNIFORM32(v[1] >> 32)) << 32 | IFHIP_UNIFORM32(v[1] & 0xffffffffu);
    const uint64_t v2 = static_cast<uint64_t>(IFHIP_UNIFORM32(v[2] >> 32)) << 32 | IFHIP_UNIFORM32(v[2] & 0xffffffffu);
    return c == 0u ? v0 : c == 1u ? v1 : v2;
}

// n / d for n below 2^24 (scan indices are: enc_make_geom bounds the block count) without the 30-instruction integer
// division: the float quotient is at most one off in either direction, the remainder says which.
IFHIP_HD uint32_t enc_div(uint32_t n, uint32_t d, float rcp) {
    uint32_t q = static_cast<uint32_t>(static_cast<float>(n) * rcp);
    const int32_t r = static_cast<int32_t>(n - q * d);
    if (r < 0) --q;
    else if (static_cast<uint32_t>(r) >= d) ++q;
    return q;
}

// Where a block of the scan lies: component, offset in blocks inside the component's plane, and the same for the block
// whose DC value predicts it -- the component's previous block in scan order (0xFFFFFFFF: none, the predictor is 0:
// first MCU; jchuff.c start_pass_huff: last_dc_val = 0).
struct EncBlockRef { uint32_t comp, offset, pred_offset; };

IFHIP_HD EncBlockRef enc_locate(const EncGeom& g, uint32_t s) {
    const uint32_t m = enc_div(s, g.bpm, g.rcp_bpm), j = s - m * g.bpm;
    const uint32_t my = enc_div(m, g.mcus_w, g.rcp_mcus_w), mx = m - my * g.mcus_w;
    const uint32_t e = static_cast<uint32_t>(g.layout >> (4u * j)) & 15u, c = e & 3u, dx = (e >> 2) & 1u, dy = e >> 3;
    const uint32_t H = enc_sel3(g.H, c), V = enc_sel3(g.V, c), pitch = enc_sel3(g.pitch, c);
    EncBlockRef r{c, (my * V + dy) * pitch + mx * H + dx, 0xFFFFFFFFu};
    if (j != enc_sel3(g.first, c)) {                        // inside the MCU: the position before this one
        const uint32_t pe = static_cast<uint32_t>(g.layout >> (4u * (j - 1u))) & 15u;
        r.pred_offset = (my * V + (pe >> 3)) * pitch + mx * H + ((pe >> 2) & 1u);
    } else if (m != 0u) {                                   // the component's last block of the MCU before
        const uint32_t pmy = mx ? my : my - 1u, pmx = mx ? mx - 1u : g.mcus_w - 1u;
        r.pred_offset = (pmy * V + V - 1u) * pitch + pmx * H + H - 1u;
    }
    return r;
}

// Host: the geometry of a scan from the frame size, the sampling factors (1 or 2; a single component counts as 1x1,
// jcmaster.c) and the planes' sizes in blocks.  0: fine; 1: unsupported factors / sizes; 2: a plane smaller than the MCU
// grid; 3: too many blocks for 32-bit bit positions.
constexpr uint32_t kEncMaxBitsPerBlock = 16 + 11 + 63 * (16 + 10);
inline int enc_make_geom(uint32_t width, uint32_t height, int ncomp, const uint8_t* hs, const uint8_t* vs, const uint32_t* bw,
                         const uint32_t* bh, EncGeom* g) {
    if ((ncomp != 1 && ncomp != 3) || width == 0 || height == 0 || width > 65535u || height > 65535u) return 1;
    *g = EncGeom{};
    g->ncomp = static_cast<uint32_t>(ncomp);
    uint32_t hmax = 1, vmax = 1;
    for (int c = 0; c < ncomp; ++c) {
        const uint32_t H = ncomp == 3 ? hs[c] : 1u, V = ncomp == 3 ? vs[c] : 1u;
        if (H < 1 || H > 2 || V < 1 || V > 2) return 1;
        g->H[c] = H; g->V[c] = V;
        hmax = H > hmax ? H : hmax; vmax = V > vmax ? V : vmax;
    }
    g->mcus_w = (width + 8u * hmax - 1u) / (8u * hmax);
    g->mcus_h = (height + 8u * vmax - 1u) / (8u * vmax);
    uint32_t j = 0;
    for (int c = 0; c < ncomp; ++c) {
        g->first[c] = j;
        for (uint32_t y = 0; y < g->V[c]; ++y)
            for (uint32_t x = 0; x < g->H[c]; ++x, ++j)
                g->layout |= static_cast<uint64_t>(static_cast<uint32_t>(c) | x << 2 | y << 3) << (4u * j);
        g->pitch[c] = bw[c]; g->rows[c] = bh[c];
        if (bw[c] < g->mcus_w * g->H[c] || bh[c] < g->mcus_h * g->V[c]) return 2;
    }
    g->bpm = j;
    g->rcp_bpm = 1.0f / static_cast<float>(j);
    g->rcp_mcus_w = 1.0f / static_cast<float>(g->mcus_w);
    const uint64_t nb = static_cast<uint64_t>(g->mcus_w) * g->mcus_h * j;
    if (nb * kEncMaxBitsPerBlock >= (1ull << 32)) return 3;
    g->nblocks = static_cast<uint32_t>(nb);
    return 0;
}

// jpeg_natural_order
#define IFHIP_ZIGZAG_LIST                                                                                              \
    0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35,   \
        42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63

IFHIP_HD int enc_zigzag(int k) {
    constexpr int t[64] = {IFHIP_ZIGZAG_LIST};
    return t[k];
}

IFHIP_HD uint32_t enc_nbits(uint32_t t) { return t ? 32u - static_cast<uint32_t>(__builtin_clz(t)) : 0u; }

// Sinks: put(code, length) with length <= 32 and `code` < 2^length.
struct EncCountSink {
    uint32_t bits = 0;
    IFHIP_HD void put(uint32_t, uint32_t len) { bits += len; }
    IFHIP_HD void put_times(uint32_t, uint32_t len, uint32_t times) { bits += (len & 0xffffu) * (times & 0xffu); }    // (a 24-bit multiply-add)
};

// Bits go into a stream of big-endian 32-bit words that starts zeroed.  A lane's first word and its last, partial word
// are shared with its neighbours in the stream (OR into the word: Store::shared); the words between belong to it alone.
template <class Store>
struct EncWordSink {
    uint64_t acc = 0;
    uint32_t n;                 // bits waiting in acc (< 32 between calls); starts at the bit offset inside the first word
    uint32_t* p;
    bool first = true;
    IFHIP_HD EncWordSink(uint32_t* words, uint32_t bit_offset) : n(bit_offset & 31u), p(words + (bit_offset >> 5)) {}
    IFHIP_HD void put(uint32_t code, uint32_t len) {
        acc = (acc << len) | code;
        n += len;
        if (n >= 32u) {
            const uint32_t w = static_cast<uint32_t>(acc >> (n - 32u));
            n -= 32u;
            if (first) Store::shared(p, __builtin_bswap32(w));
            else Store::owned(p, __builtin_bswap32(w));
            first = false;
            ++p;
        }
    }
    IFHIP_HD void put_times(uint32_t code, uint32_t len, uint32_t times) {
        for (; times; --times) put(code, len);
    }
    IFHIP_HD void finish() {
        if (n) Store::shared(p, __builtin_bswap32(static_cast<uint32_t>(acc << (32u - n))));
    }
    IFHIP_HD uint32_t bits_in_last_byte() const { return n & 7u; }
};

// The window form: no accumulator, no "is a word full" test in the symbol loop -- every field is shifted to its place in
// the two words it may touch and ORed into both (the second OR is of zero when the field ends inside the first word; the
// window carries one spare word for it).  Words are kept most-significant-bit-first as numbers; the copy-out turns them
// into the stream's byte order.
template <class Store>
struct EncWindowSink {
    uint32_t* w;
    uint32_t pos;               // bit position in the window
    IFHIP_HD EncWindowSink(uint32_t* window, uint32_t bit_offset) : w(window), pos(bit_offset) {}
    IFHIP_HD void put(uint32_t code, uint32_t len) {
        const uint64_t f = static_cast<uint64_t>(code) << ((64u - (pos & 31u) - len) & 63u);   // (64 only for an empty field at a word's start)
        uint32_t* p = w + (pos >> 5);
        Store::shared(p, static_cast<uint32_t>(f >> 32));
        Store::shared(p + 1, static_cast<uint32_t>(f));
        pos += len;
    }
    IFHIP_HD void put_times(uint32_t code, uint32_t len, uint32_t times) {
        for (; times; --times) put(code, len);
    }
    IFHIP_HD void finish() {}
    IFHIP_HD uint32_t bits_in_last_byte() const { return pos & 7u; }
};

// The order a staged block is kept in (16-bit slots, two per dword): dword j of the first 16 holds zigzag position j in its
// low half and position 16 + j in its high half, dword 16 + j positions 32 + j and 48 + j likewise.  With one flag per half
// (coefficient != 0), shifting the dwords' flag pairs into one register one after the other and swapping the register's
// halves leaves position p at bit 31 - p: the nonzero mask of 32 positions costs two instructions per dword, and a
// position's slot is its low five bits rotated left by one.
IFHIP_HD uint32_t enc_slot(uint32_t k) { return (k & 32u) | ((k << 1) & 30u) | ((k >> 4) & 1u); }
IFHIP_HD uint32_t enc_position_of_slot(uint32_t h) { return (h & 32u) + ((h >> 1) & 15u) + ((h & 1u) << 4); }
// 1 in bit 0 / bit 16 for a nonzero low / high half
IFHIP_HD uint32_t enc_pair_flags(uint32_t w) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t f;                     // (spelled out: from the vector builtin the compiler makes two compares, two selects and a permute)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(f) : "v"(w), "s"(0x00010001u));
    return f;
#else
    return ((w & 0xffffu) ? 1u : 0u) | ((w >> 16) ? 0x10000u : 0u);
#endif
}

// jchuff.c encode_one_block.  `coef` gives the staged block: coef(k) the coefficient at zigzag position k (k may differ
// from lane to lane), coef.pair(j) dword j of the slot order above; dct / act: 256
// entries `code | length << 16` (jpeg_make_c_derived_tbl; a symbol the table does not have is 0).  Returns nonzero when a
// coefficient needs more magnitude bits than 8-bit JPEG has (JERR_BAD_DCT_COEF: 11 for the DC difference, 10 for an AC
// coefficient) -- the bits put are then meaningless but their count stays the same in every pass.
// The walk never looks at a zero coefficient twice: one pass over the 32 dwords collects a mask of the nonzero positions
// (position k at bit 63 - k), the symbol loop then jumps from set bit to set bit -- its trip count in a wave is the
// largest number of nonzero coefficients among the wave's 64 blocks, not 63.
template <class Coef, class Sink>
IFHIP_HD uint32_t enc_block(const Coef& coef, int32_t pred, const uint32_t* dct, const uint32_t* act, Sink& sink) {
    uint32_t bad = 0;
    uint32_t hi = 0, lo = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) hi = (hi << 1) | enc_pair_flags(coef.pair(j));
#pragma unroll
    for (int j = 16; j < 32; ++j) lo = (lo << 1) | enc_pair_flags(coef.pair(j));
    hi = (hi << 16) | (hi >> 16);                             // positions 0 .. 15 were the low halves
    lo = (lo << 16) | (lo >> 16);
    uint64_t m = (static_cast<uint64_t>(hi & 0x7fffffffu) << 32) | lo;      // (bit 63 would be the DC value)
    {
        const int32_t diff = coef(0) - pred;
        const uint32_t t = static_cast<uint32_t>(diff < 0 ? -diff : diff), t2 = static_cast<uint32_t>(diff < 0 ? diff - 1 : diff);
        uint32_t nb = enc_nbits(t);
        if (nb > 11u) { bad = 1u; nb = 11u; }
        const uint32_t cs = dct[nb];
        sink.put(((cs & 0xffffu) << nb) | (t2 & ((1u << nb) - 1u)), (cs >> 16) + nb);
    }
    const uint32_t zrl = act[0xF0], eob = act[0];
    uint32_t prev = 0, widest = 0;
    while (m) {
        const uint32_t k = static_cast<uint32_t>(__builtin_clzll(m));
        m &= ~(0x8000000000000000ull >> k);
        const uint32_t run = k - prev - 1u;
        prev = k;
        sink.put_times(zrl & 0xffffu, zrl >> 16, (run >> 4) & 3u);    // a ZRL symbol for every 16 zeros of the run (none: a no-op)
        const int32_t v = coef(static_cast<int>(k));                  // (not zero: the mask says so)
        const uint32_t t = static_cast<uint32_t>(v < 0 ? -v : v), t2 = static_cast<uint32_t>(v < 0 ? v - 1 : v);
        uint32_t nb = 32u - static_cast<uint32_t>(__builtin_clz(t));
        widest = nb > widest ? nb : widest;
        nb = nb > 10u ? 10u : nb;
        const uint32_t cs = act[((run & 15u) << 4) + nb];
        sink.put(((cs & 0xffffu) << nb) | (t2 & ((1u << nb) - 1u)), (cs >> 16) + nb);
    }
    bad |= widest > 10u ? 1u : 0u;
    if (prev != 63u) sink.put(eob & 0xffffu, eob >> 16);
    return bad;
}

// position in zigzag order of the coefficient at natural index n (the inverse of jpeg_natural_order)
IFHIP_HD int enc_zigzag_position(int n) {
    constexpr int t[64] = {0,  1,  5,  6,  14, 15, 27, 28, 2,  4,  7,  13, 16, 26, 29, 42, 3,  8,  12, 17, 25, 30, 41, 43, 9,  11, 18, 24, 31, 40, 44, 53,
                           10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};
    return t[n];
}

// number of 0xFF bytes in a word
IFHIP_HD uint32_t enc_count_ff(uint32_t w) {
    uint32_t m = w & (w >> 4);
    m &= m >> 2;
    m &= m >> 1;
    return static_cast<uint32_t>(__builtin_popcount(m & 0x01010101u));
}

constexpr uint32_t kEncBlocksPerWg = 256;           // scan-order blocks per workgroup of the count / write passes
constexpr uint32_t kEncChunkBytes = 4096;           // unstuffed stream bytes per workgroup step of the stuffing passes
// The write pass assembles a workgroup's piece of the stream in an LDS window of this many words when it fits (96 Kbit:
// 384 bits per block on average) and copies it out with coalesced stores -- only the window's first and last word are
// shared with the neighbouring workgroups; denser pieces are written to the stream directly, word by word.
constexpr uint32_t kEncWindowWords = 3072;
// words of the stream a workgroup's piece touches: `base` its first bit, `bits` its length (with the stream's final padding)
IFHIP_HD uint32_t enc_window_words(uint32_t base, uint32_t bits) { return bits ? ((base & 31u) + bits + 31u) >> 5 : 0u; }
// jchuff.c flush_bits: 1 bits up to the next byte boundary behind the stream's last bit
IFHIP_HD uint32_t enc_final_padding(uint32_t end_bit) { return (8u - (end_bit & 7u)) & 7u; }
// status bits per image
constexpr uint32_t kEncBadCoef = 1, kEncScanOverflow = 2, kEncFileOverflow = 4;

}  // namespace ifhip
