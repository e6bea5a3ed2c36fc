// jpeg_entropy.hip -- baseline (sequential Huffman) JPEG entropy decoding on gfx950.
//
// Replaces the serial jpeg_read_coefficients / decode_mcu loop that MzDec::read_frame drives on the host
// (codecs/mozjpeg_decoder.rs:346-362 -> mozjpeg jdhuff.c) and that caps the JPEG path at ~50 MP/s per core: the scan's
// bit stream is decoded by thousands of lanes at once and lands as the quantised coefficient planes the pixel stage
// (jpeg_kernels.hip) consumes, so a compressed file is the only thing that crosses PCIe.
//
// Host (this file, plain C++): marker parsing (SOF0/SOF1, DHT, DQT, DRI, SOS), byte un-stuffing, splitting the scan at
// restart markers into *segments* (independent bit streams with a known start state), canonical Huffman tables.
// Device: the self-synchronising parallel decode (Klein & Wiseman; Weissenberger & Schmidt, ICPP 2018):
//   * the stream is cut into sub-sequences of 1024 bits, one lane each;
//   * round 0 decodes every sub-sequence speculatively from state (block 0 of the MCU, coefficient 0); because Huffman
//     codes re-synchronise, most lanes end in the true state; then, inside the workgroup, every lane whose predecessor's
//     exit state moved decodes again from it, until nothing moves (first lanes of a segment are exact, so the fixpoint is
//     the serial decode); workgroups warm up on the sub-sequences in front of their range, so nothing crosses them;
//   * the count pass walks every sub-sequence from its final entry state: blocks started and DC sums per component (and
//     it checks the fixpoint: only if a sub-sequence was decoded from another state than its predecessor's exit does the
//     host run further rounds);
//   * the write pass scans those counts (every lane's output block and DC predictors), decodes once more and stores
//     coefficients (natural order) straight into the planes.
// Integer / table work, bound by dependent bit-serial decoding per lane and L1/L2 latency, not by HBM (no MFMA).
// Coefficient-exact against oracle/jpeg_oracle.c jo_jpeg_read_coefficients (itself pinned to libjpeg-turbo).
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <type_traits>
#include <cstdio>
#include <vector>

#include "common.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace ifhip {

constexpr uint32_t kSubBits = 1024;     // bits per sub-sequence (one lane)
constexpr uint32_t kSubWords = kSubBits / 32;
constexpr uint32_t kLutBits = 9;

// Huffman tables in the form every pass reads.  Everything a pass needs from a symbol sits in one 32-bit entry, found
// with ONE lookup by the next kLutBits bits of the stream -- or two for the 2 % of the symbols whose code is longer:
//   bits 0-7   bits to skip (code length + magnitude bits); 0 = the code is longer than the lookup
//   bits 8-15  zigzag advance: DC 1; AC coefficient run + 1; ZRL 16; EOB 64 (reaching 64 ends the block)
//   bits 16-23 code length (32: no such code), bits 24-31 the symbol.
// A first-level entry with skip 0 points into the image's second level (`pool`): bits 16-31 the offset of a sub-table
// indexed by the n bits that follow the first kLutBits (n = longest code with this prefix - kLutBits), bits 8-15 hold
// 32 - n.  jdhuff.c's slow path (the serial "first l with code_l <= maxcode[l]" search) survives only for sub-tables
// that did not fit the pool (first-level entry 0): it reads SearchTab from global memory and no real file gets there.
constexpr uint32_t kLutEntries = 1u << kLutBits;
constexpr uint32_t kPoolEntries = 768;
static_assert(kLutBits >= 8u && kLutBits <= 11u, "first-level lookup: the pair entries assume a code + magnitude of < 32 bits behind it");
static_assert(kPoolEntries >= 128u && kPoolEntries <= 65535u, "a first-level pointer holds the pool offset in 16 bits; one sub-table is up to 128 entries");
struct FastTabs {                                    // one image: [comp][dc, ac] and their shared second level
    uint32_t lut[6][kLutEntries];
    uint32_t pool[kPoolEntries];
};
struct SearchTab {                                   // jdhuff.c jpeg_make_d_derived_tbl, global memory only
    int32_t maxcode[18];                             // largest code of each length (monotone, see derive_search_table)
    int32_t valoff[18];                              // huffval index of the first code of each length minus that code
    uint8_t val[256];
};
__host__ __device__ inline uint32_t fast_entry(bool ac, uint32_t len, uint32_t sym) {
    const uint32_t sz = sym & 15u, r = sym >> 4;
    const uint32_t adv = ac ? (sz ? r + 1u : (r == 15u ? 16u : 64u)) : 1u;
    return (sym << 24) | (len << 16) | (adv << 8) | (len + sz);
}
// A bit pattern no code starts with (corrupt data, or a speculative decode off the symbol grid): 16 bits are skipped as
// symbol 0, the length field says 32 (a bit no real length has) and the write pass reports it.
// The synchronisation rounds only track the decoder state, and at q85 a symbol is ~6 bits: their tables (`ptabs`) carry in
// bits 16-31 the skip and zigzag advance of this symbol AND the next one where the next code still lies inside the lookup
// window (both AC, same table; 31 % fewer table reads on 4K q85 files), else a copy of bits 0-15.  The pair applies when
// the first symbol neither ends the block nor the sub-sequence.
__host__ __device__ inline uint32_t pair_entry(uint32_t first, uint32_t both) { return (first & 0xffffu) | (both << 16); }
__host__ __device__ inline uint32_t invalid_entry(bool ac) { return (32u << 16) | ((ac ? 64u : 1u) << 8) | 16u; }


constexpr uint32_t kMaxBlocksInMcu = 10;        // libjpeg's D_MAX_BLOCKS_IN_MCU: files with more are rejected by the parser
struct EntropyGeom {
    uint32_t ncomp, blocks_per_mcu, mcus_w, mcus_h;
    uint32_t bw[3], bh[3];
    uint8_t kcomp[kMaxBlocksInMcu], kdx[kMaxBlocksInMcu], kdy[kMaxBlocksInMcu];   // block k of an MCU: component and offset inside the MCU
    uint32_t kcomp_packed;                           // kcomp as 2-bit fields: the per-symbol lookup is a shift, not a load
    uint32_t hs[3], vs[3];
};

struct Segment {                                     // an independently decodable run: (image, restart interval)
    uint32_t first_sub, n_sub;
    uint32_t bit_end;                                // absolute bit index one past the segment's data
    uint32_t n_blocks;                               // blocks it must produce
    uint32_t image, first_mcu;
};

struct EntropyArgs {
    EntropyGeom g;
    const uint32_t* words;                           // un-stuffed scan data, big-endian words, segments 1024-bit aligned
    const Segment* segs;
    const uint32_t* sub_seg;                         // segment of every sub-sequence
    const uint32_t* wg_order;                        // synchronisation launch: workgroup blockIdx.x works on this block of kOwnSubs sub-sequences
    const FastTabs* ftabs;                           // [image]
    const FastTabs* ptabs;                           // [image] the same tables with pair entries, for the synchronisation rounds
    const FastTabs* ctabs;                           // [image] pair entries in the AC tables only, for the count pass (it reads the DC symbols)
    const SearchTab* stabs;                          // [image][comp][dc, ac]: the serial search, for sub-tables that did not fit
    uint32_t n_sub, n_seg;
    uint32_t uniform_tables;                         // every image carries the same Huffman tables (e.g. the standard ones)
    uint32_t* exit_p[2];                             // exit bit position, double buffered by round parity
    uint32_t* exit_cz[2];                            // exit (block-in-MCU << 8) | zigzag index
    uint32_t* start_p;                               // start state the lane last decoded from
    uint32_t* start_cz;
    int4* cnt;                                       // per sub-sequence {blocks started, DC sum comp 0, 1, 2}
    int4* tail;                                      // per chunk of kChunkSubs sub-sequences: sum of cnt over the part that belongs to the
                                                     // segment of the chunk's last sub-sequence (what a segment carries into the next chunk)
    uint32_t* changed;                               // [16] lanes whose exit state moved in round r, at [r & 15]
    uint32_t* errors;                                // [16] bit 0 invalid code, 2 run past 63, 3 short segment
    uint32_t* unsettled;                             // [17] sub-sequences the count pass found decoded from another state than their predecessor's exit
    int16_t* coef[3];                                // [image][bh][bw][64]
    uint32_t round;
    uint32_t inner_rounds;                           // fixpoint iterations inside a workgroup per launch (kInnerRounds; tests lower it)
};

__constant__ uint8_t kZigzag[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                    41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                    15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

// ---- per-workgroup staging and the bit reader ---------------------------------------------------------------------
// A workgroup owns kWg consecutive sub-sequences of contiguous stream, copied into LDS with coalesced loads.  The copy
// is COLUMN-MAJOR with pitch 33 (word j of sub-sequence t at t * 33 + j: lanes walking their own sub-sequences at the
// same depth hit 32 different banks); bit positions are relative to the workgroup's first bit.
constexpr uint32_t kMarginSubs = 3;                  // > the longest possible block (64 symbols x 31 bits) + lookahead
constexpr uint32_t kColPitch = kSubWords + 1u;

template <uint32_t kWg, uint32_t kColsT>
__device__ __forceinline__ void stage_stream_columns(const EntropyArgs& a, uint32_t* lds_words, uint32_t first_sub) {
    const uint32_t word0 = first_sub * kSubWords;
    const uint32_t total = (a.n_sub + 2u) * kSubWords;                 // the buffer carries 64 slack words
    constexpr uint32_t kPer = (kColsT * kSubWords + kWg - 1u) / kWg;
    uint32_t v[kPer];
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {                              // all loads in flight before the first LDS store
        const uint32_t w = word0 + threadIdx.x + i * kWg;
        v[i] = w < total ? a.words[w] : 0u;
    }
#pragma unroll
    for (uint32_t i = 0; i < kPer; ++i) {
        const uint32_t r = threadIdx.x + i * kWg;
        if (r < kColsT * kSubWords) lds_words[r + r / kSubWords] = v[i];
    }
}
__device__ __forceinline__ uint32_t stream_word(const uint32_t* lds_words, uint32_t i) { return lds_words[i + i / kSubWords]; }

// The stream behind a bit position, in registers: the two words the 32-bit window straddles, the word after them
// (requested one refill early, so the only LDS read on a symbol's dependent chain is the table entry) and the number
// of unread bits left in the first word.  The window is one v_alignbit_b32; moving on is branch-free -- with 64 lanes
// some lane refills in every iteration anyway, and a branch costs more than the four selects.
struct Reader {
    uint32_t w0, w1, next;
    uint32_t left;                                   // unread bits of w0: 0..31 (0: the window is w1)
    uint32_t at;                                     // word index of `next`
    __device__ __forceinline__ void open(const uint32_t* lds_words, uint32_t p) {
        const uint32_t i1 = (p + 31u) >> 5;
        w0 = stream_word(lds_words, i1 ? i1 - 1u : 0u);
        w1 = stream_word(lds_words, i1);
        at = i1 + 1u;
        next = stream_word(lds_words, at);
        left = (0u - p) & 31u;
    }
    __device__ __forceinline__ uint32_t peek() const { return __builtin_amdgcn_alignbit(w0, w1, left); }
    __device__ __forceinline__ void skip(const uint32_t* lds_words, uint32_t n) {       // n <= 31
        const int32_t rest = static_cast<int32_t>(left) - static_cast<int32_t>(n);
        const bool over = rest < 0;
        w0 = over ? w1 : w0;
        w1 = over ? next : w1;
        at += over ? 1u : 0u;
        left = static_cast<uint32_t>(rest) & 31u;
        next = stream_word(lds_words, at);
    }
};

// Entry of a symbol whose code is longer than the first-level lookup (e = that lookup's entry, skip field 0).
// kPairs: 0 plain entries, 1 pair entries everywhere (ptabs), 2 pair entries in the AC tables only (ctabs)
template <int kPairs = 0, typename Tabs>
__device__ __forceinline__ uint32_t long_entry(const Tabs* T, const SearchTab* S, uint32_t slot, uint32_t e, uint32_t bits) {
    if (e != 0u) return T->pool[(e >> 16) + ((bits << kLutBits) >> ((e >> 8) & 255u))];
    // jdhuff.c's slow path "l = min{l : code_l <= maxcode[l]}" without its dependent chain: all candidate lengths are
    // compared at once (the host stores a monotone maxcode, see derive_search_table)
    const SearchTab* t = S + slot;
    const bool ac = (slot & 1u) != 0u;
    uint32_t l = kLutBits + 1u;
#pragma unroll
    for (uint32_t k = kLutBits + 1u; k <= 16u; ++k)
        l += static_cast<int32_t>(bits >> (32u - k)) > t->maxcode[k] ? 1u : 0u;
    uint32_t r = invalid_entry(ac);
    if (l <= 16u) {
        const int32_t code = static_cast<int32_t>(bits >> (32u - l));
        const uint32_t sym = t->val[(code + t->valoff[l]) & 255];
        if (ac || sym <= 11u) r = fast_entry(ac, l, sym);
    }
    return (kPairs == 1 || (kPairs == 2 && ac)) ? pair_entry(r, r) : r;
}

// ---- synchronisation and count passes: the lean walker ----------------------------------------------------------
// These passes need the decoder STATE only (bit position, block-in-MCU, zigzag index), not the coefficients.  A dense
// pass is bound by instruction issue, a re-decode iteration of the fixpoint by the dependent chain of one symbol; both
// shrink with the instruction count.  Per symbol: the window (1 instruction), the table entry (address + LDS read),
// the reader's refill and the end-of-block bookkeeping as selects -- no divergent branch but the long-code lookup.
// kCount: also count the blocks started and sum the DC differences per component (into the lane's LDS slots dcs[comp *
// kDcPitch]: a dynamic index into three registers costs eight selects).
template <bool kCount, uint32_t kDcPitch, int kPairs, typename Tabs>
__device__ __forceinline__ void walk(const EntropyGeom& g, const uint32_t* lds_words, const Tabs* T, const SearchTab* S, uint32_t end,
                                     uint32_t& p, uint32_t& c, uint32_t& z, int32_t& n, int32_t* dcs) {
    const auto* lut0 = &T->lut[0][0];
    uint32_t comp = (g.kcomp_packed >> (2u * c)) & 3u;
    const auto* tcur = lut0 + comp * (2u * kLutEntries) + (z ? kLutEntries : 0u);
    Reader rd;
    rd.open(lds_words, p);
    while (p < end) {
        const uint32_t bits = rd.peek();
        uint32_t e = tcur[bits >> (32u - kLutBits)];
        if ((e & 255u) == 0u) e = long_entry<kPairs>(T, S, static_cast<uint32_t>(tcur - lut0) / kLutEntries, e, bits);
        if constexpr (kPairs == 1) {                                 // this symbol and the next, if this one ends neither block nor walk
            const bool both = z + ((e >> 8) & 255u) < 64u && p + (e & 255u) < end;
            e = both ? e >> 16 : e;
        }
        if constexpr (kCount) if (z == 0u) {                                     // DC symbol: jdhuff.c HUFF_EXTEND of the sz bits behind the code
            const uint32_t len = (e >> 16) & 255u, sz = (e >> 24) & 15u;
            const uint32_t v = ((bits << len) >> 1) >> (31u - sz);
            const int32_t ext = static_cast<int32_t>(v) - ((static_cast<int32_t>(bits << len) < 0 || sz == 0u) ? 0 : static_cast<int32_t>((1u << sz) - 1u));
            ++n;
            __hip_atomic_fetch_add(dcs + comp * kDcPitch, ext, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // ds_add, nothing returns
        }
        if constexpr (kPairs == 2) {                                 // (the DC entries of these tables are plain: the count pass reads them)
            const bool both = z != 0u && z + ((e >> 8) & 255u) < 64u && p + (e & 255u) < end;
            e = both ? e >> 16 : e;
        }
        const uint32_t skip = e & 255u;
        p += skip;
        rd.skip(lds_words, skip);
        z += (e >> 8) & 255u;
        const bool done = z >= 64u;                                  // block complete: next block of the MCU
        z = done ? 0u : z;
        const uint32_t c1 = c + 1u == g.blocks_per_mcu ? 0u : c + 1u;
        c = done ? c1 : c;
        comp = (g.kcomp_packed >> (2u * c)) & 3u;
        tcur = lut0 + comp * (2u * kLutEntries) + (done ? 0u : kLutEntries);
    }
}

// The same walk by a whole WAVE for one sub-sequence: lane d looks up the symbol that would start d bits behind the
// current position (all 64 offsets at once: one LDS round trip), then the chain "symbol at 0 -> skip -> symbol at skip
// -> ..." is resolved with scalar instructions and readlane -- no memory access on the dependent chain -- until the block
// ends (the tables change), the 64 looked-up offsets are used up or the sub-sequence ends.  About one block per window.
// A lone lane needs ~400 cycles per symbol (a wave issues one instruction per 4 cycles and every symbol waits for its
// table entry); the late iterations of the fixpoint, where a few sub-sequences carry a correction forward one step per
// iteration, are nothing but such lone walks.  p, c, z and end are wave-uniform.
__device__ __forceinline__ uint32_t uniform(uint32_t v) { return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(v))); }

template <typename Tabs>
__device__ __forceinline__ void walk_wave(const EntropyGeom& g, const uint32_t* lds_words, const Tabs* T, const SearchTab* S, uint32_t end_,
                                          uint32_t& p_, uint32_t& c_, uint32_t& z_) {
    const uint32_t lane = threadIdx.x & 63u;
    uint32_t p = uniform(p_), c = uniform(c_), z = uniform(z_);     // scalar registers: the chain below is scalar code
    const uint32_t end = uniform(end_), bpm = uniform(g.blocks_per_mcu), kc = uniform(g.kcomp_packed);
    while (p < end) {
        const uint32_t slot_ac = ((kc >> (2u * c)) & 3u) * 2u + 1u;
        const uint32_t q = p + lane, w = q >> 5;
        const uint32_t bits = __builtin_amdgcn_alignbit(stream_word(lds_words, w), stream_word(lds_words, w + 1u), (0u - q) & 31u);
        // (alignbit by 0 returns its second word: q & 31 == 0 needs the first)
        const uint32_t win = (q & 31u) ? bits : stream_word(lds_words, w);
        const uint32_t slot = (lane == 0u && z == 0u) ? slot_ac - 1u : slot_ac;       // only the symbol at the current position can be a DC symbol
        uint32_t e = T->lut[slot][win >> (32u - kLutBits)];
        // The chain, in scalar registers: entry of the symbol at `pos`, advance pos and z, until the block ends (z >= 64),
        // the window is used up (pos >= lim) or an entry has no skip (a code longer than the lookup ON the chain, 2 % of
        // the symbols: the long codes of the window are resolved then, once, and the chain goes on).  Hand-written because
        // the loop is the critical path of the late iterations: ten instructions and one taken branch per symbol.
        const uint32_t lim = min(64u, end - p);
        uint32_t pos = 0u, es, tmp;
        for (;;) {
            asm volatile(
                "1:\n\t"
                "v_readlane_b32 %[es], %[e], %[pos]\n\t"
                "s_and_b32 %[tmp], %[es], 0xff\n\t"
                "s_cbranch_scc0 2f\n\t"
                "s_add_u32 %[pos], %[pos], %[tmp]\n\t"
                "s_bfe_u32 %[tmp], %[es], 0x80008\n\t"
                "s_add_u32 %[z], %[z], %[tmp]\n\t"
                "s_cmp_ge_u32 %[z], 64\n\t"
                "s_cbranch_scc1 2f\n\t"
                "s_cmp_lt_u32 %[pos], %[lim]\n\t"
                "s_cbranch_scc1 1b\n\t"
                "2:"
                : [pos] "+s"(pos), [z] "+s"(z), [es] "=&s"(es), [tmp] "=&s"(tmp)
                : [e] "v"(e), [lim] "s"(lim)
                : "scc");
            if ((es & 255u) != 0u) break;
            if ((e & 255u) == 0u) e = long_entry<1>(T, S, slot, e, win);         // (only bits 0-15 of an entry are used here)
        }
        if (z >= 64u) {                                              // block complete: the lanes behind looked up the old tables
            z = 0u;
            c = c + 1u == bpm ? 0u : c + 1u;
        }
        p += pos;
    }
    p_ = p; c_ = c; z_ = z;
}

template <typename F>
__device__ __forceinline__ void with_fast_tables(const EntropyArgs& a, const FastTabs* gtabs, const FastTabs* lds_tabs, uint32_t lds_image,
                                                 uint32_t image, F&& f) {
    const SearchTab* S = a.stabs + static_cast<size_t>(image) * 6u;
    if (image == lds_image || a.uniform_tables) f(reinterpret_cast<const __attribute__((address_space(3))) FastTabs*>(
                                  static_cast<uint32_t>(reinterpret_cast<uintptr_t>(lds_tabs))), S);
    else f(gtabs + image, S);
}
template <uint32_t kWg>
__device__ __forceinline__ void stage_fast_tables(const FastTabs* gtabs, FastTabs* lds_tabs, uint32_t image) {
    const uint32_t* src = reinterpret_cast<const uint32_t*>(gtabs + image);
    uint32_t* dst = reinterpret_cast<uint32_t*>(lds_tabs);
    for (uint32_t i = threadIdx.x; i < sizeof(FastTabs) / 4u; i += kWg) dst[i] = src[i];
}


constexpr uint32_t kSyncLanes = 1024;               // sub-sequences per workgroup in these passes
constexpr uint32_t kSyncCols = kSyncLanes + 1u;                     // a walk stops within 31 bits of its end and looks 3 words ahead
constexpr uint32_t kFastStageDwords = kSyncCols * kColPitch;

// One synchronisation launch.  Inside the workgroup the fixpoint iteration runs in LDS: in every iteration the
// sub-sequences whose entry state (= the predecessor's exit) moved are decoded again, until nothing moves or
// kInnerRounds is reached; the host only iterates for corrections that cross workgroup boundaries.
// After the first two iterations few sub-sequences need another decode, but they are scattered -- a wave with one busy
// lane costs as much as a full one -- so the iteration COMPACTS them: sub-sequence states live in LDS (one packed word:
// relative bit position | block-in-MCU << 21 | zigzag index << 25), the lanes that need a decode enter a work list
// through a ballot / prefix count, and lane k decodes the k-th entry.  An iteration then costs what its dense waves cost.
// (No cap below the workgroup's own length: the loop ends as soon as nothing is pending, and a chain of corrections that
// is cut short costs a host round trip, more launches and a REPEATED count + write pass.  With a cap of 24 that was half of
// all 64-file decodes of BASELINE cfg4's files -- noise files re-synchronise with p ~ 0.4 per sub-sequence, 430 000
// sub-sequences: the longest chain of a batch is ~ ln 430 000 / ln (1 / 0.6) = 25 -- round 6, profiles/r6_entropy_rounds.txt.)
constexpr uint32_t kInnerRounds = 1024;
constexpr uint32_t kWaveWalkMax = 16;                          // at most this many pending sub-sequences: one wave per walk
// The first kWarmLanes lanes of a workgroup decode the sub-sequences IN FRONT of its own range in round 0 and discard the
// result: the workgroup's first own sub-sequence then starts from a state that has had kWarmLanes sub-sequences to
// synchronise (one fails to with probability ~0.4), so corrections across workgroup boundaries -- a whole extra launch
// of lone walks -- all but disappear.  Later rounds run without them.
constexpr uint32_t kWarmLanes = 16;
constexpr uint32_t kOwnSubs = kSyncLanes - kWarmLanes;                         // sub-sequences a workgroup owns
static_assert(kWarmLanes < kSyncLanes && kSyncLanes % 64u == 0u && kSyncLanes <= 1024u, "whole waves, one workgroup");
constexpr uint32_t kNever = 0xffffffffu;
constexpr uint32_t kFlagWords = 18;                                             // changed[16], errors, unsettled
constexpr uint32_t kChunkSubs = 512;                                            // sub-sequences per workgroup of the write pass                                       // no decode yet (no real state packs to this)
static_assert((kSyncLanes + 2u) * kSubBits < (1u << 21), "packed state: 21 bits of relative position");

__device__ __forceinline__ uint32_t pack_state(uint32_t p_rel, uint32_t cz) { return p_rel | ((cz >> 8) << 21) | ((cz & 255u) << 25); }

__global__ void __launch_bounds__(kSyncLanes) entropy_round_kernel(const EntropyArgs a) {
    __shared__ uint32_t lds_words[kFastStageDwords];
    __shared__ FastTabs lds_tabs;
    __shared__ uint2 st[kSyncLanes];                                 // .x exit state, .y the entry state it was decoded from (written as a pair)
    __shared__ uint16_t endinfo[kSyncLanes], work[kSyncLanes];       // end - t * 1024 (| kChase); sub-sequences to decode this iteration
    __shared__ uint32_t wave_cnt[kSyncLanes / 64u];
    // Workgroups are dispatched in blockIdx order and the launch lasts until its slowest one is done.  Their times differ by
    // what the fixpoint needs -- typically 3 iterations on a smooth image and 8 on a noisy one, ~19 us each -- but the LONGEST
    // correction chains of a batch (24 - 50 iterations on BASELINE cfg4's batches) sit in the SMOOTH images: where every block
    // looks like its neighbour a decoder that is bit-synchronous but one block out of phase in the MCU stays out of phase,
    // while noise throws it out and lets it re-synchronise.  One such chain that starts in the last dispatch round sets the
    // launch's length (measured: 1 425 us with a 50-iteration workgroup dispatched at 506 us, where perfect packing is 575).  So
    // the blocks of the images with the SMALLEST scans are dispatched first: their rare long chains run beside everybody else
    // (the same two batches: 848 -> 724 and 1 425 -> 946 us per launch; largest-first made it 1 016 and 1 409;
    // profiles/r6_entropy_dispatch_order.txt).  The true state crosses such a region one sub-sequence per iteration while
    // everything behind the front is decoded again from entries that are still wrong, so these iterations are lane walks of many
    // sub-sequences, not lone wave walks: letting a lone chain's wave follow it link by link (measured, round 6) met 3 of 50.
    const uint32_t own_sub = a.wg_order[blockIdx.x] * kOwnSubs, first_sub = own_sub - kWarmLanes;     // (wraps for block 0: those lanes are off)
    const uint32_t t = threadIdx.x, s = first_sub + t, lane = t & 63u, wave = t >> 6;
    if (a.round == 0u && blockIdx.x == 0u && t < kFlagWords) a.changed[t] = 0u;     // the flags of this decode (later launches set them)
    const uint32_t t0 = a.round == 0u ? 0u : kWarmLanes;             // first lane at work
    const bool on = s < a.n_sub && t >= t0;
    const uint32_t cur = a.round & 1u, prv = cur ^ 1u;
    const uint32_t bit0 = first_sub * kSubBits;
    const Segment sg = a.segs[a.sub_seg[on ? s : own_sub]];
    const bool first = on && s == sg.first_sub;
    // state carried over from the previous launch (none in round 0)
    uint32_t st_ex = 0u, st_used = kNever;
    if (on && a.round > 0u) {
        st_ex = pack_state(a.exit_p[prv][s] - bit0, a.exit_cz[prv][s]);
        st_used = a.start_p[s] == kNever ? kNever : pack_state(a.start_p[s] - bit0, a.start_cz[s]);
    }
    const uint32_t old_ex = st_ex;
    // entry state of lane 0 of the workgroup comes from the previous workgroup's last launch; round 0 starts every
    // sub-sequence speculatively at its own first bit
    uint32_t fixed_entry = pack_state(t * kSubBits, 0u);
    const bool fixed = first || a.round == 0u || t == t0;
    if (on && !first && a.round > 0u && t == t0) fixed_entry = pack_state(a.exit_p[prv][s - 1u] - bit0, a.exit_cz[prv][s - 1u]);
    if (a.round > 0u) {
        const uint32_t entry = (fixed || !on) ? fixed_entry : pack_state(a.exit_p[prv][s - 1u] - bit0, a.exit_cz[prv][s - 1u]);
        if (!__syncthreads_or(on && entry != st_used ? 1 : 0)) {     // nothing moved in front of this workgroup
            if (on) { a.exit_p[cur][s] = a.exit_p[prv][s]; a.exit_cz[cur][s] = a.exit_cz[prv][s]; }
            return;
        }
    }
    const uint32_t wg_image = a.segs[a.sub_seg[own_sub]].image;
    stage_stream_columns<kSyncLanes, kSyncCols>(a, lds_words, first_sub);
    stage_fast_tables<kSyncLanes>(a.ptabs, &lds_tabs, wg_image);
    st[t] = make_uint2(st_ex, st_used);
    constexpr uint32_t kChase = 0x8000u;                             // the entry is the predecessor's exit (not a segment's first sub-sequence)
    endinfo[t] = static_cast<uint16_t>(on ? (min((s + 1u) * kSubBits, sg.bit_end) - s * kSubBits) | (first ? 0u : kChase) : 0u);
    __syncthreads();
    bool pending = false;
    for (uint32_t it = 0; it < a.inner_rounds; ++it) {
        // speculative first decode of round 0: from the sub-sequence's own first bit; afterwards from the predecessor's exit
        const uint32_t entry = (fixed && !(a.round == 0u && it > 0u && !first && t > 0u)) ? fixed_entry : st[t ? t - 1u : 0u].x;
        const bool need = on && entry != st[t].y;
        const uint64_t vote = __ballot(need);
        if (lane == 0u) wave_cnt[wave] = static_cast<uint32_t>(__popcll(vote));
        __syncthreads();
        uint32_t base = 0u, total = 0u;
#pragma unroll
        for (uint32_t w = 0; w < kSyncLanes / 64u; ++w) { const uint32_t n = wave_cnt[w]; base += w < wave ? n : 0u; total += n; }
        pending = total != 0u;
        if (!pending) break;
        if (need) {
            st[t].y = entry;
            work[base + static_cast<uint32_t>(__popcll(vote & ((1ull << lane) - 1ull)))] = static_cast<uint16_t>(t);
        }
        __syncthreads();
        if (total <= kWaveWalkMax) {                                 // few sub-sequences: one wave each, see walk_wave
            // (Letting the wave FOLLOW a correction into the next sub-sequence instead of leaving it to the next iteration
            // was measured: 459 -> 576 us per launch -- a chain followed from an exit that is itself corrected later is
            // walked twice, and the other work of the wave waits.)
            for (uint32_t k = wave; k < total; k += kSyncLanes / 64u) {
                const uint32_t j = uniform(work[k]);
                const uint32_t e0 = uniform(st[j].y);
                uint32_t p = e0 & 0x1fffffu, c = (e0 >> 21) & 15u, z = e0 >> 25;
                const uint32_t end = j * kSubBits + (endinfo[j] & (kChase - 1u));
                const uint32_t image = a.uniform_tables ? wg_image : a.segs[a.sub_seg[first_sub + j]].image;
                with_fast_tables(a, a.ptabs, &lds_tabs, wg_image, image, [&](auto tabs, const SearchTab* S) { walk_wave(a.g, lds_words, tabs, S, end, p, c, z); });
                if (lane == 0u) st[j].x = p | (c << 21) | (z << 25);
            }
        } else if (t < total) {
            const uint32_t j = work[t];
            const uint32_t e0 = st[j].y;
            uint32_t p = e0 & 0x1fffffu, c = (e0 >> 21) & 15u, z = e0 >> 25;
            const uint32_t end = j * kSubBits + (endinfo[j] & (kChase - 1u));
            int32_t n = 0;
            const uint32_t image = a.uniform_tables ? wg_image : a.segs[a.sub_seg[first_sub + j]].image;
            with_fast_tables(a, a.ptabs, &lds_tabs, wg_image, image, [&](auto tabs, const SearchTab* S) { walk<false, 0u, 1>(a.g, lds_words, tabs, S, end, p, c, z, n, nullptr); });
            st[j].x = p | (c << 21) | (z << 25);
        }
        __syncthreads();
    }
    if (!on || t < kWarmLanes) return;
    const uint32_t fin = st[t].x, fu = st[t].y;
    a.exit_p[cur][s] = (fin & 0x1fffffu) + bit0;
    a.exit_cz[cur][s] = (((fin >> 21) & 15u) << 8) | (fin >> 25);
    a.start_p[s] = fu == kNever ? kNever : (fu & 0x1fffffu) + bit0;
    a.start_cz[s] = (((fu >> 21) & 15u) << 8) | (fu >> 25);
    // another launch is needed if this lane's exit moved (its successor may sit in the next workgroup) or the inner
    // iteration was cut short
    if (a.round > 0u && (fin != old_ex || pending)) atomicAdd(a.changed + (a.round & 15u), 1u);
}

// Count pass, once the exit states are final: every lane walks its sub-sequence from its true entry state and records
// {blocks started, DC difference sum per component}; the write pass turns them into every lane's first block and DC
// predictors (an exclusive scan over the sub-sequences of a segment).  To spare that scan a launch of its own, the
// count pass also leaves, per chunk of kChunkSubs sub-sequences, what a segment carries out of the chunk (`tail`): a
// write workgroup adds the tails of the chunks between its segment's start and itself (a 4K file: at most 26) and
// scans its own 512 counts.  The pass also CHECKS the fixpoint: a sub-sequence that was last decoded from another state
// than its predecessor's exit (a correction that crossed a workgroup boundary of round 0, or an iteration limit) is
// counted in `unsettled`, and the host runs further rounds before it trusts count and write pass.
__device__ __forceinline__ int4 add4(int4 a, int4 b) { return make_int4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ int4 sub4(int4 a, int4 b) { return make_int4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
__device__ __forceinline__ int4 shfl_up4(int4 v, uint32_t d) {
    return make_int4(__shfl_up(v.x, d, 64), __shfl_up(v.y, d, 64), __shfl_up(v.z, d, 64), __shfl_up(v.w, d, 64));
}
__device__ __forceinline__ int4 shfl_down4(int4 v, uint32_t d) {
    return make_int4(__shfl_down(v.x, d, 64), __shfl_down(v.y, d, 64), __shfl_down(v.z, d, 64), __shfl_down(v.w, d, 64));
}
__device__ __forceinline__ int4 wave_sum4(int4 v) {                 // lane 0 gets the sum over the wave
#pragma unroll
    for (uint32_t d = 32u; d > 0u; d >>= 1) v = add4(v, shfl_down4(v, d));
    return v;
}
static_assert(kSyncLanes % kChunkSubs == 0u, "a count workgroup covers whole chunks");

__global__ void __launch_bounds__(kSyncLanes) entropy_count_kernel(const EntropyArgs a) {
    __shared__ uint32_t lds_words[kFastStageDwords];
    __shared__ FastTabs lds_tabs;
    __shared__ int32_t lds_dc[3u * kSyncLanes];
    __shared__ int32_t lds_tail[kSyncLanes / kChunkSubs][4];
    const uint32_t first_sub = blockIdx.x * kSyncLanes;
    const uint32_t s = first_sub + threadIdx.x;
    const uint32_t wg_image = a.segs[a.sub_seg[first_sub]].image;
    stage_stream_columns<kSyncLanes, kSyncCols>(a, lds_words, first_sub);
    constexpr int kCountPairs = 2;
    const FastTabs* gtabs = kCountPairs ? a.ctabs : a.ftabs;
    stage_fast_tables<kSyncLanes>(gtabs, &lds_tabs, wg_image);
    int32_t* dcs = lds_dc + threadIdx.x;                     // this lane's three sums, kSyncLanes apart (one bank per lane)
    dcs[0] = 0; dcs[kSyncLanes] = 0; dcs[2u * kSyncLanes] = 0;
    if (threadIdx.x < (kSyncLanes / kChunkSubs) * 4u) (&lds_tail[0][0])[threadIdx.x] = 0;
    __syncthreads();
    const bool on = s < a.n_sub;
    int4 mine = make_int4(0, 0, 0, 0);
    uint32_t my_seg = 0xffffffffu;
    if (on) {
        my_seg = a.sub_seg[s];
        const Segment sg = a.segs[my_seg];
        const uint32_t fin = a.round & 1u;
        const uint32_t bit0 = first_sub * kSubBits;
        uint32_t p = s * kSubBits - bit0, c = 0, z = 0;
        if (s != sg.first_sub) {
            const uint32_t ep = a.exit_p[fin][s - 1u], ecz = a.exit_cz[fin][s - 1u];
            if (a.start_p[s] != ep || a.start_cz[s] != ecz) atomicAdd(a.unsettled, 1u);
            p = ep - bit0; c = ecz >> 8; z = ecz & 255u;
        }
        const uint32_t end = min((s + 1u) * kSubBits, sg.bit_end) - bit0;
        int32_t n = 0;
        with_fast_tables(a, gtabs, &lds_tabs, wg_image, sg.image, [&](auto tabs, const SearchTab* S) { walk<true, kSyncLanes, kCountPairs>(a.g, lds_words, tabs, S, end, p, c, z, n, dcs); });
        mine = make_int4(n, dcs[0], dcs[kSyncLanes], dcs[2u * kSyncLanes]);
        a.cnt[s] = mine;
    }
    // what the segment of each chunk's last sub-sequence carries out of the chunk
    const uint32_t half = threadIdx.x / kChunkSubs;
    const uint32_t chunk_first = first_sub + half * kChunkSubs;
    const bool chunk_on = chunk_first < a.n_sub;
    const uint32_t last_seg = chunk_on ? a.sub_seg[min(chunk_first + kChunkSubs, a.n_sub) - 1u] : 0xfffffffeu;
    const int4 part = wave_sum4(my_seg == last_seg ? mine : make_int4(0, 0, 0, 0));
    if ((threadIdx.x & 63u) == 0u) {
        atomicAdd(&lds_tail[half][0], part.x); atomicAdd(&lds_tail[half][1], part.y);
        atomicAdd(&lds_tail[half][2], part.z); atomicAdd(&lds_tail[half][3], part.w);
    }
    __syncthreads();
    if (threadIdx.x < kSyncLanes / kChunkSubs && first_sub + threadIdx.x * kChunkSubs < a.n_sub)
        a.tail[blockIdx.x * (kSyncLanes / kChunkSubs) + threadIdx.x] =
            make_int4(lds_tail[threadIdx.x][0], lds_tail[threadIdx.x][1], lds_tail[threadIdx.x][2], lds_tail[threadIdx.x][3]);
}

// Write pass.  A block belongs to the lane in whose sub-sequence it STARTS: that lane decodes it to the end (running
// past its sub-sequence if need be), assembles the 64 coefficients in a private LDS row (pitch 36 dwords: 16-byte
// aligned, eight consecutive lanes cover all banks) and stores the whole block with eight 16-byte stores -- every block
// is written exactly once, zeros included, so the planes need no clearing and no two lanes ever touch the same block.
// A lane that starts inside a block skips to its end without storing.  Same reader and table entries as the walker;
// workgroups of 512 sub-sequences (stream 66 KiB + rows 72 KiB + tables: one workgroup per CU).
// Storing a block is ~40 instructions that every lane of the wave pays for whenever ONE lane needs them, and with 64
// lanes a block ends somewhere in almost every symbol step; waiting for all lanes to finish their block (the loop
// structure a compiler makes of "decode a block, store it") leaves most lanes idle, blocks differ that much in length.
// So a lane that completed its block WAITS until kFlushLanes lanes of its wave wait (or nobody decodes any more) and
// they store together: the store path runs every third or fourth symbol step and a lane idles ~2 steps per block.
constexpr uint32_t kWriteLanes = kChunkSubs;
constexpr uint32_t kWriteCols = kWriteLanes + kMarginSubs;
constexpr uint32_t kBlkPitch = 36;                   // dwords per lane row (32 + 4: int16 slots 64..71 are padding)

constexpr uint32_t kFlushLanes = 16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct BlockPlace { uint32_t hv, bw, bh, comp; };    // block k of an MCU: hs | vs << 8 | dx << 16 | dy << 24, plane dimensions in blocks

__global__ void __launch_bounds__(kWriteLanes) entropy_write_kernel(const EntropyArgs a) {
    __shared__ uint32_t lds_words[kWriteCols * kColPitch];
    __shared__ FastTabs lds_tabs;
    __shared__ __attribute__((aligned(16))) uint32_t lds_blk[kWriteLanes * kBlkPitch];
    __shared__ BlockPlace lds_place[kMaxBlocksInMcu];
    __shared__ int16_t* lds_plane[kMaxBlocksInMcu];  // coefficient plane of block k's component
    __shared__ uint8_t lds_zz[64];                   // zigzag -> natural order (a divergent index into __constant__ memory is a
    if (threadIdx.x < 64u) lds_zz[threadIdx.x] = kZigzag[threadIdx.x];      // vector-memory load per coefficient)
    if (threadIdx.x < a.g.blocks_per_mcu) {
        const uint32_t k = threadIdx.x, cm = a.g.kcomp[k];
        lds_place[k] = BlockPlace{a.g.hs[cm] | (a.g.vs[cm] << 8) | (static_cast<uint32_t>(a.g.kdx[k]) << 16) | (static_cast<uint32_t>(a.g.kdy[k]) << 24),
                                  a.g.bw[cm], a.g.bh[cm], cm};
        lds_plane[k] = a.coef[cm];
    }
    const uint32_t first_sub = blockIdx.x * kWriteLanes;
    const uint32_t s = first_sub + threadIdx.x;
    // This lane's first block and DC predictors: the exclusive scan of the counts over the sub-sequences of its segment
    // = (what the segment carried through the chunks in front of this workgroup) + (the scan inside the workgroup).
    int4 pre;
    {
        __shared__ int4 wave_tot[kWriteLanes / 64u], lead;
        int4* excl_at = reinterpret_cast<int4*>(lds_blk);            // (the rows are not in use yet)
        const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
        const bool on = s < a.n_sub;
        const Segment sg0 = a.segs[a.sub_seg[first_sub]], sgm = a.segs[a.sub_seg[on ? s : first_sub]];
        const int4 mine = on ? a.cnt[s] : make_int4(0, 0, 0, 0);
        int4 inc = mine;                                             // inclusive scan inside the wave
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const int4 o = shfl_up4(inc, d);
            if (lane >= d) inc = add4(inc, o);
        }
        if (lane == 63u) wave_tot[wave] = inc;
        if (wave == 0u) {                                            // the leading segment's counts in front of this chunk
            int4 v = make_int4(0, 0, 0, 0);
            if (sg0.first_sub < first_sub)
                for (uint32_t i = sg0.first_sub / kChunkSubs + lane; i < blockIdx.x; i += 64u) v = add4(v, a.tail[i]);
            v = wave_sum4(v);
            if (lane == 0u) lead = v;
        }
        __syncthreads();
        int4 excl = sub4(inc, mine);
        for (uint32_t w = 0; w < wave; ++w) excl = add4(excl, wave_tot[w]);
        excl_at[threadIdx.x] = excl;
        __syncthreads();
        const bool from_before = sgm.first_sub < first_sub;          // the lane's segment began in front of this workgroup
        pre = sub4(excl, excl_at[from_before ? 0u : sgm.first_sub - first_sub]);
        if (from_before) pre = add4(pre, lead);
        __syncthreads();                                             // every lane has read: the rows may be cleared
    }
    int16_t* row = reinterpret_cast<int16_t*>(lds_blk + threadIdx.x * kBlkPitch);
    u32x4* row4 = reinterpret_cast<u32x4*>(lds_blk + threadIdx.x * kBlkPitch);
#pragma unroll
    for (uint32_t i = 0; i < 8u; ++i) row4[i] = u32x4{0u, 0u, 0u, 0u};               // a row is cleared again when it is stored
    const uint32_t wg_image = a.segs[a.sub_seg[first_sub]].image;
    stage_stream_columns<kWriteLanes, kWriteCols>(a, lds_words, first_sub);
    stage_fast_tables<kWriteLanes>(a.ftabs, &lds_tabs, wg_image);
    __syncthreads();
    if (s >= a.n_sub) return;
    const Segment sg = a.segs[a.sub_seg[s]];
    const uint32_t fin = a.round & 1u;                       // parity of the last round run
    const uint32_t bit0 = first_sub * kSubBits;
    uint32_t p = s * kSubBits - bit0, c = 0u, z = 0u;
    if (s != sg.first_sub) { p = a.exit_p[fin][s - 1u] - bit0; const uint32_t cz = a.exit_cz[fin][s - 1u]; c = cz >> 8; z = cz & 255u; }
    const uint32_t end = min((s + 1u) * kSubBits, sg.bit_end) - bit0;
    uint32_t err = 0;
    int32_t dc0 = pre.y, dc1 = pre.z, dc2 = pre.w;
    uint32_t block = static_cast<uint32_t>(pre.x);           // the next block this lane starts
    const uint32_t B = a.g.blocks_per_mcu;
    uint32_t my, mx;                                         // MCU of that block (the block-in-MCU is the decoder's c)
    { const uint32_t m = sg.first_mcu + block / B; my = m / a.g.mcus_w; mx = m - my * a.g.mcus_w; }
    with_fast_tables(a, a.ftabs, &lds_tabs, wg_image, sg.image, [&](auto T, const SearchTab* S) {
        const auto* lut0 = &T->lut[0][0];
        uint32_t comp = (a.g.kcomp_packed >> (2u * c)) & 3u;
        const auto* tcur = lut0 + comp * (2u * kLutEntries) + (z ? kLutEntries : 0u);
        Reader rd;
        rd.open(lds_words, p);
        // one symbol: entry and the 32 bits it was decoded from; the reader moves on, z / tables are the caller's
        auto symbol = [&](uint32_t& bits) {
            bits = rd.peek();
            uint32_t e = tcur[bits >> (32u - kLutBits)];
            if ((e & 255u) == 0u) e = long_entry(T, S, static_cast<uint32_t>(tcur - lut0) / kLutEntries, e, bits);
            const uint32_t skip = e & 255u;
            p += skip;
            rd.skip(lds_words, skip);
            return e;
        };
        auto advance = [&](uint32_t e) {                     // -> true when the block is complete
            z += (e >> 8) & 255u;
            const bool done = z >= 64u;
            z = done ? 0u : z;
            const uint32_t c1 = c + 1u == B ? 0u : c + 1u;
            c = done ? c1 : c;
            comp = (a.g.kcomp_packed >> (2u * c)) & 3u;
            tcur = lut0 + comp * (2u * kLutEntries) + (done ? 0u : kLutEntries);
            return done;
        };
        uint32_t bits;
        while (z != 0u) advance(symbol(bits));               // tail of the predecessor's block
        bool run = p < end && block < sg.n_blocks;           // a block to decode (pad bits may follow the last block)
        bool waiting = false;                                // block complete, not stored yet
        uint32_t k = c;                                      // block-in-MCU of the block being decoded
        // The DC predictors are touched when a block is stored, not per symbol; a coefficient goes into the row one symbol late: its natural-order index is an LDS lookup, and the wave would
        // sit out that round trip; this way it overlaps the next symbol's table read.
        constexpr uint32_t kNoStore = 64u;                   // a slot in the row's padding: "nothing to store" without a branch
        // (the looked-up index is not even LOOKED AT in the step that requests it -- a select on it would wait for the read --
        // but at the next step's row store: `pend_raw` is that read's destination, `pend_ok` whether it counts)
        uint32_t pend_raw = kNoStore;
        bool pend_ok = false;
        int32_t pend_val = 0;
        for (;;) {
            if (run && !waiting) {
                const bool is_dc = z == 0u;
                bits = rd.peek();
                uint32_t e = tcur[bits >> (32u - kLutBits)];
                row[pend_ok ? pend_raw : kNoStore] = static_cast<int16_t>(pend_val);
                if ((e & 255u) == 0u) e = long_entry(T, S, static_cast<uint32_t>(tcur - lut0) / kLutEntries, e, bits);
                p += e & 255u;
                rd.skip(lds_words, e & 255u);
                const uint32_t len = (e >> 16) & 255u, sz = (e >> 24) & 15u;
                const uint32_t v = ((bits << len) >> 1) >> (31u - sz);           // sz bits behind the code (sz = 0 -> 0)
                const int32_t neg = static_cast<int32_t>((1u << sz) - 1u);
                pend_val = static_cast<int32_t>(v) - ((static_cast<int32_t>(bits << len) < 0 || sz == 0u) ? 0 : neg);   // jdhuff.c HUFF_EXTEND
                const uint32_t pos = z + ((e >> 8) & 255u) - 1u;                // zigzag index of an AC coefficient (DC: 0)
                const bool over = sz != 0u && pos > 63u;                         // a coefficient behind the block's end
                err |= ((e >> 21) & 1u) | (over ? 4u : 0u);                      // (length field 32: no such code)
                pend_raw = lds_zz[pos & 63u];
                pend_ok = is_dc || (sz != 0u && !over);                          // a DC entry holds the DIFFERENCE until the block is stored
                waiting = advance(e);
            }
            const uint32_t n_wait = static_cast<uint32_t>(__popcll(__ballot(waiting)));
            const bool decoding = __ballot(run && !waiting) != 0ull;
            if (n_wait < kFlushLanes && decoding) continue;
            if (n_wait == 0u) break;                         // nobody decodes, nothing to store
            if (waiting) {
                row[pend_ok ? pend_raw : kNoStore] = static_cast<int16_t>(pend_val);
                pend_ok = false;
                // every LDS read of the store (the row, where the block goes) is requested before the first is used
                const BlockPlace pl = lds_place[k];
                int16_t* plane = lds_plane[k];
                u32x4 r[8];
#pragma unroll
                for (uint32_t i = 0; i < 8u; ++i) r[i] = row4[i];
                {                                            // DC: difference -> value (jdhuff.c last_dc_val)
                    const int32_t diff = static_cast<int16_t>(r[0].x & 0xffffu);
                    dc0 += pl.comp == 0u ? diff : 0; dc1 += pl.comp == 1u ? diff : 0; dc2 += pl.comp == 2u ? diff : 0;
                    const int32_t dcv = pl.comp == 0u ? dc0 : (pl.comp == 1u ? dc1 : dc2);
                    r[0].x = (r[0].x & 0xffff0000u) | (static_cast<uint32_t>(dcv) & 0xffffu);
                }
                const uint32_t bx = mx * (pl.hv & 255u) + ((pl.hv >> 16) & 255u), by = my * ((pl.hv >> 8) & 255u) + (pl.hv >> 24);
                auto* dst = reinterpret_cast<__attribute__((address_space(1))) u32x4*>(reinterpret_cast<uintptr_t>(       // (global, not flat, stores)
                    plane + (static_cast<size_t>(sg.image * pl.bh + by) * pl.bw + bx) * 64u));
                // The store is bounded by POSITION as well as by the segment's block count: this launch is enqueued behind the count
                // pass before the host has seen whether the fixpoint settled, and on exit states that are not final a lane's first
                // block and its block-in-MCU phase need not agree -- its MCU row can run past the frame (found by
                // tools/scan_entropy_shapes.py: a fault behind the last image's plane; the decode is repeated after the next
                // rounds, so the result was right, the transient stores were not).
                if (my < a.g.mcus_h) {
#pragma unroll
                    for (uint32_t i = 0; i < 8u; ++i) dst[i] = r[i];
                }
#pragma unroll
                for (uint32_t i = 0; i < 8u; ++i) row4[i] = u32x4{0u, 0u, 0u, 0u};
                ++block;
                if (c == 0u) { ++mx; if (mx == a.g.mcus_w) { mx = 0u; ++my; } }      // the MCU is complete
                k = c;
                waiting = false;
                run = p < end && block < sg.n_blocks;
            }
        }
    });
    const bool last = s + 1u == sg.first_sub + sg.n_sub;
    if (last && block < sg.n_blocks) err |= 8u;              // the segment ran out of data
    if (err) atomicOr(a.errors, err);
}

}  // namespace ifhip

using namespace ifhip;

// ==================================================================================================================
// host: parser, un-stuffing, tables
// ==================================================================================================================
namespace {

struct HuffSpec { uint8_t bits[17]; uint8_t vals[256]; bool present = false; };

struct ParsedJpeg {
    uint32_t width = 0, height = 0;
    int ncomp = 0;
    uint8_t comp_id[3] = {0, 0, 0}, hs[3] = {0, 0, 0}, vs[3] = {0, 0, 0}, tq[3] = {0, 0, 0}, td[3] = {0, 0, 0}, ta[3] = {0, 0, 0};
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    HuffSpec dc[4], ac[4];
    uint32_t restart_interval = 0;
    int adobe_transform = -1;                        // APP14 "Adobe" colour transform flag, -1: no such marker
    size_t scan_begin = 0;
    uint32_t mcus_w = 0, mcus_h = 0, bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0}, blocks_per_mcu = 0;
};

const uint8_t kZigzagHost[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                                 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                                 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

int parse_jpeg(const uint8_t* d, size_t len, ParsedJpeg* out) {
    ParsedJpeg& P = *out;
    if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: not a JPEG (no SOI)");
    size_t i = 2;
    bool have_sof = false;
    while (i + 4 <= len) {
        if (d[i] != 0xFF) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: marker expected at byte %zu", i);
        while (i < len && d[i] == 0xFF) ++i;                                   // fill bytes
        if (i >= len) break;
        const uint8_t m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;      // no payload
        if (m == 0xD9) break;
        if (i + 2 > len) break;
        const size_t seg = (static_cast<size_t>(d[i]) << 8) | d[i + 1];
        if (seg < 2 || i + seg > len) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: truncated marker segment FF%02X", m);
        const uint8_t* q = d + i + 2;
        const size_t n = seg - 2;
        if (m == 0xDB) {                                                       // DQT
            size_t k = 0;
            while (k < n) {
                const int pq = q[k] >> 4, t = q[k] & 15;
                ++k;
                if (t > 3 || k + (pq ? 128u : 64u) > n) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DQT");
                for (int z = 0; z < 64; ++z) {
                    const uint16_t v = pq ? static_cast<uint16_t>((q[k] << 8) | q[k + 1]) : q[k];
                    k += pq ? 2 : 1;
                    P.qt[t][kZigzagHost[z]] = v;
                }
                P.qt_present[t] = true;
            }
        } else if (m == 0xC4) {                                                // DHT
            size_t k = 0;
            while (k + 17 <= n) {
                const int tc = q[k] >> 4, th = q[k] & 15;
                if (tc > 1 || th > 3) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DHT");
                HuffSpec& h = tc ? P.ac[th] : P.dc[th];
                h.bits[0] = 0;
                size_t total = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = q[k + l]; total += q[k + l]; }
                k += 17;
                if (total > 256 || k + total > n) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DHT");
                std::memset(h.vals, 0, sizeof h.vals);
                std::memcpy(h.vals, q + k, total);
                k += total;
                h.present = true;
            }
        } else if (m == 0xC0 || m == 0xC1) {                                   // baseline / extended sequential, Huffman
            if (n < 6 || q[0] != 8) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-bit JPEG", n ? q[0] : 0);
            P.height = (q[1] << 8) | q[2];
            P.width = (q[3] << 8) | q[4];
            P.ncomp = q[5];
            if (P.ncomp != 1 && P.ncomp != 3) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-component JPEG", P.ncomp);
            if (n < 6u + 3u * P.ncomp) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOF");
            for (int c = 0; c < P.ncomp; ++c) {
                P.comp_id[c] = q[6 + 3 * c];
                P.hs[c] = q[7 + 3 * c] >> 4; P.vs[c] = q[7 + 3 * c] & 15; P.tq[c] = q[8 + 3 * c] & 3;
            }
            have_sof = true;
        } else if (m >= 0xC2 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: JPEG process SOF%d (only baseline Huffman)", m - 0xC0);
        } else if (m == 0xDD) {
            if (n >= 2) P.restart_interval = (q[0] << 8) | q[1];
        } else if (m == 0xEE && n >= 12 && std::memcmp(q, "Adobe", 5) == 0) {
            P.adobe_transform = q[11];                                         // 0: RGB / CMYK stored as is, 1: YCbCr, 2: YCCK
        } else if (m == 0xDA) {                                                // SOS
            if (!have_sof) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: SOS before SOF");
            if (n < 1 || q[0] != P.ncomp || n < 4u + 2u * P.ncomp)
                return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: non-interleaved / multi-scan JPEG");
            for (int s = 0; s < P.ncomp; ++s) {
                int c = -1;
                for (int k = 0; k < P.ncomp; ++k) if (P.comp_id[k] == q[1 + 2 * s]) c = k;
                if (c != s) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: scan component order");
                P.td[c] = q[2 + 2 * s] >> 4; P.ta[c] = q[2 + 2 * s] & 15;
                if (P.td[c] > 3 || P.ta[c] > 3) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOS");
            }
            P.scan_begin = i + seg;
            break;
        }
        i += seg;
    }
    if (!have_sof || !P.scan_begin || P.width == 0 || P.height == 0) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: no frame / scan found");
    // libjpeg's colour-space guess (jdapimin.c default_decompress_parms): the pixel stage converts YCbCr only
    if (P.ncomp == 3 && (P.adobe_transform == 0 ||
                         (P.adobe_transform < 0 && P.comp_id[0] == 'R' && P.comp_id[1] == 'G' && P.comp_id[2] == 'B')))
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: RGB-coded JPEG (no YCbCr transform)");
    uint32_t hmax = 1, vmax = 1;
    for (int c = 0; c < P.ncomp; ++c) {
        if (P.hs[c] < 1 || P.hs[c] > 2 || P.vs[c] < 1 || P.vs[c] > 2)
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: sampling factor %dx%d", P.hs[c], P.vs[c]);
        if (!P.qt_present[P.tq[c]] || !P.dc[P.td[c]].present || !P.ac[P.ta[c]].present)
            return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: missing quantisation or Huffman table");
        hmax = std::max<uint32_t>(hmax, P.hs[c]); vmax = std::max<uint32_t>(vmax, P.vs[c]);
    }
    if (P.ncomp == 1) { P.hs[0] = P.vs[0] = 1; hmax = vmax = 1; }             // a single-component scan is never interleaved
    P.mcus_w = (P.width + 8 * hmax - 1) / (8 * hmax);
    P.mcus_h = (P.height + 8 * vmax - 1) / (8 * vmax);
    P.blocks_per_mcu = 0;
    for (int c = 0; c < P.ncomp; ++c) { P.bw[c] = P.mcus_w * P.hs[c]; P.bh[c] = P.mcus_h * P.vs[c]; P.blocks_per_mcu += P.hs[c] * P.vs[c]; }
    // The same whitelist as the pixel stage (make_geom): luma carries the maximum factors, chroma is 1x1 -- 4:4:4, 4:2:2,
    // 4:4:0, 4:2:0.  That bounds an MCU at 6 blocks; libjpeg itself refuses more than D_MAX_BLOCKS_IN_MCU = 10
    // (e.g. 2x2 / 2x2 / 2x2), and the per-MCU block tables of the decoder are sized by that constant.
    if (P.ncomp == 3 && (P.hs[0] != hmax || P.vs[0] != vmax || P.hs[1] != 1 || P.vs[1] != 1 || P.hs[2] != 1 || P.vs[2] != 1))
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: sampling %dx%d,%dx%d,%dx%d (chroma must be 1x1)",
                    P.hs[0], P.vs[0], P.hs[1], P.vs[1], P.hs[2], P.vs[2]);
    if (P.blocks_per_mcu > kMaxBlocksInMcu)
        return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: %u blocks per MCU (limit %u)", P.blocks_per_mcu, kMaxBlocksInMcu);
    return IFHIP_OK;
}

void derive_search_table(const HuffSpec& h, SearchTab* t) {
    std::memset(t, 0, sizeof *t);
    std::memcpy(t->val, h.vals, 256);
    int32_t code = 0;
    int k = 0;
    for (int l = 1; l <= 16; ++l) {
        t->valoff[l] = k - code;
        k += h.bits[l];
        code += h.bits[l];
        // Largest code of this length.  A length without codes gets the previous bound extended by a 1 bit instead of
        // jdhuff's -1: the serial search "first l with code_l <= maxcode[l]" is unchanged (it only ever reaches l when
        // code_(l-1) > maxcode[l-1], and then code_l > (maxcode[l-1] << 1 | 1) as well), and "code_l > maxcode[l]" becomes
        // monotone in l, which lets the kernel count the lengths in parallel.
        t->maxcode[l] = h.bits[l] ? code - 1 : (l > 1 ? (t->maxcode[l - 1] < 0 ? -1 : ((t->maxcode[l - 1] << 1) | 1)) : -1);
        code <<= 1;
    }
    t->maxcode[17] = 0x7fffffff;
}
// First level of one table into F->lut[slot], its second level appended to the image's pool (*pool_used entries taken).
void derive_fast_table(const HuffSpec& h, bool ac, FastTabs* F, uint32_t slot, uint32_t* pool_used, uint32_t pool_limit) {
    uint32_t* lut = F->lut[slot];
    for (uint32_t i = 0; i < kLutEntries; ++i) lut[i] = invalid_entry(ac);
    // a DC symbol is a magnitude category, at most 11 in baseline JPEG: larger ones decode as "no such code"
    auto entry = [&](uint32_t l, uint32_t sym) { return (!ac && sym > 11u) ? invalid_entry(ac) : fast_entry(ac, l, sym); };
    struct LongCode { uint32_t code; uint8_t len, sym; };
    std::vector<LongCode> longs;
    uint8_t maxlen[kLutEntries] = {};
    uint32_t code = 0;
    int k = 0;
    for (uint32_t l = 1; l <= 16u; ++l) {
        for (int i = 0; i < h.bits[l]; ++i, ++k, ++code) {
            if (l <= kLutBits) {
                const uint32_t first = code << (kLutBits - l);
                for (uint32_t f = 0; f < (1u << (kLutBits - l)); ++f)
                    if (first + f < kLutEntries) lut[first + f] = entry(l, h.vals[k & 255]);
            } else {
                const uint32_t prefix = code >> (l - kLutBits);
                if (prefix >= kLutEntries) continue;                 // over-subscribed table: the pattern cannot occur
                maxlen[prefix] = static_cast<uint8_t>(l);            // lengths ascend
                longs.push_back(LongCode{code, static_cast<uint8_t>(l), h.vals[k & 255]});
            }
        }
        code <<= 1;
    }
    for (uint32_t prefix = 0; prefix < kLutEntries; ++prefix) {
        if (!maxlen[prefix]) continue;
        const uint32_t n = maxlen[prefix] - kLutBits, size = 1u << n;
        if (*pool_used + size > pool_limit) { lut[prefix] = 0u; continue; }          // left to the serial search
        for (uint32_t i = 0; i < size; ++i) F->pool[*pool_used + i] = invalid_entry(ac);
        lut[prefix] = (*pool_used << 16) | ((32u - n) << 8);
        *pool_used += size;
    }
    for (const LongCode& lc : longs) {
        const uint32_t prefix = lc.code >> (lc.len - kLutBits);
        if (lut[prefix] == 0u) continue;
        const uint32_t n = maxlen[prefix] - kLutBits, rem = lc.len - kLutBits, off = lut[prefix] >> 16;
        const uint32_t sub = (lc.code & ((1u << rem) - 1u)) << (n - rem);
        for (uint32_t f = 0; f < (1u << (n - rem)); ++f) F->pool[off + sub + f] = entry(lc.len, lc.sym);
    }
}
// The pair form of an image's tables (see pair_entry): same first-level pointers and pool layout, every entry carries its
// own skip / advance twice unless the next AC code is visible in the rest of the lookup window.
void derive_pair_tables(const FastTabs& F, int ncomp, FastTabs* Pt) {
    for (uint32_t i = 0; i < kPoolEntries; ++i) Pt->pool[i] = pair_entry(F.pool[i], F.pool[i]);
    for (uint32_t slot = 0; slot < 6u; ++slot)
        for (uint32_t i = 0; i < kLutEntries; ++i) {
            const uint32_t e = F.lut[slot][i];
            uint32_t out = (e & 255u) ? pair_entry(e, e) : e;                        // skip 0: pointer into the pool / serial search
            const uint32_t skip1 = e & 255u, adv1 = (e >> 8) & 255u;
            if ((slot & 1u) && slot < 2u * static_cast<uint32_t>(ncomp) && skip1 != 0u && skip1 < kLutBits && adv1 < 64u) {
                const uint32_t e2 = F.lut[slot][(i << skip1) & (kLutEntries - 1u)];   // the window behind the first symbol
                const uint32_t len2 = (e2 >> 16) & 255u;
                if ((e2 & 255u) != 0u && len2 <= kLutBits - skip1)                   // a whole code (never the 32 of "no such code")
                    out = pair_entry(e, (skip1 + (e2 & 255u)) | ((adv1 + ((e2 >> 8) & 255u)) << 8));
            }
            Pt->lut[slot][i] = out;
        }
}
// The count pass's form: pair entries in the AC tables, the DC tables (and their parts of the pool) as they are.
void derive_count_tables(const FastTabs& F, const FastTabs& Pt, FastTabs* Ct) {
    *Ct = Pt;
    for (uint32_t slot = 0; slot < 6u; slot += 2u)
        for (uint32_t i = 0; i < kLutEntries; ++i) {
            const uint32_t e = F.lut[slot][i];
            Ct->lut[slot][i] = e;
            if ((e & 255u) == 0u && e != 0u) {
                const uint32_t off = e >> 16, size = 1u << (32u - ((e >> 8) & 255u));
                for (uint32_t k = 0; k < size && off + k < kPoolEntries; ++k) Ct->pool[off + k] = F.pool[off + k];
            }
        }
}
// The six tables of one image; components that name the same table share its second level.
void derive_image_tables(const ParsedJpeg& P, FastTabs* F, SearchTab* S6, uint32_t pool_limit, uint32_t* pool_used_out = nullptr) {
    std::memset(F, 0, sizeof *F);
    std::memset(S6, 0, 6u * sizeof *S6);
    uint32_t pool_used = 0;
    for (int c = 0; c < P.ncomp; ++c)
        for (uint32_t ac = 0; ac < 2u; ++ac) {
            const uint32_t slot = 2u * static_cast<uint32_t>(c) + ac;
            const uint8_t id = ac ? P.ta[c] : P.td[c];
            derive_search_table(ac ? P.ac[id] : P.dc[id], &S6[slot]);
            int same = -1;
            for (int o = 0; o < c; ++o) if ((ac ? P.ta[o] : P.td[o]) == id) same = o;
            if (same >= 0) std::memcpy(F->lut[slot], F->lut[2u * static_cast<uint32_t>(same) + ac], sizeof F->lut[slot]);
            else derive_fast_table(ac ? P.ac[id] : P.dc[id], ac != 0u, F, slot, &pool_used, pool_limit);
        }
    if (pool_used_out) *pool_used_out = pool_used;
}

// Un-stuffs the scan (runs between 0xFF bytes are copied whole) and cuts it at restart markers into segments.
// Returns the number of MCUs the segments cover (fewer than the image has: the scan ended early).
uint32_t unstuff_scan(const uint8_t* d, size_t len, const ParsedJpeg& P, std::vector<std::vector<uint8_t>>* seg_bytes,
                      std::vector<uint32_t>* seg_mcu0, std::vector<uint32_t>* seg_mcus) {
    const uint32_t total_mcus = P.mcus_w * P.mcus_h;
    const uint32_t per_seg = P.restart_interval ? P.restart_interval : total_mcus;
    uint32_t mcu0 = 0;
    size_t i = P.scan_begin;
    bool more = true;
    while (more && mcu0 < total_mcus) {
        seg_bytes->emplace_back();
        std::vector<uint8_t>& bytes = seg_bytes->back();
        more = false;
        while (i < len) {
            const uint8_t* ff = static_cast<const uint8_t*>(std::memchr(d + i, 0xFF, len - i));
            const size_t run_end = ff ? static_cast<size_t>(ff - d) : len;
            bytes.insert(bytes.end(), d + i, d + run_end);
            i = run_end;
            if (i >= len) break;
            ++i;                                                           // the 0xFF
            while (i < len && d[i] == 0xFF) ++i;                           // fill bytes before a marker
            if (i >= len) break;
            const uint8_t m = d[i++];
            if (m == 0x00) { bytes.push_back(0xFF); continue; }
            if (m >= 0xD0 && m <= 0xD7) { more = true; break; }            // restart: next segment
            break;                                                         // EOI or any other marker: scan ends
        }
        const uint32_t mcus = std::min(per_seg, total_mcus - mcu0);
        seg_mcu0->push_back(mcu0);
        seg_mcus->push_back(mcus);
        mcu0 += mcus;
    }
    return mcu0;
}

// The same walk, straight into the staging buffer the device reads: every restart segment starts on a sub-sequence boundary
// and is zero-padded to the next one (an empty segment still takes one sub-sequence).  No per-segment vectors: a prepared
// file used to cost a 400 KB vector grown by doubling, a copy out of it and a free that glibc answered with madvise() -- 40 %
// of the host CPU of an ABI job (round 5, profiles/r5_abi_jobs_host_cpu_slots32.txt).  `cap` bytes are available at dst
// (packed_scan_bound); returns the MCUs covered, *n_sub the sub-sequences written.
size_t packed_scan_bound(size_t len, const ParsedJpeg& P) {
    const uint32_t total_mcus = P.mcus_w * P.mcus_h;
    const uint32_t per_seg = P.restart_interval ? P.restart_interval : total_mcus;
    const size_t max_segs = per_seg ? (static_cast<size_t>(total_mcus) + per_seg - 1) / per_seg : 1;
    return (len > P.scan_begin ? len - P.scan_begin : 0) + (max_segs + 1) * (kSubBits / 8u);
}
uint32_t unstuff_scan_packed(const uint8_t* d, size_t len, const ParsedJpeg& P, uint8_t* dst, size_t cap, std::vector<uint32_t>* seg_mcu0,
                             std::vector<uint32_t>* seg_mcus, std::vector<uint64_t>* seg_bits, std::vector<uint32_t>* seg_nsub,
                             std::vector<uint32_t>* seg_first_word, uint64_t* n_sub) {
    constexpr size_t kSubBytes = kSubBits / 8u;
    const uint32_t total_mcus = P.mcus_w * P.mcus_h;
    const uint32_t per_seg = P.restart_interval ? P.restart_interval : total_mcus;
    uint32_t mcu0 = 0;
    uint64_t subs = 0;
    size_t i = P.scan_begin;
    bool more = true;
    while (more && mcu0 < total_mcus) {
        uint8_t* out = dst + subs * kSubBytes;
        size_t nb = 0;
        const size_t room = cap - static_cast<size_t>(subs) * kSubBytes;          // (the bound leaves a sub-sequence per segment beyond the scan's bytes)
        more = false;
        while (i < len) {
            const uint8_t* ff = static_cast<const uint8_t*>(std::memchr(d + i, 0xFF, len - i));
            const size_t run_end = ff ? static_cast<size_t>(ff - d) : len;
            if (nb + (run_end - i) + 1 + kSubBytes > room) throw std::length_error("scan staging bound exceeded");
            std::memcpy(out + nb, d + i, run_end - i);
            nb += run_end - i;
            i = run_end;
            if (i >= len) break;
            ++i;                                                           // the 0xFF
            while (i < len && d[i] == 0xFF) ++i;                           // fill bytes before a marker
            if (i >= len) break;
            const uint8_t m = d[i++];
            if (m == 0x00) { out[nb++] = 0xFF; continue; }
            if (m >= 0xD0 && m <= 0xD7) { more = true; break; }            // restart: next segment
            break;                                                         // EOI or any other marker: scan ends
        }
        const uint64_t bits = static_cast<uint64_t>(nb) * 8u;
        const uint64_t ns = std::max<uint64_t>(1, (bits + kSubBits - 1) / kSubBits);
        std::memset(out + nb, 0, static_cast<size_t>(ns) * kSubBytes - nb);                   // pad to the 1 024-bit boundary
        const uint32_t mcus = std::min(per_seg, total_mcus - mcu0);
        seg_mcu0->push_back(mcu0);
        seg_mcus->push_back(mcus);
        seg_bits->push_back(bits);
        seg_nsub->push_back(static_cast<uint32_t>(ns));
        seg_first_word->push_back(static_cast<uint32_t>(subs * kSubWords));
        subs += ns;
        mcu0 += mcus;
        if (subs * kSubBits + 4096 >= (1ull << 32)) break;                 // (the caller refuses: more than 512 MB of scan data)
    }
    *n_sub = subs;
    return mcu0;
}

}  // namespace

struct ifhip_jpeg_entropy {
    int device = -1;
    uint32_t n_images = 0;
    ParsedJpeg first;                                // geometry shared by the batch
    std::vector<uint16_t> qt;                        // [n][3][64]
    EntropyArgs a;
    std::vector<void*> owned;
    uint32_t* h_flags = nullptr;                     // pinned copy of changed[16] + errors
    uint8_t* h_tables = nullptr;                     // pinned staging of the batch's tables (create_prepared_impl)
    ~ifhip_jpeg_entropy() {
        for (void* p : owned) if (p) (void)DEV_FREE(p);
        if (h_flags) (void)cached_host_free(h_flags);
        if (h_tables) (void)cached_host_free(h_tables);
    }
};

template <typename T>
static int dev_alloc(ifhip_jpeg_entropy* e, T** out, size_t count, const T* init = nullptr) {
    *out = nullptr;
    HIP_TRY(DEV_MALLOC(out, std::max<size_t>(count, 1) * sizeof(T)));
    e->owned.push_back(*out);
    if (init && count) HIP_TRY(static_cast<hipError_t>(copy_to_device(*out, init, count * sizeof(T))));
    return IFHIP_OK;
}

extern "C" {

int ifhip_jpeg_parse_headers(const uint8_t* jpeg, size_t len, uint32_t* width, uint32_t* height, int* n_components,
                             uint8_t* h_samp3, uint8_t* v_samp3, uint32_t* blocks_w3, uint32_t* blocks_h3,
                             uint16_t* qt3x64, uint32_t* restart_interval) {
    ParsedJpeg P;
    int rc = parse_jpeg(jpeg, len, &P);
    if (rc) return rc;
    if (width) *width = P.width;
    if (height) *height = P.height;
    if (n_components) *n_components = P.ncomp;
    for (int c = 0; c < 3; ++c) {
        if (h_samp3) h_samp3[c] = c < P.ncomp ? P.hs[c] : 0;
        if (v_samp3) v_samp3[c] = c < P.ncomp ? P.vs[c] : 0;
        if (blocks_w3) blocks_w3[c] = c < P.ncomp ? P.bw[c] : 0;
        if (blocks_h3) blocks_h3[c] = c < P.ncomp ? P.bh[c] : 0;
        if (qt3x64) {
            if (c < P.ncomp) std::memcpy(qt3x64 + 64 * c, P.qt[P.tq[c]], 128);
            else std::memset(qt3x64 + 64 * c, 0, 128);
        }
    }
    if (restart_interval) *restart_interval = P.restart_interval;
    return IFHIP_OK;
}

// EXIF orientation of a JPEG as MozJpegDecoder reads it (codecs/mozjpeg_decoder_helpers.rs:107-202; the decoder saves
// APP1 and APP2 markers up to 0xffff bytes, mozjpeg_decoder.rs:551-558): the FIRST saved marker whose data starts with
// "Exif\0\0" decides -- shorter than 32 bytes: none; the TIFF header is looked for at offsets 0..15; IFD0's entries are
// walked for tag 0x0112, rejected only if its type is not SHORT AND its count is not 1 (sic, `&&`), values above 8 are none.
// Every read behind the data's end is the reference's io error, i.e. none.  *flag: -1 none, else 0..8.
int ifhip_jpeg_exif_orientation(const uint8_t* d, size_t len, int* flag) {
    if (!flag) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    *flag = -1;
    if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: not a JPEG (no SOI)");
    size_t i = 2;
    while (i + 4 <= len) {
        if (d[i] != 0xFF) break;
        while (i < len && d[i] == 0xFF) ++i;
        if (i >= len) break;
        const uint8_t m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9 || m == 0xDA || i + 2 > len) break;                       // jpeg_read_header stops at SOS
        const size_t seg = (static_cast<size_t>(d[i]) << 8) | d[i + 1];
        if (seg < 2 || i + seg > len) break;
        const uint8_t* q = d + i + 2;
        const size_t n = seg - 2;
        i += seg;
        if ((m != 0xE1 && m != 0xE2) || n < 6 || std::memcmp(q, "Exif\0\0", 6) != 0) continue;
        if (n < 32) return IFHIP_OK;                                             // "EXIF too short"
        size_t tiff = 16;
        bool little = false;
        for (size_t t = 0; t < 16 && tiff == 16; ++t) {
            if (std::memcmp(q + t, "II\x2a\0", 4) == 0) { tiff = t; little = true; }
            else if (std::memcmp(q + t, "MM\0\x2a", 4) == 0) { tiff = t; little = false; }
        }
        if (tiff == 16) return IFHIP_OK;
        const uint8_t* r = q + tiff + 4;
        const uint64_t rn = n - tiff - 4;
        uint64_t pos = 0;
        bool eof = false;
        auto rd = [&](uint32_t bytes) -> uint32_t {
            if (eof || pos + bytes > rn) { eof = true; return 0; }
            uint32_t v = 0;
            for (uint32_t k = 0; k < bytes; ++k) v |= static_cast<uint32_t>(r[pos + k]) << (little ? 8u * k : 8u * (bytes - 1u - k));
            pos += bytes;
            return v;
        };
        const uint32_t offset = rd(4);
        if (eof) return IFHIP_OK;
        pos = static_cast<uint64_t>(std::max<uint32_t>(4u, offset)) - 4u;
        const uint32_t tags = rd(2);
        for (uint32_t k = 0; k < tags && !eof; ++k) {
            if (rd(2) == 0x112u && !eof) {
                const uint32_t type = rd(2), count = rd(4);
                if (eof || (type != 3u && count != 1u)) return IFHIP_OK;
                const uint32_t v = rd(2);
                if (!eof && v <= 8u) *flag = static_cast<int>(v);
                return IFHIP_OK;
            }
            pos += 10;
        }
        return IFHIP_OK;
    }
    return IFHIP_OK;
}

// The embedded ICC profile (see include/imageflow_hip.h).  libjpeg's jpeg_read_icc_profile rule: APP2 markers whose data starts
// with "ICC_PROFILE\0", then a 1-based sequence number and the marker count; every number must occur exactly once with the
// same count.  A chunk set that breaks the rule -- inconsistent counts, a sequence number out of range or twice, a missing
// chunk, nothing but empty chunks -- is NO profile to the reference (read_icc_profile returns None,
// mozjpeg_decoder_helpers.rs:42-83: the frame is decoded without a transform) and kind 0 here; so is a GRAY profile on a
// colour frame (mozjpeg_decoder.rs:391-395 -> SourceProfile::Srgb).
int ifhip_jpeg_icc_profile_kind(const uint8_t* d, size_t len, int* kind) {
    if (!kind) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    *kind = 0;
    if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: not a JPEG (no SOI)");
    std::vector<std::pair<const uint8_t*, size_t>> chunk(256, {nullptr, 0});
    uint32_t count = 0, seen = 0;
    bool broken = false;
    int frame_components = 0;
    size_t i = 2;
    while (i + 4 <= len) {
        if (d[i] != 0xFF) break;
        while (i < len && d[i] == 0xFF) ++i;
        if (i >= len) break;
        const uint8_t m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9 || m == 0xDA || i + 2 > len) break;
        const size_t seg = (static_cast<size_t>(d[i]) << 8) | d[i + 1];
        if (seg < 2 || i + seg > len) break;
        const uint8_t* q = d + i + 2;
        const size_t n = seg - 2;
        i += seg;
        if (m >= 0xC0 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC && n >= 6) frame_components = q[5];     // SOFn: P, Y, X, Nf
        if (m != 0xE2 || n < 14 || std::memcmp(q, "ICC_PROFILE\0", 12) != 0) continue;
        const uint32_t seq = q[12], num = q[13];
        if (seq == 0 || num == 0 || seq > num || (count && num != count) || chunk[seq].first) { broken = true; continue; }
        count = num;
        chunk[seq] = {q + 14, n - 14};
        ++seen;
    }
    if (!seen || broken || seen != count) return IFHIP_OK;              // no profile, or a chunk set the reference drops
    std::vector<uint8_t> icc;
    for (uint32_t k = 1; k <= count; ++k) icc.insert(icc.end(), chunk[k].first, chunk[k].first + chunk[k].second);
    if (icc.empty()) return IFHIP_OK;                                    // only empty markers: None
    if (icc.size() >= 20 && std::memcmp(&icc[16], "GRAY", 4) == 0 && frame_components != 1) return IFHIP_OK;   // -> SourceProfile::Srgb
    *kind = 2;
    // ICC.1 header: size (0..3), colour space (16..19), PCS (20..23); tag table at 128: count, then {sig, offset, size}
    auto be32 = [&](size_t o) { return (static_cast<uint32_t>(icc[o]) << 24) | (static_cast<uint32_t>(icc[o + 1]) << 16) | (static_cast<uint32_t>(icc[o + 2]) << 8) | icc[o + 3]; };
    if (icc.size() < 132 || std::memcmp(&icc[16], "RGB ", 4) != 0 || std::memcmp(&icc[20], "XYZ ", 4) != 0) return IFHIP_OK;
    const uint32_t tags = be32(128);
    if (tags > 200 || 132 + static_cast<size_t>(tags) * 12 > icc.size()) return IFHIP_OK;
    auto find = [&](const char* sig, size_t* off, size_t* size) {
        for (uint32_t t = 0; t < tags; ++t) {
            const size_t e = 132 + static_cast<size_t>(t) * 12;
            if (std::memcmp(&icc[e], sig, 4) != 0) continue;
            *off = be32(e + 4); *size = be32(e + 8);
            return *off + *size <= icc.size() && *size >= 8;
        }
        return false;
    };
    // primaries, D50-adapted (IEC 61966-2-1 through Bradford, as every sRGB profile carries them)
    static const double want[3][3] = {{0.4360, 0.2225, 0.0139}, {0.3851, 0.7169, 0.0971}, {0.1431, 0.0606, 0.7141}};
    const char* xyz_sig[3] = {"rXYZ", "gXYZ", "bXYZ"};
    for (int c = 0; c < 3; ++c) {
        size_t off = 0, size = 0;
        if (!find(xyz_sig[c], &off, &size) || size < 20 || std::memcmp(&icc[off], "XYZ ", 4) != 0) return IFHIP_OK;
        for (int k = 0; k < 3; ++k) {
            const double v = static_cast<int32_t>(be32(off + 8 + 4 * static_cast<size_t>(k))) / 65536.0;
            if (std::fabs(v - want[c][k]) > 0.003) return IFHIP_OK;
        }
    }
    // tone curves: the sRGB parametric form (type 3: g 2.4, a 1/1.055, b 0.055/1.055, c 1/12.92, d 0.04045) or a sampled curve
    // whose entries follow the sRGB function (checked at every entry; 2 of 65535 is the 16-bit tables' own rounding)
    const char* trc_sig[3] = {"rTRC", "gTRC", "bTRC"};
    for (int c = 0; c < 3; ++c) {
        size_t off = 0, size = 0;
        if (!find(trc_sig[c], &off, &size) || size < 12) return IFHIP_OK;
        if (std::memcmp(&icc[off], "para", 4) == 0) {
            if (size < 12 + 20 || ((static_cast<uint32_t>(icc[off + 8]) << 8) | icc[off + 9]) != 3u) return IFHIP_OK;
            static const double p[5] = {2.4, 1.0 / 1.055, 0.055 / 1.055, 1.0 / 12.92, 0.04045};
            for (int k = 0; k < 5; ++k)
                if (std::fabs(static_cast<int32_t>(be32(off + 12 + 4 * static_cast<size_t>(k))) / 65536.0 - p[k]) > 0.002) return IFHIP_OK;
        } else if (std::memcmp(&icc[off], "curv", 4) == 0) {
            const uint32_t n = be32(off + 8);
            if (n < 256 || 12 + static_cast<size_t>(n) * 2 > size) return IFHIP_OK;      // (a single gamma value is not the sRGB curve)
            for (uint32_t k = 0; k < n; ++k) {
                const double x = static_cast<double>(k) / (n - 1), y = x <= 0.04045 ? x / 12.92 : std::pow((x + 0.055) / 1.055, 2.4);
                const double got = ((static_cast<uint32_t>(icc[off + 12 + 2 * static_cast<size_t>(k)]) << 8) | icc[off + 13 + 2 * static_cast<size_t>(k)]) / 65535.0;
                if (std::fabs(got - y) > 2.0 / 65535.0 + 1e-4) return IFHIP_OK;
            }
        } else {
            return IFHIP_OK;
        }
    }
    *kind = 1;
    return IFHIP_OK;
}

}  // extern "C"  (reopened below)

// One file, prepared on the host by whoever owns it: parsed, its scan un-stuffed, cut at the restart markers and packed as
// big-endian words (every segment padded to 1 024 bits) in PINNED memory, its decode tables derived.  A batch is then
// assembled from n of these with no further pass over the data: n asynchronous uploads.  (A service that runs one job per
// thread prepares each job's file on that job's thread -- imageflow_abi/src/lib.rs:20-27 -- and hands the handles of
// whatever is waiting to ONE device call.)
struct ifhip_jpeg_prepared {
    ParsedJpeg P;
    uint16_t qt[192];
    std::vector<uint32_t> seg_mcu0, seg_mcus, seg_nsub, seg_first_word;      // per restart segment
    std::vector<uint64_t> seg_bits;
    uint32_t* words = nullptr;                       // pinned (devmem cache)
    size_t n_words = 0;
    uint32_t n_sub = 0;
    FastTabs ftabs, ptabs, ctabs;
    SearchTab stabs[6];
    uint32_t pool_limit = kPoolEntries;
    // ifhip_jpeg_prepared_upload: the words on the device already, queued on the owner's stream; `uploaded` marks the copy's end
    // (the event stays with a recycled handle, the block does not)
    uint32_t* d_words = nullptr;
    hipEvent_t uploaded = nullptr;
    int d_device = -1, event_device = -1;
    void drop_device_copy() {                        // nothing may still write the block when it goes back to the cache
        if (!d_words) return;
        if (uploaded && hipEventQuery(uploaded) != hipSuccess) { (void)hipGetLastError(); (void)hipEventSynchronize(uploaded); }
        (void)hipGetLastError();
        (void)DEV_FREE(d_words);
        d_words = nullptr; d_device = -1;
    }
    ~ifhip_jpeg_prepared() {
        if (words) (void)cached_host_free(words);
        drop_device_copy();
        if (uploaded) (void)hipEventDestroy(uploaded);
    }
};

// Prepared handles are recycled: 50 KB of tables each, one per job through the libimageflow ABI.
namespace {
struct PreparedPool { std::mutex mu; std::vector<ifhip_jpeg_prepared*> spare; };
PreparedPool& prepared_pool() { static PreparedPool* p = new PreparedPool; return *p; }   // (never destroyed: handles may outlive static teardown)
ifhip_jpeg_prepared* take_prepared() {
    {
        PreparedPool& pool = prepared_pool();
        std::lock_guard<std::mutex> lk(pool.mu);
        if (!pool.spare.empty()) { ifhip_jpeg_prepared* p = pool.spare.back(); pool.spare.pop_back(); return p; }
    }
    return new ifhip_jpeg_prepared;
}
void give_prepared(ifhip_jpeg_prepared* p) {
    if (!p) return;
    p->drop_device_copy();                                               // (before the pinned source goes: the copy reads it)
    if (p->words) { (void)cached_host_free(p->words); p->words = nullptr; }
    p->n_words = 0; p->n_sub = 0;
    p->seg_mcu0.clear(); p->seg_mcus.clear(); p->seg_nsub.clear(); p->seg_first_word.clear(); p->seg_bits.clear();
    PreparedPool& pool = prepared_pool();
    {
        std::lock_guard<std::mutex> lk(pool.mu);
        if (pool.spare.size() < 64) { pool.spare.push_back(p); return; }
    }
    delete p;
}
struct GivePrepared { void operator()(ifhip_jpeg_prepared* p) const { give_prepared(p); } };
}  // namespace

static uint32_t test_hook_u32(const char* name, uint32_t dflt, uint32_t hi) {
    // test hooks: the rarely taken paths (serial code search for sub-tables that overflow the pool; further rounds after an
    // unsettled count pass) are forced by shrinking the pool / the iterations per launch
    const char* v = debug_switch(name);
    return v ? std::min<uint32_t>(static_cast<uint32_t>(std::strtoul(v, nullptr, 10)), hi) : dflt;
}

static int prepare_impl(ifhip_jpeg_prepared** out, const uint8_t* file, size_t len, uint32_t index_for_messages) {
    std::unique_ptr<ifhip_jpeg_prepared, GivePrepared> R(take_prepared());
    ParsedJpeg& P = R->P;
    P = ParsedJpeg();
    if (int rc = parse_jpeg(file, len, &P)) return rc;
    std::memset(R->qt, 0, sizeof R->qt);
    std::memset(&R->ftabs, 0, sizeof R->ftabs); std::memset(&R->ptabs, 0, sizeof R->ptabs); std::memset(&R->ctabs, 0, sizeof R->ctabs);   // (a recycled handle)
    std::memset(R->stabs, 0, sizeof R->stabs);
    for (int c = 0; c < P.ncomp; ++c) std::memcpy(&R->qt[c * 64], P.qt[P.tq[c]], 128);
    R->pool_limit = test_hook_u32("ent_test_pool", kPoolEntries, kPoolEntries);
    derive_image_tables(P, &R->ftabs, R->stabs, R->pool_limit);
    derive_pair_tables(R->ftabs, P.ncomp, &R->ptabs);
    derive_count_tables(R->ftabs, R->ptabs, &R->ctabs);
    const uint32_t total_mcus = P.mcus_w * P.mcus_h;
    void* pinned = nullptr;
    size_t pinned_bytes = 4096;                                          // powers of two: few size classes in the pinned cache, whatever the files
    const size_t bound = packed_scan_bound(len, P);
    while (pinned_bytes < bound) pinned_bytes *= 2;
    if (cached_host_malloc(&pinned, pinned_bytes) != 0)
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: %zu bytes of pinned staging", pinned_bytes);
    R->words = static_cast<uint32_t*>(pinned);                           // (R owns it from here: every return below releases it)
    uint64_t subs = 0;
    const uint32_t mcu0 = unstuff_scan_packed(file, len, P, reinterpret_cast<uint8_t*>(pinned), pinned_bytes, &R->seg_mcu0, &R->seg_mcus, &R->seg_bits,
                                              &R->seg_nsub, &R->seg_first_word, &subs);
    if (subs * kSubBits + 4096 >= (1ull << 32)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: more than 512 MB of scan data");
    if (mcu0 < total_mcus)
        return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: scan of image %u ends after %u of %u MCUs", index_for_messages, mcu0, total_mcus);
    R->n_sub = static_cast<uint32_t>(subs);
    R->n_words = static_cast<size_t>(subs) * kSubWords;
    for (size_t q = 0; q < R->n_words; ++q) R->words[q] = __builtin_bswap32(R->words[q]);   // stream order bytes -> big-endian words
    *out = R.release();
    return IFHIP_OK;
}

static int create_prepared_impl(ifhip_jpeg_entropy** out, ifhip_jpeg_prepared* const* prep, uint32_t n_images) {
    std::unique_ptr<ifhip_jpeg_entropy> e(new ifhip_jpeg_entropy);
    if (int arc = require_gfx950(&e->device)) return arc;
    e->n_images = n_images;
    e->qt.assign(static_cast<size_t>(n_images) * 192u, 0);
    // The batch's tables -- segments, sub-sequence -> segment, three look-up tables and six search tables per image -- are laid
    // out in ONE pinned staging block and go to ONE device block in one copy (they were six allocations and six uploads, each
    // with its own wait for the stream: most of the 1.9 ms a 16-file batch spent here, profiles/r5_abi_jobs_cliff_7_decode_batches.txt).
    size_t n_segs = 0;
    uint64_t subs_in_batch = 0;
    for (uint32_t img = 0; img < n_images; ++img) { n_segs += prep[img]->seg_nsub.size(); subs_in_batch += prep[img]->n_sub; }
    if (subs_in_batch * kSubBits + 4096 >= (1ull << 32)) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: batch holds more than 512 MB of scan data");
    auto aligned = [](size_t v) { return (v + 255u) & ~static_cast<size_t>(255u); };
    const size_t o_segs = 0, o_sub = aligned(o_segs + std::max<size_t>(n_segs, 1) * sizeof(Segment)),
                 o_ftabs = aligned(o_sub + std::max<size_t>(subs_in_batch, 1) * sizeof(uint32_t)), o_ptabs = aligned(o_ftabs + n_images * sizeof(FastTabs)),
                 o_ctabs = aligned(o_ptabs + n_images * sizeof(FastTabs)), o_stabs = aligned(o_ctabs + n_images * sizeof(FastTabs)),
                 o_order = aligned(o_stabs + static_cast<size_t>(n_images) * 6u * sizeof(SearchTab)),
                 table_bytes = aligned(o_order + ((subs_in_batch + kOwnSubs - 1u) / kOwnSubs + 1u) * sizeof(uint32_t));
    size_t stage_bytes = 4096;                                           // powers of two: few size classes in the pinned cache
    while (stage_bytes < table_bytes) stage_bytes *= 2;
    if (cached_host_malloc(reinterpret_cast<void**>(&e->h_tables), stage_bytes) != 0)
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: %zu bytes of pinned staging", stage_bytes);
    uint8_t* const stage = e->h_tables;
    Segment* const segs = reinterpret_cast<Segment*>(stage + o_segs);
    uint32_t* const sub_seg = reinterpret_cast<uint32_t*>(stage + o_sub);
    FastTabs* const ftabs = reinterpret_cast<FastTabs*>(stage + o_ftabs);
    FastTabs* const ptabs = reinterpret_cast<FastTabs*>(stage + o_ptabs);
    FastTabs* const ctabs = reinterpret_cast<FastTabs*>(stage + o_ctabs);
    SearchTab* const stabs = reinterpret_cast<SearchTab*>(stage + o_stabs);
    uint32_t* const wg_order = reinterpret_cast<uint32_t*>(stage + o_order);
    std::vector<uint32_t> first_sub_of(n_images);
    uint64_t total_subs = 0;
    size_t seg_count = 0;
    for (uint32_t img = 0; img < n_images; ++img) {
        const ifhip_jpeg_prepared& R = *prep[img];
        const ParsedJpeg& P = R.P;
        if (img == 0) e->first = P;
        else {
            const ParsedJpeg& F = e->first;
            bool same = P.width == F.width && P.height == F.height && P.ncomp == F.ncomp;
            for (int c = 0; c < P.ncomp && same; ++c) same = P.hs[c] == F.hs[c] && P.vs[c] == F.vs[c];
            if (!same) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: image %u differs in size or sampling from image 0 (one batch = one geometry)", img);
        }
        std::memcpy(&e->qt[static_cast<size_t>(img) * 192u], R.qt, sizeof R.qt);
        std::memcpy(&ftabs[img], &R.ftabs, sizeof(FastTabs)); std::memcpy(&ptabs[img], &R.ptabs, sizeof(FastTabs)); std::memcpy(&ctabs[img], &R.ctabs, sizeof(FastTabs));
        std::memcpy(&stabs[static_cast<size_t>(img) * 6u], R.stabs, sizeof R.stabs);
        first_sub_of[img] = static_cast<uint32_t>(total_subs);
        for (size_t k = 0; k < R.seg_nsub.size(); ++k) {
            Segment sg;
            std::memset(&sg, 0, sizeof sg);
            sg.image = img;
            sg.first_mcu = R.seg_mcu0[k];
            sg.n_blocks = R.seg_mcus[k] * P.blocks_per_mcu;
            sg.first_sub = static_cast<uint32_t>(total_subs);
            sg.n_sub = R.seg_nsub[k];
            sg.bit_end = static_cast<uint32_t>(total_subs * kSubBits + R.seg_bits[k]);
            for (uint32_t q = 0; q < sg.n_sub; ++q) sub_seg[total_subs + q] = static_cast<uint32_t>(seg_count);
            total_subs += sg.n_sub;
            segs[seg_count++] = sg;
        }
    }
    const size_t n_words = static_cast<size_t>(total_subs) * kSubWords + 64u;  // + lookahead slack behind the last segment
    {   // dispatch order of the synchronisation launch (see entropy_round_kernel): blocks of the smallest scans first, an image's
        // blocks in their own order
        const uint32_t n_wg = static_cast<uint32_t>((total_subs + kOwnSubs - 1u) / kOwnSubs);
        std::vector<uint32_t> key(n_wg);
        uint32_t img = 0;
        for (uint32_t w = 0; w < n_wg; ++w) {
            const uint64_t s0 = static_cast<uint64_t>(w) * kOwnSubs;
            while (img + 1u < n_images && first_sub_of[img + 1u] <= s0) ++img;
            key[w] = prep[img]->n_sub;
            wg_order[w] = w;
        }
        std::stable_sort(wg_order, wg_order + n_wg, [&](uint32_t x, uint32_t y) { return key[x] < key[y]; });
    }
    const ParsedJpeg& F = e->first;
    EntropyArgs& a = e->a;
    std::memset(&a, 0, sizeof a);
    a.g.ncomp = static_cast<uint32_t>(F.ncomp); a.g.blocks_per_mcu = F.blocks_per_mcu; a.g.mcus_w = F.mcus_w; a.g.mcus_h = F.mcus_h;
    uint32_t k = 0;
    for (int c = 0; c < F.ncomp; ++c) {
        a.g.bw[c] = F.bw[c]; a.g.bh[c] = F.bh[c]; a.g.hs[c] = F.hs[c]; a.g.vs[c] = F.vs[c];
        for (uint32_t dy = 0; dy < F.vs[c]; ++dy)
            for (uint32_t dx = 0; dx < F.hs[c]; ++dx, ++k) {
                a.g.kcomp[k] = static_cast<uint8_t>(c); a.g.kdx[k] = static_cast<uint8_t>(dx); a.g.kdy[k] = static_cast<uint8_t>(dy);
                a.g.kcomp_packed |= static_cast<uint32_t>(c) << (2u * k);
            }
    }
    a.inner_rounds = std::max<uint32_t>(1u, test_hook_u32("ent_test_inner", kInnerRounds, kInnerRounds));
    a.n_sub = static_cast<uint32_t>(total_subs);
    a.n_seg = static_cast<uint32_t>(seg_count);
    a.uniform_tables = 1u;                           // batches of small files put many images into one workgroup: with
    for (uint32_t img = 1; img < n_images && a.uniform_tables; ++img)     // identical tables they all use the LDS copy
        if (std::memcmp(&ftabs[img], &ftabs[0], sizeof(FastTabs)) != 0 ||
            std::memcmp(&stabs[static_cast<size_t>(img) * 6u], &stabs[0], 6u * sizeof(SearchTab)) != 0) a.uniform_tables = 0u;
    int rc;
    uint32_t *d_words = nullptr, *d_sub = nullptr;
    Segment* d_segs = nullptr;
    FastTabs *d_ftabs = nullptr, *d_ptabs = nullptr, *d_ctabs = nullptr;
    SearchTab* d_stabs = nullptr;
    if ((rc = dev_alloc<uint32_t>(e.get(), &d_words, n_words))) return rc;
    // From here on copies into `d_words` FROM the callers' pinned buffers may be queued on the thread's stream.  On any failure
    // below `e` is destroyed and its blocks go back to the cache -- inside an ABI job without a device-wide wait (QuiescedScope)
    // -- and the callers free their pinned sources: neither may happen with a copy still in flight, or the next owner of
    // either block is written over.  Every exit path of this function therefore waits for the stream first.
    struct StreamDrain {
        hipStream_t st;
        ~StreamDrain() { (void)static_cast<hipError_t>(ifhip::wait_stream(st)); }
    } drain{static_cast<hipStream_t>(thread_stream())};
    {   // every file's words straight from its pinned buffer to its place: n asynchronous copies, one wait (below, with the tables')
        hipStream_t st = drain.st;
        for (uint32_t img = 0; img < n_images; ++img) {
            const ifhip_jpeg_prepared& R = *prep[img];
            if (!R.n_words) continue;
            uint32_t* const dst = d_words + static_cast<size_t>(first_sub_of[img]) * kSubWords;
            if (R.d_words && R.d_device == e->device) {              // its owner queued the upload already (ifhip_jpeg_prepared_upload): behind that copy, device to device
                HIP_TRY(hipStreamWaitEvent(st, R.uploaded, 0));
                HIP_TRY(hipMemcpyAsync(dst, R.d_words, R.n_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
            } else {
                HIP_TRY(hipMemcpyAsync(dst, R.words, R.n_words * sizeof(uint32_t), hipMemcpyHostToDevice, st));
            }
        }
        HIP_TRY(hipMemsetAsync(d_words + (n_words - 64u), 0, 64u * sizeof(uint32_t), st));
    }
    uint8_t* d_tables = nullptr;
    if ((rc = dev_alloc<uint8_t>(e.get(), &d_tables, table_bytes))) return rc;
    HIP_TRY(hipMemcpyAsync(d_tables, stage, table_bytes, hipMemcpyHostToDevice, drain.st));      // (the wait: StreamDrain, on every way out)
    d_segs = reinterpret_cast<Segment*>(d_tables + o_segs); d_sub = reinterpret_cast<uint32_t*>(d_tables + o_sub);
    d_ftabs = reinterpret_cast<FastTabs*>(d_tables + o_ftabs); d_ptabs = reinterpret_cast<FastTabs*>(d_tables + o_ptabs);
    d_ctabs = reinterpret_cast<FastTabs*>(d_tables + o_ctabs); d_stabs = reinterpret_cast<SearchTab*>(d_tables + o_stabs);
    a.wg_order = reinterpret_cast<const uint32_t*>(d_tables + o_order);
    a.words = d_words; a.segs = d_segs; a.sub_seg = d_sub; a.ftabs = d_ftabs; a.ptabs = d_ptabs; a.ctabs = d_ctabs; a.stabs = d_stabs;
    for (int b = 0; b < 2; ++b) {
        if ((rc = dev_alloc<uint32_t>(e.get(), &a.exit_p[b], a.n_sub))) return rc;
        if ((rc = dev_alloc<uint32_t>(e.get(), &a.exit_cz[b], a.n_sub))) return rc;
    }
    if ((rc = dev_alloc<uint32_t>(e.get(), &a.start_p, a.n_sub))) return rc;
    if ((rc = dev_alloc<uint32_t>(e.get(), &a.start_cz, a.n_sub))) return rc;
    if ((rc = dev_alloc<int4>(e.get(), &a.cnt, a.n_sub))) return rc;
    if ((rc = dev_alloc<int4>(e.get(), &a.tail, (a.n_sub + kChunkSubs - 1u) / kChunkSubs))) return rc;
    if ((rc = dev_alloc<uint32_t>(e.get(), &a.changed, kFlagWords))) return rc;
    a.errors = a.changed + 16;
    a.unsettled = a.changed + 17;
    HIP_TRY(static_cast<hipError_t>(cached_host_malloc(reinterpret_cast<void**>(&e->h_flags), kFlagWords * sizeof(uint32_t))));
    *out = e.release();
    return IFHIP_OK;
}

static int entropy_create_impl(ifhip_jpeg_entropy** out, const uint8_t* const* files, const size_t* lengths, uint32_t n_images) {
    if (!out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    *out = nullptr;
    if (!files || !lengths || n_images == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: empty batch");
    if (int arc = require_gfx950(nullptr)) return arc;
    // Host preparation runs on a few threads, one file at a time each: parsing and un-stuffing are independent per file
    // and would otherwise take several times longer than the GPU needs to decode the batch.
    std::vector<std::unique_ptr<ifhip_jpeg_prepared, GivePrepared>> prep(n_images);
    std::vector<int> rcs(n_images, IFHIP_OK);
    std::vector<std::string> messages(n_images);
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const uint32_t n_threads = std::min<uint32_t>(std::min<uint32_t>(n_images, hw), 16u);
    // Workers never let an exception escape (a bad_alloc on a huge scan becomes that file's error), and a thread that
    // cannot be created (container thread limits) leaves its share of the files to the calling thread.
    auto guarded = [&](uint32_t i) {
        try {
            ifhip_jpeg_prepared* one = nullptr;
            rcs[i] = prepare_impl(&one, files[i], lengths[i], i);
            if (rcs[i]) messages[i] = last_error();
            prep[i].reset(one);
        }
        catch (const std::bad_alloc&) { rcs[i] = IFHIP_ALLOCATION_FAILED; messages[i] = "AllocationFailed: host memory while preparing the scan"; }
        catch (const std::exception& ex) { rcs[i] = IFHIP_INVALID_STATE; messages[i] = std::string("InvalidState: ") + ex.what(); }
    };
    {
        std::vector<std::thread> pool;
        std::vector<uint32_t> inline_lanes{0u};
        for (uint32_t t = 1; t < n_threads; ++t) {
            try { pool.emplace_back([&, t] { for (uint32_t i = t; i < n_images; i += n_threads) guarded(i); }); }
            catch (const std::exception&) { inline_lanes.push_back(t); }
        }
        for (uint32_t t : inline_lanes)
            for (uint32_t i = t; i < n_images; i += n_threads) guarded(i);
        for (auto& th : pool) th.join();
    }
    for (uint32_t img = 0; img < n_images; ++img)                              // errors in file order
        if (rcs[img]) return fail(rcs[img], "%s", messages[img].c_str());
    std::vector<ifhip_jpeg_prepared*> raw(n_images);
    for (uint32_t img = 0; img < n_images; ++img) raw[img] = prep[img].get();
    return create_prepared_impl(out, raw.data(), n_images);
}

extern "C" {

int ifhip_jpeg_entropy_prepare(ifhip_jpeg_prepared** out, const uint8_t* jpeg, size_t len) {
    if (!out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    *out = nullptr;
    if (!jpeg) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null file");
    try { return prepare_impl(out, jpeg, len, 0); }
    catch (const std::bad_alloc&) { return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory while preparing the scan"); }
    catch (const std::exception& ex) { return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what()); }
}
void ifhip_jpeg_prepared_destroy(ifhip_jpeg_prepared* p) { give_prepared(p); }
int ifhip_jpeg_prepared_upload(ifhip_jpeg_prepared* p, void* hip_stream) {
    if (!p) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null handle");
    if (p->d_words || !p->n_words) return IFHIP_OK;
    int dev = -1;
    if (int arc = require_gfx950(&dev)) return arc;
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (p->uploaded && p->event_device != dev) { (void)hipEventDestroy(p->uploaded); p->uploaded = nullptr; }   // (a recycled handle, last used on another device)
    if (!p->uploaded) { HIP_TRY(hipEventCreateWithFlags(&p->uploaded, hipEventDisableTiming)); p->event_device = dev; }
    HIP_TRY(DEV_MALLOC(&p->d_words, p->n_words * sizeof(uint32_t)));
    p->d_device = dev;
    hipError_t er = hipMemcpyAsync(p->d_words, p->words, p->n_words * sizeof(uint32_t), hipMemcpyHostToDevice, st);
    if (er == hipSuccess) er = hipEventRecord(p->uploaded, st);
    if (er != hipSuccess) {
        (void)hipGetLastError();
        (void)static_cast<hipError_t>(ifhip::wait_stream(st));        // (whatever was queued is over before the block goes back)
        (void)DEV_FREE(p->d_words);
        p->d_words = nullptr; p->d_device = -1;
        return fail(IFHIP_GPU_ERROR, "GpuError: %s while queueing the scan upload", hipGetErrorString(er));
    }
    return IFHIP_OK;
}
int ifhip_jpeg_prepared_info(const ifhip_jpeg_prepared* p, uint32_t* width, uint32_t* height, int* n_components, uint8_t* h_samp3, uint8_t* v_samp3) {
    if (!p) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null handle");
    if (width) *width = p->P.width;
    if (height) *height = p->P.height;
    if (n_components) *n_components = p->P.ncomp;
    for (int c = 0; c < 3; ++c) {
        if (h_samp3) h_samp3[c] = c < p->P.ncomp ? p->P.hs[c] : 0;
        if (v_samp3) v_samp3[c] = c < p->P.ncomp ? p->P.vs[c] : 0;
    }
    return IFHIP_OK;
}
int ifhip_jpeg_entropy_create_prepared(ifhip_jpeg_entropy** out, ifhip_jpeg_prepared* const* prepared, uint32_t n_images) {
    if (!out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null out-pointer");
    *out = nullptr;
    if (!prepared || n_images == 0) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: empty batch");
    for (uint32_t i = 0; i < n_images; ++i) if (!prepared[i]) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null prepared file %u", i);
    try { return create_prepared_impl(out, prepared, n_images); }
    catch (const std::bad_alloc&) { return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory while assembling the batch"); }
    catch (const std::exception& ex) { return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what()); }
}

void ifhip_jpeg_entropy_destroy(ifhip_jpeg_entropy* e) { delete e; }

int ifhip_jpeg_entropy_create(ifhip_jpeg_entropy** out, const uint8_t* const* files, const size_t* lengths, uint32_t n_images) {
    try {                                                   // nothing C++ crosses the C ABI
        return entropy_create_impl(out, files, lengths, n_images);
    } catch (const std::bad_alloc&) {
        if (out) *out = nullptr;
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory while preparing the batch");
    } catch (const std::exception& ex) {
        if (out) *out = nullptr;
        return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what());
    }
}

// ---- diagnostic (host only, no GPU): the tables and the scan layout of one file, checked against each other ------------
// Everything the kernels rely on that can be checked without them: the two-level tables decode the scan to exactly the
// blocks the geometry asks for (and to the DC values returned in dc_last: the caller compares them with a serial decoder's),
// and a walk with the pair tables / the count tables passes through the same state at every sub-sequence boundary as the
// plain walk -- the property that lets the synchronisation rounds, the count pass and the write pass hand states to
// each other.  Host mirrors of long_entry / walk (same entry formats, same rules); nothing here produces coefficients.
namespace {
struct HostStream {
    const std::vector<uint8_t>& b;
    uint32_t peek(uint64_t p) const {                                // 32 bits behind bit position p, zero behind the end
        uint64_t v = 0;
        const size_t at = static_cast<size_t>(p >> 3);
        for (size_t i = 0; i < 5; ++i) v = (v << 8) | (at + i < b.size() ? b[at + i] : 0u);
        return static_cast<uint32_t>((v << (p & 7u)) >> 8);
    }
};
uint32_t host_long_entry(const FastTabs& T, const SearchTab* S6, uint32_t slot, uint32_t e, uint32_t bits, int pairs) {
    if (e != 0u) return T.pool[(e >> 16) + ((bits << kLutBits) >> ((e >> 8) & 255u))];
    const SearchTab& t = S6[slot];
    const bool ac = (slot & 1u) != 0u;
    uint32_t l = kLutBits + 1u;
    for (uint32_t k = kLutBits + 1u; k <= 16u; ++k) l += static_cast<int32_t>(bits >> (32u - k)) > t.maxcode[k] ? 1u : 0u;
    uint32_t r = invalid_entry(ac);
    if (l <= 16u) {
        const uint32_t sym = t.val[(static_cast<int32_t>(bits >> (32u - l)) + t.valoff[l]) & 255];
        if (ac || sym <= 11u) r = fast_entry(ac, l, sym);
    }
    return (pairs == 1 || (pairs == 2 && ac)) ? pair_entry(r, r) : r;
}
struct HostState { uint64_t p; uint32_t c, z; };
struct HostCounts { uint64_t reads = 0, blocks = 0; int32_t dc[3] = {0, 0, 0}; bool bad = false; };
// mirror of walk(): from state s to the first symbol boundary at or behind `end` (or until `max_blocks` blocks started)
HostState host_walk(const EntropyGeom& g, const HostStream& in, const FastTabs& T, const SearchTab* S6, int pairs, HostState s, uint64_t end,
                    uint64_t max_blocks, HostCounts* n) {
    while (s.p < end) {
        const uint32_t comp = (g.kcomp_packed >> (2u * s.c)) & 3u, slot = comp * 2u + (s.z ? 1u : 0u);
        if (s.z == 0u && n->blocks >= max_blocks) break;             // pad bits behind the last block
        const uint32_t bits = in.peek(s.p);
        uint32_t e = T.lut[slot][bits >> (32u - kLutBits)];
        if ((e & 255u) == 0u) e = host_long_entry(T, S6, slot, e, bits, pairs);
        ++n->reads;
        if (s.z == 0u) {
            ++n->blocks;
            if (pairs != 1) {                                        // DC entries are plain in the plain and the count tables
                const uint32_t len = (e >> 16) & 255u, sz = (e >> 24) & 15u;
                if (len > 16u) n->bad = true;
                else if (sz) {
                    const uint32_t v = (bits << len) >> (32u - sz);
                    n->dc[comp] += static_cast<int32_t>(v) - ((v >> (sz - 1u)) ? 0 : static_cast<int32_t>((1u << sz) - 1u));
                }
            }
        }
        bool both = false;
        if (pairs == 1 || (pairs == 2 && s.z != 0u)) both = s.z + ((e >> 8) & 255u) < 64u && s.p + (e & 255u) < end;
        if (pairs == 0 && ((e >> 21) & 1u)) n->bad = true;
        if (both) e >>= 16;
        s.p += e & 255u;
        s.z += (e >> 8) & 255u;
        if (s.z >= 64u) { s.z = 0u; s.c = s.c + 1u == g.blocks_per_mcu ? 0u : s.c + 1u; }
    }
    return s;
}
}  // namespace

int ifhip_jpeg_debug_scan_report(const uint8_t* jpeg, size_t len, ifhip_jpeg_scan_report* out) {
    if (!jpeg || !out) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    try {
        std::memset(out, 0, sizeof *out);
        ParsedJpeg P;
        if (int rc = parse_jpeg(jpeg, len, &P)) return rc;
        auto F = std::make_unique<FastTabs>(), Pt = std::make_unique<FastTabs>(), Ct = std::make_unique<FastTabs>();
        SearchTab S6[6];
        const uint32_t pool_limit = test_hook_u32("ent_test_pool", kPoolEntries, kPoolEntries);
        derive_image_tables(P, F.get(), S6, pool_limit, &out->pool_entries_used);
        derive_pair_tables(*F, P.ncomp, Pt.get());
        derive_count_tables(*F, *Pt, Ct.get());
        out->pool_entries = kPoolEntries;
        for (uint32_t slot = 0; slot < 2u * static_cast<uint32_t>(P.ncomp); ++slot)
            for (uint32_t i = 0; i < kLutEntries; ++i) {
                if (F->lut[slot][i] == 0u) ++out->prefixes_left_to_search;
                if ((slot & 1u) && (Pt->lut[slot][i] >> 16) != (Pt->lut[slot][i] & 0xffffu)) ++out->pair_entries;
            }
        EntropyGeom g;
        std::memset(&g, 0, sizeof g);
        g.blocks_per_mcu = P.blocks_per_mcu;
        uint32_t k = 0;
        for (int c = 0; c < P.ncomp; ++c)
            for (uint32_t b = 0; b < static_cast<uint32_t>(P.hs[c]) * P.vs[c]; ++b, ++k) g.kcomp_packed |= static_cast<uint32_t>(c) << (2u * k);
        std::vector<std::vector<uint8_t>> seg_bytes;
        std::vector<uint32_t> seg_mcu0, seg_mcus;
        const uint32_t covered = unstuff_scan(jpeg, len, P, &seg_bytes, &seg_mcu0, &seg_mcus);
        {   // the product's un-stuffer (straight into the staging block) against the walk above: same segments, same bytes, and
            // -- under the sanitizer build, tools/sanitize -- never a byte beyond the bound the staging block is sized by
            const size_t bound = packed_scan_bound(len, P);
            std::unique_ptr<uint8_t[]> packed(new uint8_t[bound]);
            std::vector<uint32_t> p_mcu0, p_mcus, p_nsub, p_first;
            std::vector<uint64_t> p_bits;
            uint64_t p_subs = 0;
            const uint32_t p_covered = unstuff_scan_packed(jpeg, len, P, packed.get(), bound, &p_mcu0, &p_mcus, &p_bits, &p_nsub, &p_first, &p_subs);
            bool same = p_covered == covered && p_mcu0 == seg_mcu0 && p_mcus == seg_mcus && p_bits.size() == seg_bytes.size();
            for (size_t k = 0; same && k < seg_bytes.size(); ++k)
                same = p_bits[k] == seg_bytes[k].size() * 8u &&
                       std::memcmp(packed.get() + static_cast<size_t>(p_first[k]) * 4u, seg_bytes[k].data(), seg_bytes[k].size()) == 0;
            if (!same) return fail(IFHIP_INVALID_STATE, "InvalidState: the packed un-stuffer disagrees with the per-segment walk");
        }
        out->segments = static_cast<uint32_t>(seg_bytes.size());
        out->scan_complete = covered == P.mcus_w * P.mcus_h ? 1u : 0u;
        for (size_t sgi = 0; sgi < seg_bytes.size(); ++sgi) {
            const HostStream in{seg_bytes[sgi]};
            const uint64_t bit_end = static_cast<uint64_t>(seg_bytes[sgi].size()) * 8u, want = static_cast<uint64_t>(seg_mcus[sgi]) * P.blocks_per_mcu;
            const uint64_t n_sub = std::max<uint64_t>(1u, (bit_end + kSubBits - 1u) / kSubBits);
            out->sub_sequences += static_cast<uint32_t>(n_sub);
            HostCounts plain, paired, counted;
            HostState s{0u, 0u, 0u};
            for (uint64_t t = 0; t < n_sub; ++t) {                   // every sub-sequence from its true entry state, three ways
                const uint64_t end = std::min<uint64_t>((t + 1u) * kSubBits, bit_end);
                const HostState a0 = host_walk(g, in, *F, S6, 0, s, end, want, &plain);
                const HostState a1 = host_walk(g, in, *Pt, S6, 1, s, end, want, &paired);
                const HostState a2 = host_walk(g, in, *Ct, S6, 2, s, end, want, &counted);
                if (a1.p != a0.p || a1.c != a0.c || a1.z != a0.z) ++out->pair_walk_mismatches;
                if (a2.p != a0.p || a2.c != a0.c || a2.z != a0.z) ++out->count_walk_mismatches;
                s = a0;
            }
            out->symbols += plain.reads;
            out->table_reads_with_pairs += paired.reads;
            out->blocks += plain.blocks;
            if (plain.blocks != want) ++out->segments_with_wrong_block_count;
            if (plain.bad) ++out->segments_with_invalid_codes;
            for (int c = 0; c < 3; ++c) {
                if (counted.dc[c] != plain.dc[c]) ++out->count_walk_mismatches;
                out->dc_sum[c] += plain.dc[c];                       // (per segment the predictor restarts at 0: sums of the last segment
                if (sgi + 1 == seg_bytes.size()) out->dc_last_segment[c] = plain.dc[c];        // give the last block's DC value)
            }
        }
        return IFHIP_OK;
    } catch (const std::bad_alloc&) {
        return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory");
    } catch (const std::exception& ex) {
        return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what());
    }
}

int ifhip_jpeg_entropy_info(const ifhip_jpeg_entropy* e, uint32_t* width, uint32_t* height, int* n_components, uint8_t* h_samp3,
                            uint8_t* v_samp3, uint32_t* blocks_w3, uint32_t* blocks_h3, uint32_t* n_subsequences, uint32_t* n_segments) {
    if (!e) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null handle");
    const ParsedJpeg& F = e->first;
    if (width) *width = F.width;
    if (height) *height = F.height;
    if (n_components) *n_components = F.ncomp;
    for (int c = 0; c < 3; ++c) {
        if (h_samp3) h_samp3[c] = c < F.ncomp ? F.hs[c] : 0;
        if (v_samp3) v_samp3[c] = c < F.ncomp ? F.vs[c] : 0;
        if (blocks_w3) blocks_w3[c] = c < F.ncomp ? F.bw[c] : 0;
        if (blocks_h3) blocks_h3[c] = c < F.ncomp ? F.bh[c] : 0;
    }
    if (n_subsequences) *n_subsequences = e->a.n_sub;
    if (n_segments) *n_segments = e->a.n_seg;
    return IFHIP_OK;
}

int ifhip_jpeg_entropy_quant_tables(const ifhip_jpeg_entropy* e, uint16_t* qt_n3x64) {
    if (!e || !qt_n3x64) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    std::memcpy(qt_n3x64, e->qt.data(), e->qt.size() * sizeof(uint16_t));
    return IFHIP_OK;
}

// Decodes the whole batch into d_coef* (each [n_images][bh_c][bw_c][64] int16, natural order).  The fixpoint check needs
// its answer on the host, so the call synchronises the stream; *rounds (optional) reports how many synchronisation
// rounds ran (1 unless a correction crossed a workgroup boundary).
int ifhip_jpeg_entropy_decode_device(ifhip_jpeg_entropy* e, int16_t* d_coef0, int16_t* d_coef1, int16_t* d_coef2,
                                     uint32_t* rounds, void* hip_stream) {
    if (!e) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null handle");
    const ParsedJpeg& F = e->first;
    if (!d_coef0 || (F.ncomp == 3 && (!d_coef1 || !d_coef2))) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null coefficient plane");
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != e->device) return fail(IFHIP_INVALID_STATE, "InvalidState: handle belongs to device %d, current device is %d", e->device, dev);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    EntropyArgs a = e->a;
    a.coef[0] = d_coef0; a.coef[1] = d_coef1; a.coef[2] = d_coef2;
    if ((reinterpret_cast<uintptr_t>(d_coef0) | reinterpret_cast<uintptr_t>(d_coef1) | reinterpret_cast<uintptr_t>(d_coef2)) & 15u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient planes must be 16-byte aligned");
    const dim3 sync_grid((a.n_sub + kSyncLanes - 1u) / kSyncLanes), sync_block(kSyncLanes);
    const dim3 round_grid((a.n_sub + kOwnSubs - 1u) / kOwnSubs);
    const uint32_t max_rounds = a.n_sub + 2u;
    // Round 0 (speculative decode with the fixpoint iteration inside every workgroup; it also clears the flags), the
    // count pass (which checks that every sub-sequence was decoded from its predecessor's exit: with the warm-up lanes of
    // round 0 no correction crosses a workgroup boundary) and the write pass are enqueued back to back -- three launches,
    // every launch boundary costs ~8 us -- and the host looks at the flags once.  If the count pass found an unsettled
    // sub-sequence, further rounds run (one look per round) and count + write are repeated: the write pass stores every
    // block in full, so the repeat simply overwrites.
    uint32_t r = 0;
    a.round = 0u;
    hipLaunchKernelGGL(entropy_round_kernel, round_grid, sync_block, 0, st, a);
    HIP_TRY(hipGetLastError());
    for (;;) {
        a.round = r;
        hipLaunchKernelGGL(entropy_count_kernel, sync_grid, sync_block, 0, st, a);
        HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(entropy_write_kernel, dim3((a.n_sub + kWriteLanes - 1u) / kWriteLanes), dim3(kWriteLanes), 0, st, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(e->h_flags, a.changed, kFlagWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(static_cast<hipError_t>(ifhip::wait_stream(st)));
        if (e->h_flags[17] == 0u) break;                             // every sub-sequence starts where its predecessor ended
        for (;;) {                                                   // rare: corrections across workgroups
            ++r;
            if (r >= max_rounds) return fail(IFHIP_INVALID_STATE, "InvalidState: entropy decode did not converge");
            HIP_TRY(hipMemsetAsync(a.changed + (r & 15u), 0, sizeof(uint32_t), st));
            a.round = r;
            hipLaunchKernelGGL(entropy_round_kernel, round_grid, sync_block, 0, st, a);
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(e->h_flags, a.changed, kFlagWords * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
            HIP_TRY(static_cast<hipError_t>(ifhip::wait_stream(st)));
            if (e->h_flags[r & 15u] == 0u) break;
        }
        HIP_TRY(hipMemsetAsync(a.errors, 0, 2u * sizeof(uint32_t), st));     // flags of the discarded count and write passes
    }
    if (rounds) *rounds = r + 1u;
    if (e->h_flags[16]) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data (flags 0x%x)", e->h_flags[16]);
    return IFHIP_OK;
}

}  // extern "C"
