// enc_emulate.cpp -- test infrastructure, not part of the product: the device entropy coder's passes (csrc/jpeg_encode.hip)
// run lane by lane on the CPU with the SAME block routine, scan order and placement rules (csrc/jpeg_encode_core.hpp), so
// that the algorithm is checked against the host writer (and through it libjpeg-turbo) without a GPU.  The write pass
// visits the blocks in a scrambled order and checks the ownership rule of the word stream: a word a lane stores plainly is
// touched by nobody else.  Built by tests/test_jpeg_device_coder.py with g++.
#include <cstdint>
#include <algorithm>
#include <cstring>
#include <utility>
#include <vector>

#include "../imageflow_amd/csrc/jpeg_encode_core.hpp"

using namespace ifhip;

namespace {
uint32_t* g_words = nullptr;
std::vector<uint8_t>* g_mark = nullptr;      // 1: stored plainly (owned), 2: ORed (shared)
int g_violations = 0;

struct HostStore {
    static void shared(uint32_t* p, uint32_t v) {
        uint8_t& m = (*g_mark)[static_cast<size_t>(p - g_words)];
        if (m == 1) ++g_violations;
        m = 2;
        *p |= v;
    }
    static void owned(uint32_t* p, uint32_t v) {
        uint8_t& m = (*g_mark)[static_cast<size_t>(p - g_words)];
        if (m != 0 || *p != 0) ++g_violations;
        m = 1;
        *p = v;
    }
};

struct PlaneCoef {                            // a block of the plane (natural order) seen as the kernels stage it
    const int16_t* b;
    int32_t operator()(int k) const { return b[enc_zigzag(k)]; }
    uint32_t pair(int j) const {
        const uint16_t lo = static_cast<uint16_t>(b[enc_zigzag(static_cast<int>(enc_position_of_slot(2u * j)))]);
        const uint16_t hi = static_cast<uint16_t>(b[enc_zigzag(static_cast<int>(enc_position_of_slot(2u * j + 1u)))]);
        return static_cast<uint32_t>(lo) | static_cast<uint32_t>(hi) << 16;
    }
};
}  // namespace

extern "C" int enc_emulate(const int16_t* c0, const int16_t* c1, const int16_t* c2, uint32_t width, uint32_t height, int ncomp,
                           const uint8_t* hs, const uint8_t* vs, const uint32_t* bw, const uint32_t* bh, const uint8_t* header,
                           uint32_t header_len, const uint32_t* tabs, uint8_t* out, size_t capacity, size_t* len, uint32_t* status,
                           int* violations, int* window_paths) {
    EncGeom g;
    if (enc_make_geom(width, height, ncomp, hs, vs, bw, bh, &g)) return 1;
    const int16_t* planes[3] = {c0, c1, c2};
    auto block = [&](uint32_t s) { const EncBlockRef r = enc_locate(g, s); return planes[r.comp] + static_cast<size_t>(r.offset) * 64u; };
    auto pred_of = [&](uint32_t s) -> int32_t {
        const EncBlockRef r = enc_locate(g, s);
        return r.pred_offset == 0xFFFFFFFFu ? 0 : planes[r.comp][static_cast<size_t>(r.pred_offset) * 64u];
    };
    auto tab_of = [&](uint32_t s) { return tabs + (enc_locate(g, s).comp ? 512u : 0u); };
    // count pass + per-workgroup sums
    const uint32_t n_wg = (g.nblocks + kEncBlocksPerWg - 1u) / kEncBlocksPerWg;
    std::vector<uint16_t> nbits(g.nblocks);
    std::vector<uint32_t> wg(n_wg, 0);
    uint32_t st = 0;
    for (uint32_t s = 0; s < g.nblocks; ++s) {
        EncCountSink sink;
        const uint32_t* t = tab_of(s);
        if (enc_block(PlaneCoef{block(s)}, pred_of(s), t, t + 256, sink)) st |= kEncBadCoef;
        nbits[s] = static_cast<uint16_t>(sink.bits);
        wg[s / kEncBlocksPerWg] += sink.bits;
    }
    // scan
    uint32_t tot_bits = 0;
    for (uint32_t i = 0; i < n_wg; ++i) { const uint32_t v = wg[i]; wg[i] = tot_bits; tot_bits += v; }
    *status = st;
    *violations = 0;
    if (st) { *len = 0; return 0; }
    const uint32_t bytes = (tot_bits + 7u) >> 3;
    const size_t cap_bytes = (static_cast<size_t>(bytes) + kEncChunkBytes - 1u) / kEncChunkBytes * kEncChunkBytes + kEncChunkBytes;
    std::vector<uint32_t> words(cap_bytes / 4u, 0u);
    std::vector<uint8_t> mark(words.size(), 0);
    g_words = words.data(); g_mark = &mark; g_violations = 0;
    // write pass: workgroups and the blocks inside each in a scrambled order (any order must do).  A workgroup whose piece of
    // the stream fits the window assembles it there (window words follow the same ownership rule) and copies it out: first
    // and last word ORed into the stream, the words between stored; a denser piece goes to the stream word by word.
    uint64_t rng = 0x9E3779B97F4A7C15ull;
    auto scramble = [&](std::vector<uint32_t>& v) {
        for (size_t i = v.size(); i > 1; --i) {
            rng = rng * 6364136223846793005ull + 1442695040888963407ull;
            std::swap(v[i - 1], v[static_cast<size_t>((rng >> 33) % i)]);
        }
    };
    std::vector<uint32_t> wgs(n_wg);
    for (uint32_t i = 0; i < n_wg; ++i) wgs[i] = i;
    scramble(wgs);
    int window_wgs = 0;
    for (uint32_t w : wgs) {
        const uint32_t s0 = w * kEncBlocksPerWg, s1 = std::min(g.nblocks, s0 + kEncBlocksPerWg);
        uint32_t total = 0;
        for (uint32_t q = s0; q < s1; ++q) total += nbits[q];
        if (s1 == g.nblocks) total += enc_final_padding(wg[w] + total);
        const uint32_t n_words = enc_window_words(wg[w], total);
        const bool windowed = n_words <= kEncWindowWords;
        std::vector<uint32_t> win(windowed ? n_words + 1u : 0u, 0u);              // (one spare word: EncWindowSink)
        std::vector<uint8_t> win_mark(win.size(), 0);
        if (windowed) { g_words = win.data(); g_mark = &win_mark; ++window_wgs; }
        else { g_words = words.data(); g_mark = &mark; }
        std::vector<uint32_t> order(s1 - s0);
        for (uint32_t i = 0; i < s1 - s0; ++i) order[i] = s0 + i;
        scramble(order);
        for (uint32_t s : order) {
            uint32_t off = windowed ? (wg[w] & 31u) : wg[w];
            for (uint32_t q = s0; q < s; ++q) off += nbits[q];
            const uint32_t* t = tab_of(s);
            auto code = [&](auto& sink) {
                enc_block(PlaneCoef{block(s)}, pred_of(s), t, t + 256, sink);
                if (s == g.nblocks - 1u) {
                    const uint32_t pad = (8u - sink.bits_in_last_byte()) & 7u;
                    if (pad) sink.put((1u << pad) - 1u, pad);
                }
                sink.finish();
            };
            if (windowed) { EncWindowSink<HostStore> sink(g_words, off); code(sink); }
            else { EncWordSink<HostStore> sink(g_words, off); code(sink); }
        }
        if (windowed) {
            if (win[n_words] != 0u) ++g_violations;                                 // the spare word only ever takes zeros
            g_words = words.data(); g_mark = &mark;
            uint32_t* dst = words.data() + (wg[w] >> 5);
            for (uint32_t i = 0; i < n_words; ++i) {
                if (i == 0 || i == n_words - 1u) HostStore::shared(dst + i, __builtin_bswap32(win[i]));
                else HostStore::owned(dst + i, __builtin_bswap32(win[i]));
            }
        }
    }
    if (window_paths) *window_paths = window_wgs;
    *violations = g_violations;
    // 0xFF counts per chunk, scan, stuffed bytes
    const uint32_t chunks = (bytes + kEncChunkBytes - 1u) / kEncChunkBytes;
    std::vector<uint32_t> ff(chunks, 0);
    for (uint32_t c = 0; c < chunks; ++c)
        for (uint32_t at = c * kEncChunkBytes; at < (c + 1u) * kEncChunkBytes && at < bytes; at += 4u) ff[c] += enc_count_ff(words[at >> 2]);
    uint32_t tot_ff = 0;
    for (uint32_t c = 0; c < chunks; ++c) { const uint32_t v = ff[c]; ff[c] = tot_ff; tot_ff += v; }
    const size_t file_len = static_cast<size_t>(header_len) + bytes + tot_ff + 2u;
    *len = file_len;
    if (file_len > capacity) { *status = kEncFileOverflow; *len = 0; return 0; }
    std::memcpy(out, header, header_len);
    const uint8_t* stream = reinterpret_cast<const uint8_t*>(words.data());
    for (uint32_t c = 0; c < chunks; ++c) {
        uint32_t lane_ff = 0;                           // exclusive count inside the chunk, 16 bytes per lane
        for (uint32_t at = c * kEncChunkBytes; at < (c + 1u) * kEncChunkBytes && at < bytes; at += 16u) {
            uint8_t* d = out + header_len + at + ff[c] + lane_ff;
            for (uint32_t j = 0; j < 16u && at + j < bytes; ++j) {
                const uint8_t b = stream[at + j];
                *d++ = b;
                if (b == 255u) { *d++ = 0; ++lane_ff; }
            }
        }
    }
    out[file_len - 2u] = 0xFF; out[file_len - 1u] = 0xD9;
    return 0;
}
