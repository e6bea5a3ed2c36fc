// jpeg_read.cpp -- host-side entropy decoding of the Huffman JPEGs the GPU entropy stage does not take: progressive (SOF2)
// files -- what the reference's own `mozjpeg` encoder preset writes (codecs/mozjpeg.rs:121-123) and MozJpegDecoder reads back
// (codecs/mozjpeg_decoder.rs:295-420 -> libjpeg jdphuff.c) -- and sequential files with several or non-interleaved scans.
// Output = the quantised coefficient planes of the pixel stage ([blocks_h][blocks_w][64] int16, natural order, MCU-padded:
// the layout of jpeg_read_coefficients and of csrc/jpeg_entropy.hip), so everything behind the entropy stage -- IDCT,
// up-sampling, colour, the fused resample -- is the GPU path unchanged.  The scans are serial bit streams with state that
// spans the whole image (successive approximation refines coefficients of earlier scans): host work, like the reference's.
//
// Follows libjpeg's published decoder (jdmarker.c marker syntax, jdhuff.c decode_mcu, jdphuff.c decode_mcu_DC_first /
// _AC_first / _DC_refine / _AC_refine, jdinput.c per-scan geometry).  Pinned by tests/test_jpeg_progressive.py: files this
// library's own progressive WRITER produced from known coefficients (byte-identical to libjpeg-turbo's) decode to
// exactly those coefficients, and Pillow-written progressive files decode, through the GPU pixel stage, to Pillow's pixels.
#include <cstdint>
#include <cstring>
#include <memory>
#include <vector>

#include "common.hpp"

namespace ifhip {
namespace {

const uint8_t kZig[64] = {0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48,
                          41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56, 57, 50, 43, 36, 29, 22,
                          15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63};

struct HuffTab {                       // jdhuff.c jpeg_make_d_derived_tbl: canonical codes by length
    bool present = false;
    uint8_t bits[17] = {0};
    uint8_t vals[256] = {0};
    int32_t maxcode[18];               // largest code of length l, -1 if none
    int32_t valoff[17];                // huffval index of the first code of length l minus that code
    uint8_t look_len[256], look_sym[256];      // 8-bit lookahead: code length (0: longer than 8) and symbol
    void derive() {
        int32_t code = 0;
        int k = 0;
        std::memset(look_len, 0, sizeof look_len);
        for (int l = 1; l <= 16; ++l) {
            valoff[l] = k - code;
            for (int i = 0; i < bits[l]; ++i, ++k, ++code)
                if (l <= 8) {
                    const int first = code << (8 - l);
                    for (int f = 0; f < (1 << (8 - l)); ++f)
                        if (first + f < 256) { look_len[first + f] = static_cast<uint8_t>(l); look_sym[first + f] = vals[k & 255]; }
                }
            maxcode[l] = bits[l] ? code - 1 : -1;
            code <<= 1;
        }
        maxcode[17] = 0x7fffffff;
    }
};

struct Frame {
    uint32_t width = 0, height = 0;
    int ncomp = 0;
    bool progressive = false;
    uint8_t comp_id[3] = {0, 0, 0}, hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1}, tq[3] = {0, 0, 0};
    uint32_t hmax = 1, vmax = 1, mcus_w = 0, mcus_h = 0;
    uint32_t bw[3] = {0, 0, 0}, bh[3] = {0, 0, 0};            // MCU-padded plane sizes in blocks
    uint32_t wb[3] = {0, 0, 0}, hb[3] = {0, 0, 0};            // jdinput.c width_in_blocks / height_in_blocks (what a non-interleaved scan covers)
    uint16_t qt[4][64];
    bool qt_present[4] = {false, false, false, false};
    uint16_t qt_latched[3][64];                               // jdinput.c latch_quant_tables: a component's table as it stood at its FIRST scan
    bool latched[3] = {false, false, false};
    int adobe_transform = -1;
};

struct BitReader {                     // jdhuff.c fill_bit_buffer: bytes with FF00 un-stuffed, zeros behind a marker
    const uint8_t* d;
    size_t len, pos;
    uint64_t acc = 0;
    int n = 0;
    bool hit_marker = false;
    void fill() {
        while (n <= 56) {
            uint32_t b = 0;
            if (!hit_marker && pos < len) {
                b = d[pos];
                if (b == 0xFF) {
                    size_t q = pos + 1;
                    while (q < len && d[q] == 0xFF) ++q;                 // fill bytes
                    if (q < len && d[q] == 0x00) { pos = q + 1; }
                    else { hit_marker = true; b = 0; }                    // a marker: the scan's data ends here (pos stays on the FF)
                } else ++pos;
            }
            acc |= static_cast<uint64_t>(b) << (56 - n);
            n += 8;
        }
    }
    uint32_t peek(int k) { if (n < k) fill(); return static_cast<uint32_t>(acc >> (64 - k)); }
    void drop(int k) { acc <<= k; n -= k; }
    uint32_t get(int k) { if (k == 0) return 0; const uint32_t v = peek(k); drop(k); return v; }
    void align_to_marker() { acc = 0; n = 0; hit_marker = false; }
};

inline int32_t extend(uint32_t r, int s) { return r < (1u << (s - 1)) ? static_cast<int32_t>(r) - ((1 << s) - 1) : static_cast<int32_t>(r); }

int decode_symbol(BitReader& br, const HuffTab& t, int* sym) {
    const uint32_t look = br.peek(8);
    if (t.look_len[look]) { br.drop(t.look_len[look]); *sym = t.look_sym[look]; return 0; }
    int l = 9;
    int32_t code = static_cast<int32_t>(br.peek(9));
    while (l <= 16 && code > t.maxcode[l]) { ++l; code = static_cast<int32_t>(br.peek(l)); }
    if (l > 16) return -1;                                                // no such code (jdhuff.c JWRN_HUFF_BAD_CODE)
    br.drop(l);
    *sym = t.vals[(code + t.valoff[l]) & 255];
    return 0;
}

struct ScanComp { int ci, td, ta; };

}  // namespace

// width_in_blocks etc. of a frame (jdinput.c initial_setup / per_scan_setup)
static int finish_frame(Frame& F) {
    if (F.ncomp != 1 && F.ncomp != 3) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-component JPEG", F.ncomp);
    F.hmax = F.vmax = 1;
    for (int c = 0; c < F.ncomp; ++c) {
        if (F.hs[c] < 1 || F.hs[c] > 2 || F.vs[c] < 1 || F.vs[c] > 2)
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: sampling factor %dx%d", F.hs[c], F.vs[c]);
        F.hmax = std::max<uint32_t>(F.hmax, F.hs[c]); F.vmax = std::max<uint32_t>(F.vmax, F.vs[c]);
    }
    if (F.ncomp == 1) { F.hs[0] = F.vs[0] = 1; F.hmax = F.vmax = 1; }
    if (F.ncomp == 3 && (F.hs[0] != F.hmax || F.vs[0] != F.vmax || F.hs[1] != 1 || F.vs[1] != 1 || F.hs[2] != 1 || F.vs[2] != 1))
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: sampling %dx%d,%dx%d,%dx%d (chroma must be 1x1)", F.hs[0], F.vs[0],
                    F.hs[1], F.vs[1], F.hs[2], F.vs[2]);
    if (F.ncomp == 3 && (F.adobe_transform == 0 || (F.adobe_transform < 0 && F.comp_id[0] == 'R' && F.comp_id[1] == 'G' && F.comp_id[2] == 'B')))
        return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: RGB-coded JPEG (no YCbCr transform)");
    F.mcus_w = (F.width + 8 * F.hmax - 1) / (8 * F.hmax);
    F.mcus_h = (F.height + 8 * F.vmax - 1) / (8 * F.vmax);
    for (int c = 0; c < F.ncomp; ++c) {
        F.bw[c] = F.mcus_w * F.hs[c]; F.bh[c] = F.mcus_h * F.vs[c];
        F.wb[c] = (static_cast<uint64_t>(F.width) * F.hs[c] + 8ull * F.hmax - 1) / (8ull * F.hmax);
        F.hb[c] = (static_cast<uint64_t>(F.height) * F.vs[c] + 8ull * F.vmax - 1) / (8ull * F.vmax);
    }
    return IFHIP_OK;
}

// One pass over the markers.  coef == nullptr: stop at the first SOS (frame facts only).
static int read_jpeg(const uint8_t* d, size_t len, Frame* out, int16_t* const coef[3]) {
    Frame& F = *out;
    if (!d || len < 4 || d[0] != 0xFF || d[1] != 0xD8) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: not a JPEG (no SOI)");
    HuffTab dc[4], ac[4];
    uint32_t restart_interval = 0;
    bool have_sof = false, any_scan = false;
    // (jdphuff.c's coef_bits bookkeeping only WARNS about scripts that send bits out of order -- JWRN_BOGUS_PROGRESSION -- and
    // decodes them anyway; there is no warning channel here, so such scans are decoded as libjpeg decodes them)
    size_t i = 2;
    while (i + 4 <= len) {
        if (d[i] != 0xFF) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: marker expected at byte %zu", i);
        while (i < len && d[i] == 0xFF) ++i;
        if (i >= len) break;
        const uint8_t m = d[i++];
        if (m == 0xD8 || (m >= 0xD0 && m <= 0xD7) || m == 0x01) continue;
        if (m == 0xD9) break;
        if (i + 2 > len) break;
        const size_t seg = (static_cast<size_t>(d[i]) << 8) | d[i + 1];
        if (seg < 2 || i + seg > len) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: truncated marker segment FF%02X", m);
        const uint8_t* q = d + i + 2;
        const size_t n = seg - 2;
        if (m == 0xDB) {
            size_t k = 0;
            while (k < n) {
                const int pq = q[k] >> 4, t = q[k] & 15;
                ++k;
                if (t > 3 || k + (pq ? 128u : 64u) > n) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DQT");
                for (int z = 0; z < 64; ++z) {
                    F.qt[t][kZig[z]] = pq ? static_cast<uint16_t>((q[k] << 8) | q[k + 1]) : q[k];
                    k += pq ? 2 : 1;
                }
                F.qt_present[t] = true;
            }
        } else if (m == 0xC4) {
            size_t k = 0;
            while (k + 17 <= n) {
                const int tc = q[k] >> 4, th = q[k] & 15;
                if (tc > 1 || th > 3) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DHT");
                HuffTab& h = tc ? ac[th] : dc[th];
                size_t total = 0;
                h.bits[0] = 0;
                for (int l = 1; l <= 16; ++l) { h.bits[l] = q[k + l]; total += q[k + l]; }
                k += 17;
                if (total > 256 || k + total > n) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad DHT");
                std::memset(h.vals, 0, sizeof h.vals);
                std::memcpy(h.vals, q + k, total);
                k += total;
                h.present = true;
                h.derive();
            }
        } else if (m == 0xC0 || m == 0xC1 || m == 0xC2) {
            if (have_sof) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: two frame headers");
            if (n < 6 || q[0] != 8) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-bit JPEG", n ? q[0] : 0);
            F.progressive = m == 0xC2;
            F.height = (q[1] << 8) | q[2];
            F.width = (q[3] << 8) | q[4];
            F.ncomp = q[5];
            if (F.ncomp != 1 && F.ncomp != 3) return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: %d-component JPEG", F.ncomp);
            if (n < 6u + 3u * F.ncomp || F.width == 0 || F.height == 0) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOF");
            for (int c = 0; c < F.ncomp; ++c) { F.comp_id[c] = q[6 + 3 * c]; F.hs[c] = q[7 + 3 * c] >> 4; F.vs[c] = q[7 + 3 * c] & 15; F.tq[c] = q[8 + 3 * c] & 3; }
            if (int rc = finish_frame(F)) return rc;
            have_sof = true;
        } else if (m >= 0xC3 && m <= 0xCF && m != 0xC4 && m != 0xC8 && m != 0xCC) {
            return fail(IFHIP_METHOD_NOT_IMPLEMENTED, "MethodNotImplemented: JPEG process SOF%d (Huffman sequential and progressive only)", m - 0xC0);
        } else if (m == 0xDD) {
            if (n >= 2) restart_interval = (q[0] << 8) | q[1];
        } else if (m == 0xEE && n >= 12 && std::memcmp(q, "Adobe", 5) == 0) {
            F.adobe_transform = q[11];
            if (have_sof) if (int rc = finish_frame(F)) return rc;
        } else if (m == 0xDA) {
            if (!have_sof) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: SOS before SOF");
            if (!coef) return IFHIP_OK;                                   // frame facts only
            if (n < 1) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOS");
            const int ns = q[0];
            if (ns < 1 || ns > F.ncomp || n < 4u + 2u * ns) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOS");
            ScanComp sc[3];
            for (int s = 0; s < ns; ++s) {
                int c = -1;
                for (int k = 0; k < F.ncomp; ++k) if (F.comp_id[k] == q[1 + 2 * s]) c = k;
                if (c < 0) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: scan names an unknown component");
                sc[s] = ScanComp{c, q[2 + 2 * s] >> 4, q[2 + 2 * s] & 15};
                if (sc[s].td > 3 || sc[s].ta > 3) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad SOS");
            }
            const int Ss = q[1 + 2 * ns], Se = q[2 + 2 * ns], Ah = q[3 + 2 * ns] >> 4, Al = q[3 + 2 * ns] & 15;
            if (F.progressive) {                                          // jdphuff.c start_pass_phuff_decoder's checks
                const bool dc_scan = Ss == 0;
                if ((dc_scan && Se != 0) || (!dc_scan && (Se < Ss || Se > 63 || ns != 1)) || Al > 13 || (Ah != 0 && Ah - 1 != Al))
                    return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad progressive scan parameters Ss=%d Se=%d Ah=%d Al=%d", Ss, Se, Ah, Al);
            } else if (Ss != 0 || Se != 63 || Ah != 0 || Al != 0) {
                return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: bad sequential scan parameters");
            }
            for (int s = 0; s < ns; ++s) {                                // jdinput.c latch_quant_tables (start of every scan's input pass)
                const int c = sc[s].ci;
                if (F.latched[c]) continue;                               // a DQT that redefines the table later does not reach this component
                if (!F.qt_present[F.tq[c]]) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: missing quantisation table");
                std::memcpy(F.qt_latched[c], F.qt[F.tq[c]], 128);
                F.latched[c] = true;
            }
            for (int s = 0; s < ns; ++s) {
                const bool need_dc = !F.progressive || (Ss == 0 && Ah == 0), need_ac = !F.progressive || Ss != 0;
                if ((need_dc && !dc[sc[s].td].present) || (need_ac && !ac[sc[s].ta].present))
                    return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: missing Huffman table");
            }
            // per-scan geometry (jdinput.c per_scan_setup): one component -> its own block raster; several -> MCUs
            const bool interleaved = ns > 1;
            const uint32_t mcus_w = interleaved ? F.mcus_w : F.wb[sc[0].ci], mcus_h = interleaved ? F.mcus_h : F.hb[sc[0].ci];
            BitReader br{d, len, i + seg};
            int32_t last_dc[3] = {0, 0, 0};
            uint32_t eobrun = 0, restarts_left = restart_interval;
            const uint64_t total_mcus = static_cast<uint64_t>(mcus_w) * mcus_h;
            for (uint64_t mi = 0; mi < total_mcus; ++mi) {
                if (restart_interval && restarts_left == 0) {            // process_restart: the marker, then fresh state
                    size_t p = br.pos;
                    br.align_to_marker();
                    while (p + 1 < len && !(d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7)) {
                        if (d[p] == 0xFF && d[p + 1] != 0x00 && d[p + 1] != 0xFF) break;     // another marker: the scan ended early
                        ++p;
                    }
                    if (p + 1 < len && d[p] == 0xFF && d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7) p += 2;
                    br.pos = p;
                    last_dc[0] = last_dc[1] = last_dc[2] = 0;
                    eobrun = 0;
                    restarts_left = restart_interval;
                }
                const uint32_t my = static_cast<uint32_t>(mi / mcus_w), mx = static_cast<uint32_t>(mi % mcus_w);
                for (int s = 0; s < ns; ++s) {
                    const int ci = sc[s].ci;
                    const uint32_t nh = interleaved ? F.hs[ci] : 1u, nv = interleaved ? F.vs[ci] : 1u;
                    for (uint32_t dy = 0; dy < nv; ++dy)
                        for (uint32_t dx = 0; dx < nh; ++dx) {
                            const uint32_t bx = mx * nh + dx, by = my * nv + dy;
                            int16_t* blk = coef[ci] + (static_cast<size_t>(by) * F.bw[ci] + bx) * 64u;
                            int sym = 0;
                            if (!F.progressive) {                         // jdhuff.c decode_mcu
                                if (decode_symbol(br, dc[sc[s].td], &sym)) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data");
                                int32_t diff = 0;
                                if (sym) { if (sym > 15) sym = 15; diff = extend(br.get(sym), sym); }
                                last_dc[ci] += diff;
                                blk[0] = static_cast<int16_t>(last_dc[ci]);
                                for (int k = 1; k < 64;) {
                                    if (decode_symbol(br, ac[sc[s].ta], &sym)) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data");
                                    const int r = sym >> 4, sz = sym & 15;
                                    if (sz) {
                                        k += r;
                                        const int32_t v = extend(br.get(sz), sz);
                                        blk[kZig[k > 63 ? 63 : k]] = static_cast<int16_t>(v);   // (jpeg_natural_order's 16 extra entries: a corrupt run lands on 63)
                                        ++k;
                                    } else { if (r != 15) break; k += 16; }
                                }
                            } else if (Ss == 0 && Ah == 0) {              // decode_mcu_DC_first
                                if (decode_symbol(br, dc[sc[s].td], &sym)) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data");
                                int32_t diff = 0;
                                if (sym) { if (sym > 15) sym = 15; diff = extend(br.get(sym), sym); }
                                last_dc[ci] += diff;
                                blk[0] = static_cast<int16_t>(static_cast<uint32_t>(last_dc[ci]) << Al);
                            } else if (Ss == 0) {                         // decode_mcu_DC_refine
                                if (br.get(1)) blk[0] = static_cast<int16_t>(blk[0] | (1 << Al));
                            } else if (Ah == 0) {                         // decode_mcu_AC_first
                                if (eobrun > 0) { --eobrun; continue; }
                                for (int k = Ss; k <= Se; ++k) {
                                    if (decode_symbol(br, ac[sc[s].ta], &sym)) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data");
                                    const int r = sym >> 4, sz = sym & 15;
                                    if (sz) {
                                        k += r;
                                        const int32_t v = extend(br.get(sz), sz);
                                        blk[kZig[k > 63 ? 63 : k]] = static_cast<int16_t>(static_cast<uint32_t>(v) << Al);   // (likewise; jdphuff.c does not clamp to Se either)
                                    } else if (r == 15) k += 15;
                                    else {
                                        eobrun = 1u << r;
                                        if (r) eobrun += br.get(r);
                                        --eobrun;
                                        break;
                                    }
                                }
                            } else {                                      // decode_mcu_AC_refine
                                const int32_t p1 = 1 << Al, m1 = -(1 << Al);
                                int k = Ss;
                                auto correct = [&](int16_t* c) {          // one correction bit for a coefficient that is already nonzero
                                    if (br.get(1) && (*c & p1) == 0) *c = static_cast<int16_t>(*c >= 0 ? *c + p1 : *c + m1);
                                };
                                if (eobrun == 0) {
                                    for (; k <= Se; ++k) {
                                        if (decode_symbol(br, ac[sc[s].ta], &sym)) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: corrupt entropy-coded data");
                                        int r = sym >> 4;
                                        int32_t v = 0;
                                        if (sym & 15) v = br.get(1) ? p1 : m1;     // (size must be 1: a newly nonzero coefficient)
                                        else if (r != 15) {
                                            eobrun = 1u << r;
                                            if (r) eobrun += br.get(r);
                                            break;                        // the rest of the block: the end-of-band logic below
                                        }
                                        do {                              // over nonzero history and r zeros, correction bits on the way
                                            int16_t* c = blk + kZig[k];
                                            if (*c != 0) correct(c);
                                            else if (--r < 0) break;
                                            ++k;
                                        } while (k <= Se);
                                        if (v && k <= 63) blk[kZig[k]] = static_cast<int16_t>(v);
                                    }
                                }
                                if (eobrun > 0) {
                                    for (; k <= Se; ++k) { int16_t* c = blk + kZig[k]; if (*c != 0) correct(c); }
                                    --eobrun;
                                }
                            }
                        }
                }
                if (restart_interval) --restarts_left;
            }
            any_scan = true;
            // behind the scan: skip to the next marker (pad bits, stuffing)
            size_t p = br.pos;
            while (p + 1 < len && !(d[p] == 0xFF && d[p + 1] != 0x00 && d[p + 1] != 0xFF && !(d[p + 1] >= 0xD0 && d[p + 1] <= 0xD7))) ++p;
            i = p;
            continue;
        }
        i += seg;
    }
    if (!have_sof) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: no frame found");
    if (coef && !any_scan) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: no scan found");
    return IFHIP_OK;
}

}  // namespace ifhip

using namespace ifhip;

extern "C" {

int ifhip_jpeg_frame_info(const uint8_t* jpeg, size_t len, uint32_t* width, uint32_t* height, int* n_components, uint8_t* h_samp3,
                          uint8_t* v_samp3, uint32_t* blocks_w3, uint32_t* blocks_h3, uint16_t* qt3x64, int* progressive) {
    try {
        Frame F;
        if (int rc = read_jpeg(jpeg, len, &F, nullptr)) return rc;
        if (width) *width = F.width;
        if (height) *height = F.height;
        if (n_components) *n_components = F.ncomp;
        if (progressive) *progressive = F.progressive ? 1 : 0;
        for (int c = 0; c < 3; ++c) {
            if (h_samp3) h_samp3[c] = c < F.ncomp ? F.hs[c] : 0;
            if (v_samp3) v_samp3[c] = c < F.ncomp ? F.vs[c] : 0;
            if (blocks_w3) blocks_w3[c] = c < F.ncomp ? F.bw[c] : 0;
            if (blocks_h3) blocks_h3[c] = c < F.ncomp ? F.bh[c] : 0;
            if (qt3x64) {
                if (c < F.ncomp && F.qt_present[F.tq[c]]) std::memcpy(qt3x64 + 64 * c, F.qt[F.tq[c]], 128);
                else std::memset(qt3x64 + 64 * c, 0, 128);
            }
        }
        return IFHIP_OK;
    } catch (const std::exception& ex) { return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what()); }
}

int ifhip_jpeg_read_coefficients_host(const uint8_t* jpeg, size_t len, int16_t* coef0, int16_t* coef1, int16_t* coef2, uint16_t* qt3x64) {
    try {
        Frame F;
        if (int rc = read_jpeg(jpeg, len, &F, nullptr)) return rc;       // geometry first: the planes are cleared to it
        int16_t* coef[3] = {coef0, coef1, coef2};
        for (int c = 0; c < F.ncomp; ++c) {
            if (!coef[c]) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null coefficient plane");
            std::memset(coef[c], 0, static_cast<size_t>(F.bw[c]) * F.bh[c] * 128u);
        }
        Frame G;
        if (int rc = read_jpeg(jpeg, len, &G, coef)) return rc;
        // (quantisation tables may arrive between scans: a component's table is the one in force at its first scan, as libjpeg
        // latches it; a later DQT with the same id only reaches components whose first scan is still to come)
        for (int c = 0; c < G.ncomp; ++c) {
            if (!G.latched[c]) return fail(IFHIP_INVALID_ARGUMENT, "ImageMalformed: component %d has no scan", c);
            if (qt3x64) std::memcpy(qt3x64 + 64 * c, G.qt_latched[c], 128);
        }
        return IFHIP_OK;
    } catch (const std::bad_alloc&) { return fail(IFHIP_ALLOCATION_FAILED, "AllocationFailed: host memory");
    } catch (const std::exception& ex) { return fail(IFHIP_INVALID_STATE, "InvalidState: %s", ex.what()); }
}

}  // extern "C"
