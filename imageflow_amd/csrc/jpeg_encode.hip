// jpeg_encode.hip -- gfx950 kernels + C ABI of the device entropy coder: quantised coefficient planes in HBM (what
// ifhip_jpeg_forward_batch_device leaves) -> complete baseline JPEG files in HBM, byte-identical to the host writer
// (csrc/jpeg_write.cpp ifhip_jpeg_write_baseline) and so to libjpeg-turbo for the same pixels.
//
// Replaces, for the classic preset's default (codecs/mozjpeg.rs:108-129 with neither progressive nor optimize_coding:
// set_fastest_defaults), the serial loop jchuff.c encode_mcu_huff -> encode_one_block -> emit_bits / emit_byte that
// compressor.write_scanlines / finish run (mozjpeg.rs:155-175).  Huffman coding is serial only through two running
// values, and both are prefix sums:
//   * the bit position of a block = the sum of the code lengths of the blocks before it  -> count pass, scan, write pass;
//   * the byte position after stuffing = position + the number of 0xFF bytes before it   -> count pass, scan, write pass;
//   * the DC predictor is the previous block's DC value, which lies in the plane (no running state at all).
// Launches per batch (all images of a batch in each): count (blocks), scan (per-workgroup bit sums), write (bits into a
// zeroed big-endian word stream: a lane ORs its first and last word, owns the ones between -- assembled in an LDS window
// per workgroup and copied out coalesced when the workgroup's piece fits it), count 0xFF (per 4 KiB of the
// stream), scan, write bytes (stuffed, behind the marker segments; the same pass zeroes the word stream for the next call
// and writes EOI and the file's length).  Bound by instruction issue of the block walk, not by HBM (coefficients are
// read twice, 256 B per block; the files are a tenth of that).
//
// The block routine and the placement rules live in jpeg_encode_core.hpp, shared with the CPU emulation of the tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <memory>
#include <vector>

#include "common.hpp"
#include "jpeg_encode_core.hpp"

#define HIP_TRY(expr)                                                                                   \
    do {                                                                                                \
        hipError_t e__ = (expr);                                                                        \
        if (e__ != hipSuccess)                                                                          \
            return fail(IFHIP_GPU_ERROR, "GpuError: %s failed: %s", #expr, hipGetErrorString(e__));     \
    } while (0)

namespace ifhip {

// jpeg_write.cpp
int jpeg_baseline_header(int ncomp, const uint8_t hs[3], const uint8_t vs[3], uint32_t width, uint32_t height,
                         const uint16_t qt[2][64], std::vector<uint8_t>* out);
void jpeg_std_encode_tables(uint32_t tabs[4][256]);
void jpeg_quality_tables(int quality, uint16_t qt[2][64]);

struct EncArgs {
    EncGeom g;
    const int16_t* coef[3];
    size_t plane_blocks[3];         // blocks per image in each plane
    uint32_t n_images, n_wg;        // n_wg: count / write workgroups per image
    const uint32_t* tabs;           // [4][256] dc0, ac0, dc1, ac1
    uint16_t* nbits;                // [n_images][nblocks]
    uint32_t* wg_bits;              // [n_images][n_wg] sums, then (scan) exclusive prefixes
    uint32_t* tot_bits;             // [n_images]
    uint32_t* words;                // [n_images][cap_words] the unstuffed stream, zero between calls
    size_t cap_words;
    uint32_t max_chunks;            // cap_words * 4 / kEncChunkBytes
    uint32_t* ff;                   // [n_images][max_chunks] 0xFF counts per chunk, then exclusive prefixes
    uint32_t* tot_ff;               // [n_images]
    uint32_t* status;               // [n_images] what the count pass and the scan found (read by every later pass)
    uint32_t* status_out;           // [n_images] the caller's copy, written once by the last pass (+ kEncFileOverflow)
    const uint8_t* header;
    uint32_t header_len;
    uint8_t* files;
    size_t file_pitch;
    uint32_t* lengths;              // [n_images]
};

constexpr uint32_t kBlkPitch = 33;  // dwords per staged block in LDS: 64 lanes reading the same coefficient of their own blocks hit 64 banks

struct DeviceStore {
    __device__ __forceinline__ static void shared(uint32_t* p, uint32_t v) { atomicOr(p, v); }
    __device__ __forceinline__ static void owned(uint32_t* p, uint32_t v) { *p = v; }
};

struct LdsCoef {                    // a lane's block in LDS, slot order (jpeg_encode_core.hpp enc_slot)
    const uint32_t* w;
    __device__ __forceinline__ int32_t operator()(int k) const { return reinterpret_cast<const int16_t*>(w)[enc_slot(static_cast<uint32_t>(k))]; }
    __device__ __forceinline__ uint32_t pair(int j) const { return w[j]; }
};

__device__ __forceinline__ uint32_t wave_inclusive_scan(uint32_t v, uint32_t lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t u = __shfl_up(v, d, 64);
        if (lane >= static_cast<uint32_t>(d)) v += u;
    }
    return v;
}

// exclusive scan over the T lanes of a workgroup (T a multiple of 64, <= 1024); *total = the sum.  `scratch`: T / 64 + 1 dwords.
template <uint32_t T>
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* scratch, uint32_t* total) {
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
    const uint32_t incl = wave_inclusive_scan(v, lane);
    __syncthreads();                                       // (scratch may still be read from a previous call)
    if (lane == 63u) scratch[wave] = incl;
    __syncthreads();
    uint32_t before = 0, sum = 0;
#pragma unroll
    for (uint32_t w = 0; w < T / 64u; ++w) {
        const uint32_t t = scratch[w];
        if (w < wave) before += t;
        sum += t;
    }
    *total = sum;
    return before + incl - v;
}

// Stage the workgroup's 256 scan-order blocks into LDS -- coalesced 16-byte loads (eight lanes per block), stored in the
// walk's slot order (zigzag positions, paired for the nonzero mask: enc_slot) -- and the tables.
// Returns this lane's DC predictor; *table: where its component's tables start in `tabs`.
__device__ __forceinline__ int32_t stage_blocks(const EncArgs& a, uint32_t img, uint32_t s0, uint32_t* blk, const int16_t** addr,
                                                uint32_t* tabs, uint32_t* table) {
    const uint32_t tid = threadIdx.x, s = s0 + tid;
    const int16_t* mine = nullptr;
    const int16_t* before = nullptr;                       // the DC value that predicts this lane's block
    if (s < a.g.nblocks) {
        const EncBlockRef r = enc_locate(a.g, s);
        const uint64_t planes[3] = {reinterpret_cast<uint64_t>(a.coef[0]), reinterpret_cast<uint64_t>(a.coef[1]), reinterpret_cast<uint64_t>(a.coef[2])};
        const uint64_t per_image[3] = {a.plane_blocks[0], a.plane_blocks[1], a.plane_blocks[2]};
        mine = reinterpret_cast<const int16_t*>(enc_sel3(planes, r.comp)) + (img * enc_sel3(per_image, r.comp) + r.offset) * 64u;
        if (r.pred_offset != 0xFFFFFFFFu) before = mine + (static_cast<ptrdiff_t>(r.pred_offset) - static_cast<ptrdiff_t>(r.offset)) * 64;
        *table = r.comp ? 512u : 0u;
    }
    // (every load below is unconditional, so all eight are in flight together -- with an "is this block inside the image"
    // test around each they went out one HBM round trip after the other)
    addr[tid] = mine;
    __syncthreads();
    // A lane carries the same piece (row of the block, natural order) in all eight steps: its eight coefficients go to
    // fixed slots of whatever block the step names.
    const uint32_t piece = tid & 7u;
    uint32_t pos[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pos[i] = enc_slot(static_cast<uint32_t>(enc_zigzag_position(static_cast<int>(piece) * 8 + i)));
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(1))) const u32x4 global_u32x4;    // (a pointer that went through LDS is a generic one to the compiler)
    // (a step that names a block behind the image's last names the tile's last block instead: every load is unconditional)
    const uint32_t last = min(kEncBlocksPerWg, a.g.nblocks - s0) - 1u;
    u32x4 v[8];
#pragma unroll
    for (uint32_t it = 0; it < 8u; ++it) {
        const int16_t* p = addr[min((it * kEncBlocksPerWg + tid) >> 3, last)];
        v[it] = *reinterpret_cast<global_u32x4*>(reinterpret_cast<uintptr_t>(p + piece * 8u));
    }
    // (tables and predictor behind the blocks: one round trip to memory for everything the workgroup stages)
    uint32_t tv[1024u / kEncBlocksPerWg];
#pragma unroll
    for (uint32_t i = 0; i < 1024u / kEncBlocksPerWg; ++i) tv[i] = a.tabs[i * kEncBlocksPerWg + tid];
    const int32_t pred = before ? *before : 0;
    // (step `it` names the block 32 further on: a constant distance, so the eight addresses of a lane are computed once)
    uint16_t* d[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) d[i] = reinterpret_cast<uint16_t*>(blk + (tid >> 3) * kBlkPitch) + pos[i];
#pragma unroll
    for (uint32_t it = 0; it < 8u; ++it) {
        constexpr uint32_t kStep = (kEncBlocksPerWg / 8u) * kBlkPitch * 2u;      // halfwords between the blocks of two steps
        d[0][it * kStep] = static_cast<uint16_t>(v[it].x); d[1][it * kStep] = static_cast<uint16_t>(v[it].x >> 16);
        d[2][it * kStep] = static_cast<uint16_t>(v[it].y); d[3][it * kStep] = static_cast<uint16_t>(v[it].y >> 16);
        d[4][it * kStep] = static_cast<uint16_t>(v[it].z); d[5][it * kStep] = static_cast<uint16_t>(v[it].z >> 16);
        d[6][it * kStep] = static_cast<uint16_t>(v[it].w); d[7][it * kStep] = static_cast<uint16_t>(v[it].w >> 16);
    }
#pragma unroll
    for (uint32_t i = 0; i < 1024u / kEncBlocksPerWg; ++i) tabs[i * kEncBlocksPerWg + tid] = tv[i];
    __syncthreads();
    return pred;
}

__global__ __launch_bounds__(256) void enc_count_kernel(const EncArgs a) {
    __shared__ uint32_t blk[kEncBlocksPerWg * kBlkPitch];
    __shared__ const int16_t* addr[kEncBlocksPerWg];
    __shared__ uint32_t tabs[1024];
    __shared__ uint32_t scratch[8];
    const uint32_t tid = threadIdx.x, img = blockIdx.y, s0 = blockIdx.x * kEncBlocksPerWg, s = s0 + tid;
    uint32_t t = 0;
    const int32_t pred = stage_blocks(a, img, s0, blk, addr, tabs, &t);
    uint32_t bits = 0, bad = 0;
    if (s < a.g.nblocks) {
        EncCountSink sink;
        bad = enc_block(LdsCoef{blk + tid * kBlkPitch}, pred, tabs + t, tabs + t + 256u, sink);
        bits = sink.bits;
        a.nbits[static_cast<size_t>(img) * a.g.nblocks + s] = static_cast<uint16_t>(bits);
    }
    uint32_t total;
    block_exclusive_scan<256>(bits, scratch, &total);
    if (tid == 0u) a.wg_bits[static_cast<size_t>(img) * a.n_wg + blockIdx.x] = total;
    if (bad) atomicOr(a.status + img, kEncBadCoef);
}

// Exclusive scan, one workgroup of 1024 lanes per image.  mode 0: the n_wg per-workgroup bit sums -> tot_bits (and the
// capacity check of the word stream); mode 1: the per-chunk 0xFF counts of the image's stream -> tot_ff.
__global__ __launch_bounds__(1024) void enc_scan_kernel(const EncArgs a, const int mode) {
    __shared__ uint32_t scratch[20];
    const uint32_t tid = threadIdx.x, img = blockIdx.x;
    uint32_t n;
    uint32_t* v;
    if (mode == 0) {
        n = a.n_wg;
        v = a.wg_bits + static_cast<size_t>(img) * a.n_wg;
    } else {
        const uint32_t bytes = a.status[img] ? 0u : (a.tot_bits[img] + 7u) >> 3;
        n = (bytes + kEncChunkBytes - 1u) / kEncChunkBytes;
        v = a.ff + static_cast<size_t>(img) * a.max_chunks;
    }
    uint32_t carry = 0;
    bool overflow = false;
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + tid, x = i < n ? v[i] : 0u;
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<1024>(x, scratch, &total);
        if (i < n) v[i] = carry + ex;
        overflow |= carry + total < carry;
        carry += total;
    }
    if (tid == 0u) {
        if (mode == 0) {
            a.tot_bits[img] = carry;
            // (the same bound ifhip_jpeg_enc_stage_max_file_bytes sizes a file for: the capacity minus its spare chunk -- a
            // scan between the two passed this check and was then dropped as a file overflow)
            if (overflow || (static_cast<uint64_t>(carry) + 7u) / 8u > a.cap_words * 4u - kEncChunkBytes) atomicOr(a.status + img, kEncScanOverflow);
        } else {
            a.tot_ff[img] = carry;
        }
    }
}

struct LdsStore {                   // the window: every word is ORed in (an LDS atomic costs what a store costs, and the
    __device__ __forceinline__ static void shared(uint32_t* p, uint32_t v) { atomicOr(p, v); }   // walk carries no "first word" state)
    __device__ __forceinline__ static void owned(uint32_t* p, uint32_t v) { atomicOr(p, v); }
};

__global__ __launch_bounds__(256) void enc_write_kernel(const EncArgs a) {
    __shared__ uint32_t blk[kEncBlocksPerWg * kBlkPitch];
    __shared__ uint32_t win[kEncWindowWords + 1u];         // (+ the sink's spare word; the head holds the blocks' addresses while they are staged)
    __shared__ uint32_t tabs[1024];
    __shared__ uint32_t scratch[8];
    static_assert(kEncWindowWords * sizeof(uint32_t) >= kEncBlocksPerWg * sizeof(const int16_t*), "the address list must fit the window");
    const uint32_t tid = threadIdx.x, img = blockIdx.y, s0 = blockIdx.x * kEncBlocksPerWg, s = s0 + tid;
    if (a.status[img]) return;                             // (uniform: out-of-range coefficient or stream capacity; nothing is written)
    uint32_t t = 0;
    const int32_t pred = stage_blocks(a, img, s0, blk, reinterpret_cast<const int16_t**>(win), tabs, &t);
    const bool valid = s < a.g.nblocks;
    const uint32_t mine = valid ? a.nbits[static_cast<size_t>(img) * a.g.nblocks + s] : 0u;
    uint32_t total;
    const uint32_t base = a.wg_bits[static_cast<size_t>(img) * a.n_wg + blockIdx.x];
    const uint32_t local = block_exclusive_scan<256>(mine, scratch, &total);
    const bool last_wg = blockIdx.x == a.n_wg - 1u;
    if (last_wg) total += enc_final_padding(base + total); // jchuff.c flush_bits: the last byte is filled with 1 bits
    const uint32_t n_words = enc_window_words(base, total);
    const bool windowed = n_words <= kEncWindowWords;      // (uniform)
    uint32_t* stream = a.words + static_cast<size_t>(img) * a.cap_words;
    if (windowed) {
        for (uint32_t i = tid; i <= n_words; i += kEncBlocksPerWg) win[i] = 0u;
        __syncthreads();
    }
    if (valid) {
        const LdsCoef coef{blk + tid * kBlkPitch};
        if (windowed) {
            EncWindowSink<LdsStore> sink(win, (base & 31u) + local);
            enc_block(coef, pred, tabs + t, tabs + t + 256u, sink);
            if (s == a.g.nblocks - 1u) {
                const uint32_t pad = (8u - sink.bits_in_last_byte()) & 7u;
                if (pad) sink.put((1u << pad) - 1u, pad);
            }
        } else {
            EncWordSink<DeviceStore> sink(stream, base + local);
            enc_block(coef, pred, tabs + t, tabs + t + 256u, sink);
            if (s == a.g.nblocks - 1u) {
                const uint32_t pad = (8u - sink.bits_in_last_byte()) & 7u;
                if (pad) sink.put((1u << pad) - 1u, pad);
            }
            sink.finish();
        }
    }
    if (windowed) {                                        // the piece leaves the window with coalesced stores; its two ends are shared
        __syncthreads();
        uint32_t* dst = stream + (base >> 5);
        for (uint32_t i = tid; i < n_words; i += kEncBlocksPerWg) {
            const uint32_t v = __builtin_bswap32(win[i]);
            if (i == 0u || i == n_words - 1u) atomicOr(dst + i, v);
            else dst[i] = v;
        }
    }
}

// the stream bytes of an image: 0 when the image is dropped
__device__ __forceinline__ uint32_t stream_bytes(const EncArgs& a, uint32_t img) { return a.status[img] ? 0u : (a.tot_bits[img] + 7u) >> 3; }

__global__ __launch_bounds__(256) void enc_ff_count_kernel(const EncArgs a) {
    __shared__ uint32_t scratch[8];
    const uint32_t tid = threadIdx.x, img = blockIdx.y;
    const uint32_t bytes = stream_bytes(a, img), chunks = (bytes + kEncChunkBytes - 1u) / kEncChunkBytes;
    const uint4* w = reinterpret_cast<const uint4*>(a.words + static_cast<size_t>(img) * a.cap_words);
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t at = chunk * kEncChunkBytes + tid * 16u;
        uint32_t c = 0;
        if (at < bytes) {                                  // (bytes behind the stream's end inside the last 16 are zero)
            const uint4 v = w[at >> 4];
            c = enc_count_ff(v.x) + enc_count_ff(v.y) + enc_count_ff(v.z) + enc_count_ff(v.w);
        }
        uint32_t total;
        block_exclusive_scan<256>(c, scratch, &total);
        if (tid == 0u) a.ff[static_cast<size_t>(img) * a.max_chunks + chunk] = total;
    }
}

__global__ __launch_bounds__(256) void enc_stuff_kernel(const EncArgs a) {
    __shared__ uint32_t scratch[8];
    __shared__ uint8_t obuf[2u * kEncChunkBytes];          // a chunk's stuffed bytes (every byte 0xFF at worst)
    const uint32_t tid = threadIdx.x, img = blockIdx.y;
    const uint32_t st = a.status[img];
    // (a dropped image has written no word, its stream is still zero; nothing to clean)
    const uint32_t bytes = st ? 0u : (a.tot_bits[img] + 7u) >> 3, chunks = (bytes + kEncChunkBytes - 1u) / kEncChunkBytes;
    const uint64_t file_len = static_cast<uint64_t>(a.header_len) + bytes + a.tot_ff[img] + 2u;
    const bool fits = !st && file_len <= a.file_pitch;
    uint4* w = reinterpret_cast<uint4*>(a.words + static_cast<size_t>(img) * a.cap_words);
    uint8_t* out = a.files + static_cast<size_t>(img) * a.file_pitch;
    if (blockIdx.x == 0u) {
        if (fits) {
            for (uint32_t i = tid; i < a.header_len; i += 256u) out[i] = a.header[i];
            if (tid == 0u) { out[file_len - 2u] = 0xFF; out[file_len - 1u] = 0xD9; }
        }
        if (tid == 0u) {
            a.lengths[img] = fits ? static_cast<uint32_t>(file_len) : 0u;
            if (a.status_out) a.status_out[img] = st | (!fits && !st ? kEncFileOverflow : 0u);
        }
    }
    for (uint32_t chunk = blockIdx.x; chunk < chunks; chunk += gridDim.x) {
        const uint32_t at = chunk * kEncChunkBytes + tid * 16u;
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        uint32_t c = 0;
        if (at < bytes) {
            v = w[at >> 4];
            w[at >> 4] = make_uint4(0u, 0u, 0u, 0u);       // the stream is zero again for the next call
            c = enc_count_ff(v.x) + enc_count_ff(v.y) + enc_count_ff(v.z) + enc_count_ff(v.w);
        }
        uint32_t total;
        const uint32_t ex = block_exclusive_scan<256>(c, scratch, &total);
        if (!fits) continue;                               // (uniform)
        // The lanes' bytes meet in LDS and leave as runs of consecutive bytes: stored straight from the lanes, every store
        // instruction scattered its 64 bytes over a kilobyte of the file (sixteen partial cache lines each).
        if (at < bytes) {
            uint8_t* d = obuf + tid * 16u + ex;
            const uint32_t ws[4] = {v.x, v.y, v.z, v.w};
            const uint32_t nb = bytes - at < 16u ? bytes - at : 16u;
#pragma unroll
            for (uint32_t j = 0; j < 16u; ++j) {
                if (j < nb) {
                    const uint32_t b = (ws[j >> 2] >> (8u * (j & 3u))) & 255u;
                    *d++ = static_cast<uint8_t>(b);
                    if (b == 255u) *d++ = 0u;
                }
            }
        }
        __syncthreads();
        const uint32_t n_in = bytes - chunk * kEncChunkBytes < kEncChunkBytes ? bytes - chunk * kEncChunkBytes : kEncChunkBytes;
        uint8_t* dst = out + a.header_len + chunk * kEncChunkBytes + a.ff[static_cast<size_t>(img) * a.max_chunks + chunk];
        for (uint32_t i = tid; i < n_in + total; i += 256u) dst[i] = obuf[i];
        __syncthreads();                                   // (the buffer is reused by the next chunk)
    }
}

}  // namespace ifhip

using namespace ifhip;

struct ifhip_jpeg_enc_stage {
    EncGeom g;
    uint32_t width = 0, height = 0;
    uint8_t hs[3] = {1, 1, 1}, vs[3] = {1, 1, 1};
    size_t plane_blocks[3] = {0, 0, 0};
    uint32_t max_images = 0, n_wg = 0, max_chunks = 0;
    size_t cap_words = 0;
    int device = -1;
    int header_quality = -1;
    uint32_t header_len = 0;
    // device buffers
    uint32_t* d_tabs = nullptr;
    uint16_t* d_nbits = nullptr;
    uint32_t *d_wg_bits = nullptr, *d_tot_bits = nullptr, *d_words = nullptr, *d_ff = nullptr, *d_tot_ff = nullptr, *d_status = nullptr;
    uint8_t* d_header = nullptr;
    uint8_t* h_header = nullptr;    // pinned
    ~ifhip_jpeg_enc_stage() {
        (void)DEV_FREE(d_tabs); (void)DEV_FREE(d_nbits); (void)DEV_FREE(d_wg_bits); (void)DEV_FREE(d_tot_bits); (void)DEV_FREE(d_words);
        (void)DEV_FREE(d_ff); (void)DEV_FREE(d_tot_ff); (void)DEV_FREE(d_status); (void)DEV_FREE(d_header);
        if (h_header) (void)cached_host_free(h_header);
    }
};

namespace {
constexpr uint32_t kHeaderCap = 1024;

int make_enc_geom(uint32_t width, uint32_t height, int ncomp, const uint8_t* hs, const uint8_t* vs, const uint32_t* bw, const uint32_t* bh,
                  EncGeom* g) {
    switch (enc_make_geom(width, height, ncomp, hs, vs, bw, bh, g)) {
    case 0: return IFHIP_OK;
    case 1: return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: 1 or 3 components, sampling factors 1..2, 1..65535 pixels per side");
    case 2: return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: a coefficient plane is smaller than the MCU grid");
    default: return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: more than %llu blocks per image (32-bit bit positions)",
                         static_cast<unsigned long long>((1ull << 32) / kEncMaxBitsPerBlock));
    }
}
}  // namespace

// ---- a batch's files as ONE message (the job's only exchange is the gather of its outputs) -------------
// ifhip_jpeg_encode_batch_device leaves image i's file at d_files + i * file_pitch: a pitch sized for the worst case, of which
// a q90 file fills a sixth.  What a rank sends to the root is the files back to back, every file starting on a 16-byte
// boundary (<= 15 bytes of padding per file, so that both sides of the copy are 16-byte aligned), and the n + 1 offsets.
namespace ifhip {
__global__ void __launch_bounds__(1024) pack_offsets_kernel(const uint32_t* __restrict__ lengths, uint32_t n, uint64_t* __restrict__ offsets) {
    __shared__ uint64_t wave_tot[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (uint32_t base = 0; base < n; base += 1024u) {
        const uint32_t i = base + threadIdx.x;
        const uint64_t mine = i < n ? (static_cast<uint64_t>(lengths[i]) + 15u) & ~static_cast<uint64_t>(15u) : 0u;
        uint64_t inc = mine;
#pragma unroll
        for (uint32_t d = 1; d < 64u; d <<= 1) {
            const uint64_t o = __shfl_up(inc, d, 64);
            if ((threadIdx.x & 63u) >= d) inc += o;
        }
        if ((threadIdx.x & 63u) == 63u) wave_tot[threadIdx.x >> 6] = inc;
        __syncthreads();
        uint64_t before = carry;
        for (uint32_t w = 0; w < (threadIdx.x >> 6); ++w) before += wave_tot[w];
        if (i < n) offsets[i] = before + inc - mine;
        __syncthreads();
        if (threadIdx.x == 1023u) carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) offsets[n] = carry;
}
__global__ void __launch_bounds__(256) pack_copy_kernel(const uint8_t* __restrict__ files, size_t pitch, const uint32_t* __restrict__ lengths,
                                                        const uint64_t* __restrict__ offsets, uint8_t* __restrict__ out, uint64_t capacity) {
    const uint32_t i = blockIdx.y;
    const uint64_t at = offsets[i];
    const uint32_t quads = (lengths[i] + 15u) >> 4;                   // (the tail's padding is whatever lies behind the file: inside its pitch)
    if (at + static_cast<uint64_t>(quads) * 16u > capacity) return;   // does not fit: the caller sees offsets[n] > capacity
    const uint4* src = reinterpret_cast<const uint4*>(files + static_cast<size_t>(i) * pitch);
    uint4* dst = reinterpret_cast<uint4*>(out + at);
    for (uint32_t q = blockIdx.x * 256u + threadIdx.x; q < quads; q += gridDim.x * 256u) dst[q] = src[q];
}
}  // namespace ifhip

extern "C" {

int ifhip_jpeg_enc_stage_create(ifhip_jpeg_enc_stage** stage, uint32_t width, uint32_t height, int n_components, const uint8_t* h_samp,
                                const uint8_t* v_samp, const uint32_t* blocks_w3, const uint32_t* blocks_h3, uint32_t max_images,
                                size_t scan_capacity) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage out-pointer");
    *stage = nullptr;
    if (!h_samp || !v_samp || !blocks_w3 || !blocks_h3) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    if (max_images == 0 || max_images > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: 1..65535 images per stage");
    std::unique_ptr<ifhip_jpeg_enc_stage> s(new ifhip_jpeg_enc_stage);
    int rc = make_enc_geom(width, height, n_components, h_samp, v_samp, blocks_w3, blocks_h3, &s->g);
    if (rc) return rc;
    s->width = width; s->height = height;
    for (int c = 0; c < n_components; ++c) {
        s->hs[c] = static_cast<uint8_t>(s->g.H[c]); s->vs[c] = static_cast<uint8_t>(s->g.V[c]);
        s->plane_blocks[c] = static_cast<size_t>(blocks_w3[c]) * blocks_h3[c];
    }
    s->n_wg = (s->g.nblocks + kEncBlocksPerWg - 1u) / kEncBlocksPerWg;
    // the unstuffed stream of an image: the caller's bound, or the most the geometry can produce (1 665 bits per block)
    const uint64_t worst = (static_cast<uint64_t>(s->g.nblocks) * kEncMaxBitsPerBlock + 7u) / 8u;
    uint64_t cap = scan_capacity ? std::min<uint64_t>(scan_capacity, worst) : worst;
    cap = (cap + kEncChunkBytes - 1u) / kEncChunkBytes * kEncChunkBytes + kEncChunkBytes;      // whole chunks, one to spare
    s->cap_words = static_cast<size_t>(cap / 4u);
    s->max_chunks = static_cast<uint32_t>(cap / kEncChunkBytes);
    if (int arc = require_gfx950(&s->device)) return arc;
    s->max_images = max_images;
    const size_t n = max_images;
    HIP_TRY(DEV_MALLOC(&s->d_tabs, 4096));
    HIP_TRY(DEV_MALLOC(&s->d_nbits, n * s->g.nblocks * sizeof(uint16_t)));
    HIP_TRY(DEV_MALLOC(&s->d_wg_bits, n * s->n_wg * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_tot_bits, n * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_words, n * s->cap_words * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_ff, n * s->max_chunks * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_tot_ff, n * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_status, n * sizeof(uint32_t)));
    HIP_TRY(DEV_MALLOC(&s->d_header, kHeaderCap));
    HIP_TRY(static_cast<hipError_t>(cached_host_malloc(reinterpret_cast<void**>(&s->h_header), kHeaderCap)));
    uint32_t tabs[4][256];
    jpeg_std_encode_tables(tabs);
    HIP_TRY(static_cast<hipError_t>(copy_to_device(s->d_tabs, tabs, sizeof tabs)));
    HIP_TRY(static_cast<hipError_t>(zero_device(s->d_words, n * s->cap_words * sizeof(uint32_t))));       // the stuffing pass keeps it zero from here on
    *stage = s.release();
    return IFHIP_OK;
}

void ifhip_jpeg_enc_stage_destroy(ifhip_jpeg_enc_stage* stage) { delete stage; }

size_t ifhip_jpeg_enc_stage_max_file_bytes(const ifhip_jpeg_enc_stage* stage) {
    // marker segments + the stream with every byte stuffed + EOI
    return stage ? static_cast<size_t>(kHeaderCap) + 2u * (stage->cap_words * 4u - kEncChunkBytes) + 2u : 0u;
}

int ifhip_jpeg_encode_batch_device(ifhip_jpeg_enc_stage* stage, const int16_t* d_coef0, const int16_t* d_coef1, const int16_t* d_coef2,
                                   int quality, uint32_t n_images, uint8_t* d_files, size_t file_pitch, uint32_t* d_lengths,
                                   uint32_t* d_status, void* hip_stream) {
    if (!stage) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null stage");
    if (n_images == 0) return IFHIP_OK;
    if (n_images > stage->max_images) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: %u images exceed the stage capacity %u", n_images, stage->max_images);
    if (!d_coef0 || (stage->g.ncomp == 3 && (!d_coef1 || !d_coef2)) || !d_files || !d_lengths)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    if ((reinterpret_cast<uintptr_t>(d_coef0) | reinterpret_cast<uintptr_t>(d_coef1) | reinterpret_cast<uintptr_t>(d_coef2)) & 15u)
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: coefficient planes must be 16-byte aligned");
    if (file_pitch < kHeaderCap) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: file_pitch below %u bytes", kHeaderCap);
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev != stage->device) return fail(IFHIP_INVALID_STATE, "InvalidState: stage belongs to device %d, current device is %d", stage->device, dev);
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    if (quality != stage->header_quality) {
        uint16_t qt[2][64];
        jpeg_quality_tables(quality, qt);
        std::vector<uint8_t> h;
        int rc = jpeg_baseline_header(static_cast<int>(stage->g.ncomp), stage->hs, stage->vs, stage->width, stage->height, qt, &h);
        if (rc) return rc;
        if (h.size() > kHeaderCap) return fail(IFHIP_INVALID_STATE, "InvalidState: marker segments of %zu bytes", h.size());
        HIP_TRY(static_cast<hipError_t>(ifhip::wait_stream(st)));                 // (the pinned copy may still be on its way to the device from the previous call)
        std::memcpy(stage->h_header, h.data(), h.size());
        stage->header_len = static_cast<uint32_t>(h.size());
        HIP_TRY(hipMemcpyAsync(stage->d_header, stage->h_header, h.size(), hipMemcpyHostToDevice, st));
        stage->header_quality = quality;
    }
    EncArgs a;
    std::memset(&a, 0, sizeof a);
    a.g = stage->g;
    a.coef[0] = d_coef0; a.coef[1] = d_coef1; a.coef[2] = d_coef2;
    for (int c = 0; c < 3; ++c) a.plane_blocks[c] = stage->plane_blocks[c];
    a.n_images = n_images; a.n_wg = stage->n_wg; a.tabs = stage->d_tabs; a.nbits = stage->d_nbits; a.wg_bits = stage->d_wg_bits;
    a.tot_bits = stage->d_tot_bits; a.words = stage->d_words; a.cap_words = stage->cap_words; a.max_chunks = stage->max_chunks;
    a.ff = stage->d_ff; a.tot_ff = stage->d_tot_ff; a.status = stage->d_status; a.header = stage->d_header; a.header_len = stage->header_len;
    a.files = d_files; a.file_pitch = file_pitch; a.lengths = d_lengths; a.status_out = d_status;
    HIP_TRY(hipMemsetAsync(stage->d_status, 0, n_images * sizeof(uint32_t), st));
    const dim3 blocks_grid(stage->n_wg, n_images);
    const dim3 chunk_grid(std::min<uint32_t>(stage->max_chunks, 256u), n_images);
    hipLaunchKernelGGL(enc_count_kernel, blocks_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(enc_scan_kernel, dim3(n_images), dim3(1024), 0, st, a, 0);
    hipLaunchKernelGGL(enc_write_kernel, blocks_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(enc_ff_count_kernel, chunk_grid, dim3(256), 0, st, a);
    hipLaunchKernelGGL(enc_scan_kernel, dim3(n_images), dim3(1024), 0, st, a, 1);
    hipLaunchKernelGGL(enc_stuff_kernel, chunk_grid, dim3(256), 0, st, a);
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}


int ifhip_pack_files_device(const uint8_t* d_files, size_t file_pitch, const uint32_t* d_lengths, uint32_t n_files, uint8_t* d_out,
                            size_t out_capacity, uint64_t* d_offsets, void* hip_stream) {
    if (!d_files || !d_lengths || !d_out || !d_offsets) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: null pointer");
    if ((file_pitch & 15u) || (reinterpret_cast<uintptr_t>(d_files) & 15u) || (reinterpret_cast<uintptr_t>(d_out) & 15u))
        return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: files, pitch and the packed buffer must be 16-byte aligned");
    if (n_files == 0) return IFHIP_OK;
    if (n_files > 65535u) return fail(IFHIP_INVALID_ARGUMENT, "InvalidArgument: at most 65535 files per call");
    hipStream_t st = static_cast<hipStream_t>(hip_stream);
    hipLaunchKernelGGL(ifhip::pack_offsets_kernel, dim3(1), dim3(1024), 0, st, d_lengths, n_files, d_offsets);
    hipLaunchKernelGGL(ifhip::pack_copy_kernel, dim3(16, n_files), dim3(256), 0, st, d_files, file_pitch, d_lengths, d_offsets, d_out,
                       static_cast<uint64_t>(out_capacity));
    HIP_TRY(hipGetLastError());
    return IFHIP_OK;
}

}  // extern "C"
