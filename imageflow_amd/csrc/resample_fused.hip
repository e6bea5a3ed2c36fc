// resample_fused.hip -- the fused resample + render kernel of imageflow's hot path for ONE ring size K
// (compiled once per K = 1..8 with -DIFHIP_FUSED_K=K, in parallel; see imageflow_amd/build.py).
//
// Replaces the arithmetic behind imageflow_core::graphics::scaling::scale_and_render
// (graphics/scaling.rs:19-90): sample -> working float (graphics/color.rs:22-45), vertical then horizontal
// weighted convolution driven by PixelRowWeights tables (graphics/weights.rs:521-571,681-788), and the three
// output stages (scaling.rs:211-251 ReplaceSelf, :119-148 BlendWithMatte, :254-287 BlendWithSelf).
//
// Bound: HBM.  This is a 1-D stencil per axis, so there is no MFMA here; the design points are
//   * every source byte is read from HBM exactly once, 16 B per lane, rows fully coalesced;
//   * the vertical pass never leaves registers: each lane owns 4 source columns and a ring of K live
//     output rows; the per-row weights are wave-uniform and arrive through the scalar cache (VStep);
//   * only the 10-20x smaller vertically-reduced row goes through LDS for the horizontal pass;
//   * the sRGB->linear table lives in LDS, one copy per bank (one conflict-free ds_read per channel sample).
// Build with -ffp-contract=off: every fused multiply-add below is an explicit fmaf, everything else rounds
// separately, exactly as the arithmetic contract in oracle/if_oracle.c (tests compare bit for bit).
#include <atomic>
#include <type_traits>

#include "resample_device.hpp"

#ifndef IFHIP_FUSED_K
#error "compile with -DIFHIP_FUSED_K=<ring size 1..8>"
#endif
namespace ifhip {

// ------------------------------------------------------------------------------------------------------
// Fused kernel: one workgroup = (image, band of output rows, column strip)
//
// Vertical pass in registers, one lane = 4 source columns x K live output rows; when an output row completes, its
// vertically filtered row goes to LDS (double buffered) and the horizontal pass for it is *interleaved* into the
// following source-row steps, a few taps per step, so that its LDS latency and its strictly sequential fmaf chains
// hide under the vertical pass instead of stopping it.  One LDS-only workgroup barrier per output row.
// ------------------------------------------------------------------------------------------------------
// FG > 0: the fast horizontal pass (moderate ratios: no output needs more than 4 groups).  Every output runs exactly FG
// 4-tap groups -- its weight row is zero-padded to FG groups, +0 weights are exact -- fully unrolled with immediate
// LDS offsets from two base addresses, a 4-byte record per output, groups of the vertically filtered row interleaved.
// YCC: the source is the JPEG stage's three component planes at output resolution (ResampleArgs::in / in_cb / in_cr) instead
// of a BGRA bitmap: the row fetch reads 4 samples of each plane and jdcolor.c's fixed-point YCbCr -> RGB runs in
// registers in front of the table gathers, so a decoded BGRA frame never exists in HBM (mozjpeg_decoder.rs:346-362 feeding
// scale_render.rs:304-313).  No alpha (a JPEG has none), 4 pixels per lane.
template <int K, bool ALPHA, bool WLDS, bool PERPIXEL, int FG = 0, bool YCC = false>
__global__ void __launch_bounds__(fused_max_threads(K, ALPHA ? 4 : 3))
// one workgroup per CU (LDS): the register budget is the one of exactly its waves, say so (without it the register
// allocator aims at one more wave per SIMD than can ever be resident)
__attribute__((amdgpu_waves_per_eu(fused_max_threads(K, ALPHA ? 4 : 3) / 256, fused_max_threads(K, ALPHA ? 4 : 3) / 256)))
fused_resample_kernel(const ResampleArgs a, const VStep* __restrict__ steps) {
    static_assert(FG == 0 || (WLDS && PERPIXEL), "the fast horizontal pass keeps its weights in LDS and maps one lane per pixel");
    // FG >= 16: the fast pass with FG - 16 groups of TWO source columns (8-byte LDS reads; BGRA sources without alpha): windows of
    // 5-6 taps cost 4 x 2 taps instead of 3 x 4 (cfg3 level 1, 1600 -> 1200).  Same taps in the same order, +0 padding: same pixels.
    constexpr bool TWO = FG >= 16;
    static_assert(!TWO || (!ALPHA && !YCC), "two-column groups: three channels from a BGRA source");
    static_assert(!YCC || (!ALPHA && fused_shape(K, 3).px == 4), "planar source: three channels, four pixels per lane");
    // `steps` is a separate __restrict__ argument (not a field of `a`) so that the compiler can prove the canvas
    // stores never clobber it and keeps the per-step 64-byte records on the scalar path.
    constexpr int C = ALPHA ? 4 : 3;
    constexpr int D = fused_shape(K, C).rows_in_flight;
    constexpr bool PIPE = fused_shape(K, C).pipelined != 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    // A workgroup works on F = a.frames_per_wg frames side by side (F > 1 only for sources narrower than half the
    // workgroup: the tables in LDS are shared, the CU keeps a full complement of waves).  `wtid` indexes the whole
    // workgroup (table fills), `tid` the lane within its frame slot, T the lanes per slot; slots are whole waves.
    const uint32_t wtid = threadIdx.x;
    const uint32_t WT = blockDim.x;
    const uint32_t T = a.lanes_per_frame;
    const uint32_t slot = __builtin_amdgcn_readfirstlane(wtid / T);    // slots are whole waves: wave-uniform, lives in an SGPR
    const uint32_t tid = wtid - slot * T;
    // Workgroups go to the 8 XCDs round-robin (id % 8), each XCD with its own L2.  With several column strips per frame,
    // neighbouring strips read the same halo columns at about the same time: renumber so that consecutive logical
    // workgroups (the strips of one frame) share an XCD and the halo is fetched from HBM once.
    uint32_t b = blockIdx.x;
    if (a.n_strips > 1u && (gridDim.x & 7u) == 0u) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const uint32_t strip_i = b % a.n_strips; b /= a.n_strips;
    const uint32_t band = b % a.n_bands;
    const uint32_t img_raw = (b / a.n_bands) * a.frames_per_wg + slot;
    const bool img_on = img_raw < a.n_images;              // the last workgroup may have idle slots: they run the same
    const uint32_t img = img_on ? img_raw : a.n_images - 1u;   // schedule (barriers!) on the last frame and store nothing

    const Strip strip = a.strips[strip_i];
    const uint32_t n_u = strip.u1 - strip.u0;

    constexpr uint32_t fast_g = TWO ? FG - 16 : FG;                    // groups per output (what the host planned the LDS with)
    constexpr uint32_t GP = fused_group_pitch(C);                      // bytes per interleaved 4-pixel group (fast pass)
    const FusedLds L = fused_lds_layout(n_u, strip.nquads, a.h_wu_floats, C, WLDS, a.l2s_in_lds != 0, a.lut_copies_log2, PERPIXEL,
                                        a.frames_per_wg, fast_g);
    float* lut_banked = reinterpret_cast<float*>(smem + L.lut);      // [256][32] floats, one copy per bank
    uint16_t* thr = reinterpret_cast<uint16_t*>(smem + L.thr);       // 256 linear->sRGB thresholds
    uint4* hmeta = reinterpret_cast<uint4*>(smem + L.hmeta);         // per output column {left - cx0, taps, w offset}
    float* obuf = reinterpret_cast<float*>(smem + L.obuf + slot * L.obuf_stride);   // 2 x [n_u][4] horizontally filtered rows
    const float* hw_lds = reinterpret_cast<const float*>(smem + L.hw);
    float* inter = reinterpret_cast<float*>(smem + L.inter + slot * 2u * L.inter_stride);   // 2 x vertically filtered row
    const uint32_t inter_stride = L.inter_stride >> 2;               // floats per buffered row
    const uint32_t plane_pitch = L.plane_pitch;                      // floats per sub-plane
    const uint32_t obuf_stride = n_u * 4u;                           // floats

    for (uint32_t i = wtid; i < (256u << a.lut_copies_log2); i += WT) lut_banked[i] = a.lut_in[i >> a.lut_copies_log2];
    for (uint32_t i = wtid; i < 256u; i += WT) thr[i] = a.l2s_thr[i];
    const uint8_t* l2s_lds = a.l2s_in_lds ? smem + L.l2s : nullptr;
    if (a.l2s_in_lds)
        for (uint32_t i = wtid; i < 1024u; i += WT)
            reinterpret_cast<uint4*>(smem + L.l2s)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    const BankedLut lut{lut_banked, wtid & ((1u << a.lut_copies_log2) - 1u), a.lut_copies_log2};
    uint32_t* hmeta2 = reinterpret_cast<uint32_t*>(smem + L.hmeta);  // fast pass: first group (relative to the strip) | row id << 16
    if constexpr (FG > 0) {
        for (uint32_t i = wtid; i < n_u; i += WT) hmeta2[i] = a.h_meta2[strip.u0 + i] - (strip.cx0 >> (TWO ? 1 : 2));
        // the groups past the staged columns are read (with weight +0) and never written: they must hold finite values
        float4* z = reinterpret_cast<float4*>(smem + L.inter);
        for (uint32_t i = wtid; i < (a.frames_per_wg * 2u * L.inter_stride) >> 4; i += WT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (uint32_t i = wtid; i < n_u; i += WT) {
            uint4 m = a.h_meta[strip.u0 + i];
            m.x -= strip.cx0;
            hmeta[i] = m;
        }
    }
    if (WLDS) {
        const float4* src4 = reinterpret_cast<const float4*>(a.h_wu);
        float4* dst4 = reinterpret_cast<float4*>(smem + L.hw);
        for (uint32_t i = wtid; i < (a.h_wu_floats >> 2); i += WT) dst4[i] = src4[i];
    }
    __syncthreads();
    // Workgroup barrier that orders LDS traffic only.  __syncthreads() would also drain vmcnt, i.e. throw away the
    // D source rows every lane keeps in flight.
    auto lds_barrier = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
    };

    const uint32_t s0 = a.band_begin[band], s1 = a.band_begin[band + 1];
    constexpr int PX = fused_shape(K, C).px;                // source pixels per lane: 4 (16-byte loads) or 2 (8-byte loads)
    const uint32_t n_groups = strip.nquads * (4u / PX);     // lanes that own source columns
    const bool lane_on = tid < n_groups;
    // lanes past the strip re-read its last quad instead of branching: every row load is unconditional, so the
    // number of loads in flight is known statically and the compiler can wait with vmcnt(D-1) instead of vmcnt(0)
    const uint32_t quad = lane_on ? tid : n_groups - 1u;
    // bytes per source pixel in the plane(s) read: 4 (BGRA) or 1 (one of three component planes)
    // A row's address = (frame + strip + row), all wave-uniform (a frame slot is whole waves: block_for) and left to the
    // scalar unit, + this lane's columns, a 32-bit offset: the load takes base and offset as they are, no vector
    // instruction forms an address (the 64-bit multiply-add the compiler otherwise emits per row issues at a quarter rate).
    constexpr uint32_t BPP = YCC ? 1u : 4u;
    const uint32_t img_u = __builtin_amdgcn_readfirstlane(img);
    const uint8_t* src = a.in + (static_cast<size_t>(img_u) * a.in_image_bytes + static_cast<size_t>(strip.cx0) * BPP);
    const uint32_t lane_off0 = static_cast<uint32_t>(PX) * quad * BPP;

    // Ring accumulators and converted samples live as float2 pairs over the flattened (pixel, channel) index
    // f = p*C + c, so that the vertical pass issues v_pk_fma_f32 (two IEEE fmaf per instruction, the weight broadcast
    // from its SGPR): half the VALU issue slots of scalar v_fma_f32 for bit-identical results.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    constexpr int NP = PX * C / 2;                  // pairs per lane and ring slot (PX pixels x C channels)
    f32x2 acc[K][NP];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[s][i] = f32x2{0.0f, 0.0f};
    auto acc_at = [&](int s, int p, int c) -> float {
        const int f = p * C + c;
        return (f & 1) ? acc[s][f >> 1].y : acc[s][f >> 1].x;
    };

    typedef uint32_t bgra_raw_t __attribute__((ext_vector_type(PX)));   // one lane's PX source pixels
    struct YccRaw { uint32_t y, cb, cr; };                              // 4 samples of each component plane
    typedef std::conditional_t<YCC, YccRaw, bgra_raw_t> raw_t;
    auto fetch_row = [&](int y) -> raw_t {                               // y is wave-uniform; -1 = nothing needed
        const uint32_t yy = y < 0 ? 0u : static_cast<uint32_t>(y);      // (row 0 is re-read: stays in L2)
        // (pinned to scalar registers: otherwise the lane offset is folded into the base and the row term comes back as
        // a vector 64-bit multiply-add)
        uint64_t rowp = reinterpret_cast<uint64_t>(src) + static_cast<uint64_t>(yy) * a.in_stride;
        asm("" : "+s"(rowp));
        uint32_t lane_off = lane_off0;                                  // (re-made here, in the load's own basic block: instruction
        asm("" : "+v"(lane_off));                                       // selection only then sees "scalar base + 32-bit lane offset")
        // source frames are streamed exactly once: non-temporal loads keep them from displacing the tables in L2
        // (measured -1.7% kernel time, profiles/r1_notes.md)
        // (global address space spelled out: a pointer made from an integer is a flat one, and flat loads count on lgkmcnt too)
        typedef __attribute__((address_space(1))) const uint8_t gbyte;
        typedef __attribute__((address_space(1))) const uint32_t gword;
        typedef __attribute__((address_space(1))) const bgra_raw_t graw;
        gbyte* py = reinterpret_cast<gbyte*>(rowp);
        if constexpr (YCC) {
            // (the other two planes by their wave-uniform distance)
            return YccRaw{__builtin_nontemporal_load(reinterpret_cast<gword*>(py + lane_off)),
                          __builtin_nontemporal_load(reinterpret_cast<gword*>(py + (a.in_cb - a.in) + lane_off)),
                          __builtin_nontemporal_load(reinterpret_cast<gword*>(py + (a.in_cr - a.in) + lane_off))};
        } else {
            return __builtin_nontemporal_load(reinterpret_cast<graw*>(py + lane_off));
        }
    };

    // sample -> working float for the 4 pixels of one 16-byte load (arithmetic contract step 1)
    // The LDS byte address of channel k's table entry, byte_k * (4 << copies_log2) + 4 * copy + table base, is ONE
    // instruction: v_dot4_u32_u8(px, multiplier placed in byte k, lane constant) -- the other three byte products are
    // zero.  (Byte extract + shift-add took two per gather: 12 of the 28 instructions a source pixel cost on cfg5.)
    const uint32_t lut_mul = 4u << a.lut_copies_log2;                       // <= 128: fits the byte operand
    // (the table's LDS address goes into the lane constant as an integer: an address the compiler forms itself costs an
    // extra add of the dynamic-LDS base per gather)
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    const uint32_t lut_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)(smem + L.lut))) + (lut.lane_off << 2);
    auto convert = [&](const raw_t& q, f32x2 (&vv)[NP]) {
        float v[PX][C];
        if constexpr (YCC) {
            // jdcolor.c ycc_rgb_convert with its 16-bit fixed-point tables evaluated in place (as jpeg_color_kernel does:
            // Cr_r = (91881 cr + 32768) >> 16, Cb_b = (116130 cb + 32768) >> 16, g = (-22554 cb - 46802 cr + 32768) >> 16,
            // range-limited), then the same table gathers a BGRA byte would get
            uint32_t ad[PX][3];
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                const int32_t Y = static_cast<int32_t>((q.y >> (8 * p)) & 255u);
                const int32_t cb = static_cast<int32_t>((q.cb >> (8 * p)) & 255u) - 128, cr = static_cast<int32_t>((q.cr >> (8 * p)) & 255u) - 128;
                const int32_t r = Y + ((__mul24(91881, cr) + 32768) >> 16);
                const int32_t g = Y + ((__mul24(-22554, cb) + 32768 + __mul24(-46802, cr)) >> 16);
                const int32_t b = Y + ((__mul24(116130, cb) + 32768) >> 16);
                const int32_t ch[3] = {b, g, r};
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    const int32_t c8 = ch[k] < 0 ? 0 : (ch[k] > 255 ? 255 : ch[k]);
                    ad[p][k] = __umul24(static_cast<uint32_t>(c8), lut_mul) + lut_lane;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int k = 0; k < 3; ++k) v[p][k] = *reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(ad[p][k]));
        } else {
        // all addresses first, then all reads: a gather issued right behind its own address costs wait states
        uint32_t ad[PX][3];
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k) ad[p][k] = __builtin_amdgcn_udot4(q[p], lut_mul << (8 * k), lut_lane, false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const uint32_t px = q[p];
#pragma unroll
            for (int k = 0; k < 3; ++k) v[p][k] = *reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(ad[p][k]));
            if (ALPHA) {
                const float af = static_cast<float>(px >> 24) * (1.0f / 255.0f);
                v[p][0] = v[p][0] * af;
                v[p][1] = v[p][1] * af;
                v[p][2] = v[p][2] * af;
                v[p][C - 1] = af;
            }
        }
        }
#pragma unroll
        for (int i = 0; i < NP; ++i) vv[i] = f32x2{v[(2 * i) / C][(2 * i) % C], v[(2 * i + 1) / C][(2 * i + 1) % C]};
    };

    // ---- horizontal pass of one output row (arithmetic contract step 3: per channel, the strictly ascending fmaf sum
    // over the taps).  Samples come from the row's two planes, weights from the output's de-duplicated row, all as
    // aligned 16-byte LDS reads of one 4-tap group.  A group may begin or end with +0 weights (columns before the first /
    // after the last tap): fmaf(+0, x, h) == h exactly for the finite x staged in LDS and h is never -0, so the padded
    // groups run unpredicated.  The (c0, c1) chains advance together in one v_pk_fma_f32 per tap, likewise (c2, c3).
    auto h_pair_group = [&](f32x2& h, const float4& w, const float4& t01, const float4& t23) {
        h = __builtin_elementwise_fma(f32x2{w.x, w.x}, f32x2{t01.x, t01.y}, h);
        h = __builtin_elementwise_fma(f32x2{w.y, w.y}, f32x2{t01.z, t01.w}, h);
        h = __builtin_elementwise_fma(f32x2{w.z, w.z}, f32x2{t23.x, t23.y}, h);
        h = __builtin_elementwise_fma(f32x2{w.w, w.w}, f32x2{t23.z, t23.w}, h);
    };
    auto h_single_group = [&](float& h, const float4& w, const float4& t) {
        h = __builtin_fmaf(w.x, t.x, h);
        h = __builtin_fmaf(w.y, t.y, h);
        h = __builtin_fmaf(w.z, t.z, h);
        h = __builtin_fmaf(w.w, t.w, h);
    };
    // Mapping 1: one lane per chain group g of an output column (g = 0: the (c0, c1) pair, g = 1: c2 / the (c2, c3)
    // pair); results go through obuf and are encoded by the next row's hand-over.  Used for strips with less than one
    // wave of outputs, where it spreads the (long) chains over twice the lanes.
    const uint32_t n_chain = n_u * 2u;
    const uint32_t n_store = img_on ? n_u : 0u;      // idle frame slots compute nothing they would have to store
    auto h_run_row = [&](const float* vrow, float* orow) {
        for (uint32_t idx = tid; idx < n_chain; idx += T) {
            const uint32_t ul = idx >> 1, g = idx & 1u;
            const uint4 m = hmeta[ul];
            const float4* wp = reinterpret_cast<const float4*>((WLDS ? hw_lds : a.h_wu) + m.z);
            if (g == 0u || ALPHA) {
                const float4* s0 = reinterpret_cast<const float4*>(vrow + (2u * g) * plane_pitch + m.x);
                const float4* s1 = reinterpret_cast<const float4*>(vrow + (2u * g + 1u) * plane_pitch + m.x);
                f32x2 h = {0.0f, 0.0f};
                for (uint32_t q = 0; q < m.y; ++q) h_pair_group(h, wp[q], s0[q], s1[q]);
                *reinterpret_cast<float2*>(orow + ul * 4u + 2u * g) = make_float2(h.x, h.y);
            } else {
                const float4* sp = reinterpret_cast<const float4*>(vrow + 2u * plane_pitch + m.x);
                float h = 0.0f;
                for (uint32_t q = 0; q < m.y; ++q) h_single_group(h, wp[q], sp[q]);
                orow[ul * 4u + 2u] = h;
            }
        }
    };
    // Which lane takes which column: a ds_read_b128 is served in four lane groups, {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31} and
    // the same +32, one LDS cycle each when the 16 lanes of a group hit 16 different 16-byte slots of the 256-byte bank row.
    // With lane = column a group's columns span 28 outputs: at 2.4 source pixels per output (cfg3 level 0, cfg4, cfg1) that
    // is 17 source groups, the first and the last 16 groups apart -- the same slot, an extra cycle on every other read
    // (counters on cfg3 level 0: 31 % of the LDS cycles were bank conflicts, the LDS busy 67 % of the launch).  So the 32
    // columns of a half wave are dealt out group-wise: each lane group gets 16 consecutive columns (a span of 9.6 source
    // groups at that ratio).  A permutation inside aligned blocks of 32 columns: stores still cover whole 128-byte lines,
    // the 4-byte reads (two groups of 32 consecutive lanes) see the same set of addresses; pixels are unchanged.
    uint32_t tid_h = tid;
    if constexpr (PERPIXEL) {
        const uint32_t q = (tid >> 2) & 7u;                          // lanes 0-3, 4-7, ... 28-31 -> columns 0-3, 16-19, 20-23, 4-7, 24-27, 8-11, 12-15, 28-31
        const uint32_t first4 = (0x7326'1540u >> (4u * q)) & 15u;    // column of the quad's first lane, in units of 4
        tid_h = (tid & ~31u) | (first4 << 2) | (tid & 3u);
    }
    // Mapping 2: one lane per output PIXEL (all its channels, then encode + store at once, no obuf round trip): used
    // whenever a strip has at least a wave of outputs.
    constexpr bool h_per_pixel = PERPIXEL;
    auto h_run_row_pixels = [&](uint32_t j, const float* vrow) {
        for (uint32_t ul = tid_h; ul < n_store; ul += T) {
            const uint4 m = hmeta[ul];
            const float4* wp = reinterpret_cast<const float4*>((WLDS ? hw_lds : a.h_wu) + m.z);
            const float4* sp = reinterpret_cast<const float4*>(vrow + m.x);       // sub-plane k at sp + k * (plane_pitch / 4)
            const uint32_t pp4 = plane_pitch >> 2;
            f32x2 h01 = {0.0f, 0.0f}, h23 = {0.0f, 0.0f};
            float h2 = 0.0f;
            // (narrow shapes, two waves per SIMD: the reads of group q + 1 under the chains of group q -- cfg5 2.27 -> 2.18 ms;
            // at four waves per SIMD the other waves cover that latency and the longer code costs: cfg2 1.416 -> 1.428)
            constexpr int HPU = fused_shape(K, C).threads == 512 ? 2 : 1;
#pragma unroll HPU
            for (uint32_t q = 0; q < m.y; ++q) {
                const float4 w = wp[q];
                h_pair_group(h01, w, sp[q], sp[pp4 + q]);
                if (ALPHA) h_pair_group(h23, w, sp[2u * pp4 + q], sp[3u * pp4 + q]);
                else h_single_group(h2, w, sp[2u * pp4 + q]);
            }
            const OutTables<BankedLut, ThresholdL2S> tb{lut, ThresholdL2S{thr, l2s_lds}};
            store_pixel<ALPHA>(a, img, j, strip.u0 + ul, h01.x, h01.y, ALPHA ? h23.x : h2, ALPHA ? h23.y : 1.0f, tb);
        }
    };
    // Mapping 2, fast form: two base addresses per output, everything else immediates.
    auto h_run_row_pixels_fast = [&](uint32_t j, const float* vrow, auto static_encode) {
        constexpr uint32_t G = fast_g > 0 ? fast_g : 1;
        for (uint32_t ul = tid_h; ul < n_store; ul += T) {
            const uint32_t m = hmeta2[ul];
            f32x2 h01 = {0.0f, 0.0f}, h23 = {0.0f, 0.0f};
            float h2 = 0.0f;
            if constexpr (TWO) {
                // Three base addresses the compiler cannot relate to each other: it would fuse two 8-byte reads of one base into
                // ds_read2_b64, which the LDS serves in 8 cycles per wave (two passes of four 16-lane groups) where two
                // ds_read_b64 take 2 + 2 (MI355X_MICROARCH.md, LDS table) -- and the reads are what this pass is bound by.
                typedef __attribute__((address_space(3))) const f32x2 lds_f2;    // (a plain vector type: HIP's float2 has no copy from another address space)
                uint32_t a0 = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)(reinterpret_cast<const unsigned char*>(vrow)))) + (m & 0xffffu) * (GP / 2u);
                uint32_t a1 = a0 + 8u, a2 = a0 + 16u;
                asm volatile("" : "+v"(a0), "+v"(a1), "+v"(a2));
                const float2* wp = reinterpret_cast<const float2*>(hw_lds) + (m >> 16) * G;
#pragma unroll
                for (uint32_t q = 0; q < G; ++q) {
                    const float2 w = wp[q];
                    const f32x2 t0 = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a0 + q * (GP / 2u)));
                    const f32x2 t1 = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a1 + q * (GP / 2u)));
                    const f32x2 c2 = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a2 + q * (GP / 2u)));
                    h01 = __builtin_elementwise_fma(f32x2{w.x, w.x}, f32x2{t0.x, t0.y}, h01);
                    h01 = __builtin_elementwise_fma(f32x2{w.y, w.y}, f32x2{t1.x, t1.y}, h01);
                    h2 = __builtin_fmaf(w.x, c2.x, h2);
                    h2 = __builtin_fmaf(w.y, c2.y, h2);
                    asm volatile("" : "+v"(h01));
                    asm volatile("" : "+v"(h2));
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
            const unsigned char* sp = reinterpret_cast<const unsigned char*>(vrow) + (m & 0xffffu) * GP;
            const float4* wp = reinterpret_cast<const float4*>(hw_lds) + (m >> 16) * G;
#pragma unroll
            for (uint32_t q = 0; q < G; ++q) {
                const float4* g = reinterpret_cast<const float4*>(sp + q * GP);
                const float4 w = wp[q];
                h_pair_group(h01, w, g[0], g[1]);
                if (ALPHA) h_pair_group(h23, w, g[2], g[3]);
                else h_single_group(h2, w, g[2]);
                // One group's operands live at a time.  The empty asm statements pin every chain's state here: without them
                // the optimiser sinks the c2 chain down to its first use (the encode), keeps all G groups' weights and
                // samples alive until then, and the ring accumulators of the vertical pass go to scratch -- whose
                // loads share vmcnt with the source rows in flight.
                asm volatile("" : "+v"(h01));
                if (ALPHA) asm volatile("" : "+v"(h23));
                else asm volatile("" : "+v"(h2));
                __builtin_amdgcn_sched_barrier(0);
            }
            }
            if constexpr (decltype(static_encode)::value) {
                const OutTables<BankedLut, DirectL2S> tbs{lut, DirectL2S{l2s_lds}};
                store_pixel<ALPHA, 1>(a, img, j, strip.u0 + ul, h01.x, h01.y, ALPHA ? h23.x : h2, ALPHA ? h23.y : 1.0f, tbs);
            } else {
                const OutTables<BankedLut, ThresholdL2S> tb{lut, ThresholdL2S{thr, l2s_lds}};
                store_pixel<ALPHA>(a, img, j, strip.u0 + ul, h01.x, h01.y, ALPHA ? h23.x : h2, ALPHA ? h23.y : 1.0f, tb);
            }
        }
    };
    int h_out_row = -1;              // output row whose horizontal result is waiting in obuf (uniform), -1: none
    auto h_store_row = [&](uint32_t j, const float* orow) {      // output stage of a horizontally filtered row
        // the lanes at the top of the workgroup take it: the chains sit on the lowest lanes
        for (uint32_t ul = T - 1u - tid; ul < n_store; ul += T) {
            const float4 o = *reinterpret_cast<const float4*>(orow + ul * 4u);
            const OutTables<BankedLut, ThresholdL2S> tb{lut, ThresholdL2S{thr, l2s_lds}};
            store_pixel<ALPHA>(a, img, j, strip.u0 + ul, o.x, o.y, o.z, ALPHA ? o.w : 1.0f, tb);
        }
    };

    // Software pipeline over steps (one step = one source row):
    //   raw[D]  : D source rows in flight per lane (16 B each), refilled in place -> fixed registers, vmcnt(D-1)
    //   vbuf[2] : (PIPE) converted floats of the current / the next step; the 12-16 LUT reads of step i+1 are issued
    //             before the FMAs of step i, so their LDS latency hides under the lane's own arithmetic
    //   rec[2]  : (PIPE) the 64-byte step records of the current / the next step (scalar loads, same overlap)
    // Register-heavier rings use the plain form (!PIPE): convert, refill, accumulate, one step at a time.
    // The host pads every band to a multiple of D steps, so the unrolled group has no early exit and every buffer
    // index below is a compile-time constant.
    raw_t raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        raw[d] = fetch_row(steps[s0 + d].y);
        __builtin_amdgcn_sched_barrier(0);      // keep issue order raw[0..D-1]: the loop's vmcnt(D-1) relies on it
    }
    f32x2 vbuf[PIPE ? 2 : 1][NP];
    VStep rec[PIPE ? 2 : 1];
    if (PIPE) {
        rec[0] = steps[s0];
        convert(raw[0], vbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
        raw[0] = fetch_row((s0 + D < s1) ? steps[s0 + D].y : -1);
        __builtin_amdgcn_sched_barrier(0);
    }

    for (uint32_t sb = s0; sb < s1; sb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int cur = PIPE ? (d & 1) : 0, nxt = PIPE ? (cur ^ 1) : 0;
            const uint32_t si = sb + d;
            if (PIPE) {
                // ---- stage A: start step si+1 (record, LUT gathers), refill its row slot for step si+1+D ----
                const int slot_next = (d + 1) % D;
                rec[nxt] = steps[(si + 1 < s1) ? si + 1 : si];
                convert(raw[slot_next], vbuf[nxt]);
                __builtin_amdgcn_sched_barrier(0);
                raw[slot_next] = fetch_row(rec[cur].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                rec[0] = steps[si];
                convert(raw[d], vbuf[0]);
                // The bytes of raw[d] are consumed; only now re-issue the load into the same registers (row of step
                // si+D).  Issuing it earlier would overlap the two live ranges and make the compiler rotate the
                // registers with copies (and a vmcnt(0) drain) at the loop back edge.
                __builtin_amdgcn_sched_barrier(0);
                raw[d] = fetch_row(rec[0].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- stage B: finish step si ----
            const VStep& st = rec[cur];
            f32x2 (&v)[NP] = vbuf[PIPE ? cur : 0];
            // Every ring slot accumulates unconditionally: a slot outside its window holds exactly +0.0f (initial
            // value / reset at flush) and has weight +0.0f in the step record, and fmaf(+0, v, +0) == +0 for the
            // finite non-negative v we feed it, so the result is bit-identical to skipping the slot -- without
            // K scalar branches (and their instruction-fetch bubbles) per source row.
            if (st.y >= 0) {
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const float w = st.w[s];
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[s][i] = __builtin_elementwise_fma(f32x2{w, w}, v[i], acc[s][i]);
                }
            }
            if (st.flush_slot >= 0) {
                // ---- output row j's vertical pass is complete: hand its row to the horizontal pass ----
                const uint32_t j = static_cast<uint32_t>(st.out_row);
                float* dst_row = inter + (j & 1u) * inter_stride;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st.flush_slot == s) {
                        if (lane_on) {
                            const uint32_t pp4 = plane_pitch >> 2;
                            if constexpr (FG > 0) {
                                static_assert(FG == 0 || PX == 4, "fast pass: 4 source pixels per lane");
                                float4* g4 = reinterpret_cast<float4*>(reinterpret_cast<unsigned char*>(dst_row) + tid * GP);
                                if constexpr (TWO) {
                                    // two 24-byte groups of two pixels: (c0, c1) of the first, of the second, (c2, c2)
                                    g4[0] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                    g4[1] = make_float4(acc_at(s, 0, 2), acc_at(s, 1, 2), acc_at(s, 2, 0), acc_at(s, 2, 1));
                                    g4[2] = make_float4(acc_at(s, 3, 0), acc_at(s, 3, 1), acc_at(s, 2, 2), acc_at(s, 3, 2));
                                } else {
                                    g4[0] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                    g4[1] = make_float4(acc_at(s, 2, 0), acc_at(s, 2, 1), acc_at(s, 3, 0), acc_at(s, 3, 1));
                                    if (ALPHA) {
                                        g4[2] = make_float4(acc_at(s, 0, 2), acc_at(s, 0, C - 1), acc_at(s, 1, 2), acc_at(s, 1, C - 1));
                                        g4[3] = make_float4(acc_at(s, 2, 2), acc_at(s, 2, C - 1), acc_at(s, 3, 2), acc_at(s, 3, C - 1));
                                    } else {
                                        g4[2] = make_float4(acc_at(s, 0, 2), acc_at(s, 1, 2), acc_at(s, 2, 2), acc_at(s, 3, 2));
                                    }
                                }
                            } else if constexpr (PX == 4) {
                                float4* g4 = reinterpret_cast<float4*>(dst_row + 4u * tid);
                                g4[0] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                g4[pp4] = make_float4(acc_at(s, 2, 0), acc_at(s, 2, 1), acc_at(s, 3, 0), acc_at(s, 3, 1));
                                if (ALPHA) {
                                    g4[2u * pp4] = make_float4(acc_at(s, 0, 2), acc_at(s, 0, C - 1), acc_at(s, 1, 2), acc_at(s, 1, C - 1));
                                    g4[3u * pp4] = make_float4(acc_at(s, 2, 2), acc_at(s, 2, C - 1), acc_at(s, 3, 2), acc_at(s, 3, C - 1));
                                } else {
                                    g4[2u * pp4] = make_float4(acc_at(s, 0, 2), acc_at(s, 1, 2), acc_at(s, 2, 2), acc_at(s, 3, 2));
                                }
                            } else {
                                // 2 pixels per lane: lane g holds pixels 2g, 2g+1 = half (g & 1) of 4-pixel group g >> 1
                                const uint32_t half = tid & 1u;
                                float4* g4 = reinterpret_cast<float4*>(dst_row + 4u * (tid >> 1));
                                g4[half * pp4] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                if (ALPHA) g4[(2u + half) * pp4] = make_float4(acc_at(s, 0, 2), acc_at(s, 0, C - 1), acc_at(s, 1, 2), acc_at(s, 1, C - 1));
                                else *reinterpret_cast<float2*>(dst_row + 2u * plane_pitch + 2u * tid) = make_float2(acc_at(s, 0, 2), acc_at(s, 1, 2));
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[s][i] = f32x2{0.0f, 0.0f};
                    }
                }
                // One barrier per output row.  After it: row j's vertical result (inter[j&1]) and row j-1's horizontal
                // result (obuf[(j-1)&1]) are complete.  inter[j&1] is next written at row j+2 and obuf[(j-1)&1] by
                // row j+1's chains, both only after every wave has passed the barrier of row j+1, i.e. after every
                // wave has finished reading them.
                lds_barrier();
                if constexpr (h_per_pixel) {
                    if constexpr (FG > 0) {
                        // one test per output ROW for "linear working space, encode table staged": that copy of the pixel loop
                        // runs without the per-channel tests, its three table reads in flight together
                        if (a.linear && l2s_lds != nullptr) h_run_row_pixels_fast(j, dst_row, std::true_type{});
                        else h_run_row_pixels_fast(j, dst_row, std::false_type{});
                    }
                    else h_run_row_pixels(j, dst_row);
                } else {
                    if (h_out_row >= 0) h_store_row(static_cast<uint32_t>(h_out_row), obuf + (static_cast<uint32_t>(h_out_row) & 1u) * obuf_stride);
                    h_out_row = static_cast<int>(j);
                    h_run_row(dst_row, obuf + (j & 1u) * obuf_stride);
                }
            }
        }
    }
    // drain: the last output row of the band still has its horizontal pass and output stage to do
    if (h_out_row >= 0) {
        lds_barrier();
        h_store_row(static_cast<uint32_t>(h_out_row), obuf + (static_cast<uint32_t>(h_out_row) & 1u) * obuf_stride);
    }
}


template <int K, bool ALPHA, bool WLDS, bool PERPIXEL, int FG = 0, bool YCC = false>
static hipError_t launch_variant(const ResampleArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    // raise the dynamic-LDS cap once per kernel variant and device (it is sticky), not on every launch: a bit per device
    // ordinal (ordinals beyond the mask set the attribute on every launch); failing to raise it is the launch's error
    static std::atomic<uint64_t> raised;     // (one per instantiation of this function template)
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&fused_resample_kernel<K, ALPHA, WLDS, PERPIXEL, FG, YCC>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsCap));
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((fused_resample_kernel<K, ALPHA, WLDS, PERPIXEL, FG, YCC>), grid, block, lds, st, a, a.steps);
    return hipGetLastError();
}

#define IFHIP_CAT2(a, b) a##b
#define IFHIP_CAT(a, b) IFHIP_CAT2(a, b)

// launch_fused_k<K>: picks the (alpha, weights-in-LDS, per-pixel) instantiation.  K must equal the ring size exactly:
// every slot accumulates on every row.
hipError_t IFHIP_CAT(launch_fused_k, IFHIP_FUSED_K)(const ResampleArgs& a, bool alpha, bool per_pixel, dim3 grid, dim3 block,
                                                    size_t lds, hipStream_t st) {
    constexpr int K = IFHIP_FUSED_K;
    if (a.ycc) {                               // planar YCbCr source: no alpha, weights in LDS, one lane per pixel (host guarantees all three)
        if constexpr (fused_shape(K, 3).px == 4) {
            if (!alpha && a.h_w_in_lds && per_pixel) {
                switch (a.h_groups) {
                case 0: return launch_variant<K, false, true, true, 0, true>(a, grid, block, lds, st);
                case 2: return launch_variant<K, false, true, true, 2, true>(a, grid, block, lds, st);
                case 3: return launch_variant<K, false, true, true, 3, true>(a, grid, block, lds, st);
                case 4: return launch_variant<K, false, true, true, 4, true>(a, grid, block, lds, st);
                default: break;
                }
            }
        }
        return hipErrorInvalidValue;
    }
    if (a.h_groups) {                          // fast horizontal pass: weights in LDS, one lane per pixel (host guarantees both)
        if constexpr (fused_shape(K, 3).px == 4 && fused_shape(K, 4).px == 4) {
            switch ((alpha ? 8 : 0) | a.h_groups) {
            case 2: return launch_variant<K, false, true, true, 2>(a, grid, block, lds, st);
            case 3: return launch_variant<K, false, true, true, 3>(a, grid, block, lds, st);
            case 4: return launch_variant<K, false, true, true, 4>(a, grid, block, lds, st);
            case 18: return launch_variant<K, false, true, true, 18>(a, grid, block, lds, st);     // 16 + G2: two-column groups
            case 19: return launch_variant<K, false, true, true, 19>(a, grid, block, lds, st);
            case 20: return launch_variant<K, false, true, true, 20>(a, grid, block, lds, st);
            case 21: return launch_variant<K, false, true, true, 21>(a, grid, block, lds, st);
            case 22: return launch_variant<K, false, true, true, 22>(a, grid, block, lds, st);
            case 10: return launch_variant<K, true, true, true, 2>(a, grid, block, lds, st);
            case 11: return launch_variant<K, true, true, true, 3>(a, grid, block, lds, st);
            case 12: return launch_variant<K, true, true, true, 4>(a, grid, block, lds, st);
            default: break;
            }
        }
        return hipErrorInvalidValue;
    }
    const bool wl = a.h_w_in_lds != 0;
    const int sel = (alpha ? 4 : 0) | (wl ? 2 : 0) | (per_pixel ? 1 : 0);
    switch (sel) {
    case 0: return launch_variant<K, false, false, false>(a, grid, block, lds, st);
    case 1: return launch_variant<K, false, false, true>(a, grid, block, lds, st);
    case 2: return launch_variant<K, false, true, false>(a, grid, block, lds, st);
    case 3: return launch_variant<K, false, true, true>(a, grid, block, lds, st);
    case 4: return launch_variant<K, true, false, false>(a, grid, block, lds, st);
    case 5: return launch_variant<K, true, false, true>(a, grid, block, lds, st);
    case 6: return launch_variant<K, true, true, false>(a, grid, block, lds, st);
    default: return launch_variant<K, true, true, true>(a, grid, block, lds, st);
    }
}

}  // namespace ifhip
