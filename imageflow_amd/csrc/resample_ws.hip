// resample_ws.hip -- the fused resample + render kernel for MODERATE ratios, wave-specialised, for ONE ring size K
// (compiled once per K = 1..5 with -DIFHIP_FUSED_K=K, in parallel; see imageflow_amd/build.py).
//
// Same arithmetic as resample_fused.hip (graphics/scaling.rs:19-90 behind it: sample -> working float, color.rs:22-45;
// vertical then horizontal weighted convolution from PixelRowWeights tables, weights.rs:521-571,681-788; the ReplaceSelf
// output stage, scaling.rs:211-251) -- same taps in the same order, so the same pixels bit for bit -- on a different
// division of labour.  At moderate ratios (1.3 - 4 source pixels per output: every export_4_sizes level, the JPEG chain's
// 1920 -> 800) an output row costs as much in its horizontal pass (LDS reads, chains, encode, store) as in the vertical
// steps that feed it, and in resample_fused.hip the SAME waves run the two one after the other behind one workgroup barrier
// per output row: the vertical steps wait on HBM while the LDS and the VALUs idle, then the pixel loop runs while nothing
// is requested from HBM (profiles/NOTEBOOK.md: level 0 of cfg3 = 1.30 ms of vertical steps + 0.82 ms of pixel loop).
// Here a workgroup has two kinds of waves:
//   * V waves (the first frames_per_wg * lanes_per_frame / 64): stream source rows (16 B per lane, D rows in flight, never
//     interrupted), convert through the banked sRGB table, accumulate the K live output rows in registers, and publish each
//     finished row into a ring of R row slots in LDS (per frame slot) -- then carry on with the next source row;
//   * H waves (the rest): take (output row, frame slot, 64-column chunk) units in order, wait for the row's slot to be
//     complete, run the fast horizontal pass for their 64 outputs, encode, store, and hand the slot back.
// No workgroup barrier after the tables are staged: a slot's hand-over is two monotonic LDS counters (rows published by
// the V waves / chunks consumed by the H waves), polled with s_sleep by whoever is early.  Every spin is bounded: a wave
// that waits "forever" (a bug, not a schedule) sets the canvas' first word of its frame to a marker pattern nobody can
// mistake for a pass and carries on, so that a test fails instead of a box hanging.
// Bound: HBM (a 1-D stencil per axis: no MFMA).  Build with -ffp-contract=off, as resample_fused.hip.
#include <atomic>
#include <type_traits>

#include "resample_device.hpp"

#ifndef IFHIP_FUSED_K
#error "compile with -DIFHIP_FUSED_K=<ring size 1..5>"
#endif

namespace ifhip {

typedef __attribute__((address_space(3))) uint32_t lds_u32;

// The hand-over protocol rests on one property of the LDS: it executes a wave's DS instructions in the order the wave issued
// them.  A V wave's row stores followed by its counter increment therefore need no wait in between (whoever sees the count
// sees the row), nor do an H wave's sample reads followed by ITS increment; what must not happen is the compiler moving
// DS instructions across the counter operations -- the empty asm statements with a memory clobber.
//
// wait until *p >= need (a counter only ever grows); wave-uniform result: false = gave up (see the kernel's header)
__device__ __forceinline__ bool ws_wait_ge(uint32_t* p, uint32_t need) {
    bool ok = false;
    for (uint32_t spin = 0; spin < (1u << 19); ++spin) {        // ~50 ms: three orders of magnitude above any real wait
        const uint32_t v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
        if (v >= need) { ok = true; break; }
        __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("" ::: "memory");
    return ok;
}
__device__ __forceinline__ void ws_signal(uint32_t* p) {
    asm volatile("" ::: "memory");
    if ((threadIdx.x & 63u) == 0u) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// FG: the fast horizontal pass' form, as in resample_fused.hip -- 2..4: that many 4-tap groups per output; 16 + G2: G2
// groups of TWO source columns (BGRA sources).  YCC: planar source (the JPEG stage's component planes).
// D source rows in flight per V lane; PIPE: converted samples double buffered (the table gathers of step i + 1 under the
// multiply-adds of step i).
template <int K, int FG, bool YCC, int D, bool PIPE>
__global__ void __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4)))
ws_resample_kernel(const ResampleArgs a, const VStep* __restrict__ steps) {
    constexpr bool TWO = FG >= 16;
    static_assert(!TWO || !YCC, "two-column groups: BGRA sources");
    static_assert(fused_shape(K, 3).px == 4 && fused_shape(K, 3).threads == 1024, "rings that leave room for 4 pixels per lane at 4 waves per SIMD");
    constexpr int C = 3;
    constexpr int PX = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const uint32_t wtid = threadIdx.x;
    const uint32_t WT = blockDim.x;
    const uint32_t T = a.lanes_per_frame;                  // V lanes per frame slot (whole waves)
    const uint32_t F = a.frames_per_wg;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(wtid >> 6);
    const uint32_t wpf = T >> 6;                           // V waves per frame slot
    const uint32_t n_v = F * wpf;                          // V waves; the rest are H waves
    const uint32_t n_h = (WT >> 6) - n_v;

    uint32_t b = blockIdx.x;                               // strips of one frame on one XCD (see resample_fused.hip)
    if (a.n_strips > 1u && (gridDim.x & 7u) == 0u) b = (b & 7u) * (gridDim.x >> 3) + (b >> 3);
    const uint32_t strip_i = b % a.n_strips; b /= a.n_strips;
    const uint32_t band = b % a.n_bands;
    const uint32_t img0 = (b / a.n_bands) * F;             // first frame of this workgroup
    const uint32_t f_on = min(F, a.n_images - img0);       // frame slots with a frame (the last workgroup may have idle ones)

    const Strip strip = a.strips[strip_i];
    const uint32_t n_u = strip.u1 - strip.u0;
    const uint32_t chunks = (n_u + 63u) >> 6;              // 64-output chunks per row
    const uint32_t units = (chunks + kWsUnitChunks - 1u) / kWsUnitChunks;   // what an H wave takes at a time (and a row hands back in)
    const uint32_t R = a.ws_ring;

    constexpr uint32_t fast_g = TWO ? FG - 16 : FG;
    constexpr uint32_t GP = fused_group_pitch(C);
    const WsLds L = ws_lds_layout(n_u, strip.nquads, a.h_wu_floats, a.l2s_in_lds != 0, a.lut_copies_log2, F, fast_g, R);
    float* lut_banked = reinterpret_cast<float*>(smem + L.lut);
    const float* hw_lds = reinterpret_cast<const float*>(smem + L.hw);
    uint32_t* hmeta2 = reinterpret_cast<uint32_t*>(smem + L.hmeta);
    uint32_t* vcnt = reinterpret_cast<uint32_t*>(smem + L.sync);         // [F][R] V waves that published the slot's row, ever
    uint32_t* hcnt = vcnt + F * R;                                       // [F][R] units consumed from the slot, ever

    for (uint32_t i = wtid; i < (256u << a.lut_copies_log2); i += WT) lut_banked[i] = a.lut_in[i >> a.lut_copies_log2];
    const uint8_t* l2s_lds = a.l2s_in_lds ? smem + L.l2s : nullptr;
    if (a.l2s_in_lds)
        for (uint32_t i = wtid; i < 1024u; i += WT)
            reinterpret_cast<uint4*>(smem + L.l2s)[i] = reinterpret_cast<const uint4*>(a.l2s)[i];
    for (uint32_t i = wtid; i < n_u; i += WT) hmeta2[i] = a.h_meta2[strip.u0 + i] - (strip.cx0 >> (TWO ? 1 : 2));
    {   // the groups past the staged columns are read (with weight +0) and never written: they must hold finite values
        float4* z = reinterpret_cast<float4*>(smem + L.ring);
        for (uint32_t i = wtid; i < (F * R * L.row_stride) >> 4; i += WT) z[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t i = wtid; i < 2u * F * R; i += WT) vcnt[i] = 0u;
        const float4* src4 = reinterpret_cast<const float4*>(a.h_wu);
        float4* dst4 = reinterpret_cast<float4*>(smem + L.hw);
        for (uint32_t i = wtid; i < (a.h_wu_floats >> 2); i += WT) dst4[i] = src4[i];
    }
    __syncthreads();

    // rows of this band: [j0, j1), flushed by the schedule in ascending order
    const uint32_t j0 = static_cast<uint32_t>(static_cast<uint64_t>(a.out_h) * band / a.n_bands);
    const uint32_t j1 = static_cast<uint32_t>(static_cast<uint64_t>(a.out_h) * (band + 1u) / a.n_bands);
    typedef __attribute__((address_space(3))) const float lds_cfloat;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    // a wait that ran out (see the header): the frame's first canvas word becomes a marker (an untracked store, like every
    // canvas store: the V waves' load counter must not see it)
    auto mark_stuck = [&](uint32_t im) {
        // ... and every counter jumps ahead so that no later wait of this workgroup spins again: the launch ends at once
        for (uint32_t i = 0; i < 2u * F * R; ++i) __hip_atomic_fetch_add(vcnt + i, 0x40000000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        store_u32_untracked(reinterpret_cast<uint32_t*>(a.canvas + static_cast<size_t>(im) * a.canvas_image_bytes
                                                        + static_cast<size_t>(a.y) * a.c_stride + static_cast<size_t>(a.x) * 4u), 0xDEADBEEFu);
    };

    if (wave >= n_v) {
        // =================================================== H waves ===================================================
        const uint32_t h = wave - n_v, lane = wtid & 63u;
        // output columns dealt to the lane groups of the 16-byte LDS read (resample_fused.hip, "Which lane takes which column")
        const uint32_t q8 = (lane >> 2) & 7u;
        const uint32_t first4 = (0x7326'1540u >> (4u * q8)) & 15u;
        const uint32_t lane_h = (lane & ~31u) | (first4 << 2) | (lane & 3u);
        const BankedLut lut{lut_banked, wtid & ((1u << a.lut_copies_log2) - 1u), a.lut_copies_log2};
        const bool static_encode = a.linear && l2s_lds != nullptr;
        // A unit = U chunks of 64 consecutive outputs of one row, one output of each chunk per lane: the U chains' LDS round
        // trips and multiply-adds overlap inside the wave (one chunk at a time left an H wave waiting out its own latencies:
        // 2 200 cycles per 64 outputs, the launch H-bound at any number of H waves).  `NK` <= U chunks of a row's last unit exist.
        uint32_t c = h, f = 0u, r = 0u, s = 0u, u = 0u;          // unit (row r of the band, frame slot f, unit c of the row); slot s = r % R, use u = r / R
        const uint32_t rows = j1 - j0;
        auto run_unit = [&](auto nk_const, uint32_t img, uint32_t j, const unsigned char* vrow, uint32_t* done, const uint32_t (&m)[kWsUnitChunks]) {
            constexpr uint32_t NK = decltype(nk_const)::value;
            constexpr uint32_t G = fast_g;
            uint32_t ul[NK];
            bool on[NK];
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) {
                ul[k] = ((c * kWsUnitChunks + k) << 6) + lane_h;
                on[k] = ul[k] < n_u;
            }
            f32x2 h01[NK];
            float h2[NK];
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) { h01[k] = f32x2{0.0f, 0.0f}; h2[k] = 0.0f; }
            if constexpr (TWO) {
                // (three base addresses the compiler cannot relate: two 8-byte reads of one base would be fused into ds_read2_b64,
                // 8 LDS cycles per wave where two ds_read_b64 take 2 + 2)
                typedef __attribute__((address_space(3))) const f32x2 lds_f2;
                uint32_t a0[NK], a1[NK], a2[NK];
                const float2* wp[NK];
#pragma unroll
                for (uint32_t k = 0; k < NK; ++k) {
                    a0[k] = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)(vrow))) + (m[k] & 0xffffu) * (GP / 2u);
                    a1[k] = a0[k] + 8u; a2[k] = a0[k] + 16u;
                    asm volatile("" : "+v"(a0[k]), "+v"(a1[k]), "+v"(a2[k]));
                    wp[k] = reinterpret_cast<const float2*>(hw_lds) + (m[k] >> 16) * G;
                }
#pragma unroll
                for (uint32_t q = 0; q < G; ++q) {
                    float2 w[NK];
                    f32x2 t0[NK], t1[NK], c2[NK];
#pragma unroll
                    for (uint32_t k = 0; k < NK; ++k) {
                        w[k] = wp[k][q];
                        t0[k] = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a0[k] + q * (GP / 2u)));
                        t1[k] = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a1[k] + q * (GP / 2u)));
                        c2[k] = *reinterpret_cast<lds_f2*>(static_cast<uintptr_t>(a2[k] + q * (GP / 2u)));
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NK; ++k) {
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].x, w[k].x}, f32x2{t0[k].x, t0[k].y}, h01[k]);
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].y, w[k].y}, f32x2{t1[k].x, t1[k].y}, h01[k]);
                        h2[k] = __builtin_fmaf(w[k].x, c2[k].x, h2[k]);
                        h2[k] = __builtin_fmaf(w[k].y, c2[k].y, h2[k]);
                    }
                }
            } else {
#pragma unroll
                for (uint32_t q = 0; q < G; ++q) {
                    float4 w[NK], t01[NK], t23[NK], t2[NK];
#pragma unroll
                    for (uint32_t k = 0; k < NK; ++k) {
                        const float4* g = reinterpret_cast<const float4*>(vrow + (m[k] & 0xffffu) * GP + q * GP);
                        w[k] = (reinterpret_cast<const float4*>(hw_lds) + (m[k] >> 16) * G)[q];
                        t01[k] = g[0]; t23[k] = g[1]; t2[k] = g[2];
                    }
#pragma unroll
                    for (uint32_t k = 0; k < NK; ++k) {
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].x, w[k].x}, f32x2{t01[k].x, t01[k].y}, h01[k]);
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].y, w[k].y}, f32x2{t01[k].z, t01[k].w}, h01[k]);
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].z, w[k].z}, f32x2{t23[k].x, t23[k].y}, h01[k]);
                        h01[k] = __builtin_elementwise_fma(f32x2{w[k].w, w[k].w}, f32x2{t23[k].z, t23[k].w}, h01[k]);
                        h2[k] = __builtin_fmaf(w[k].x, t2[k].x, h2[k]);
                        h2[k] = __builtin_fmaf(w[k].y, t2[k].y, h2[k]);
                        h2[k] = __builtin_fmaf(w[k].z, t2[k].z, h2[k]);
                        h2[k] = __builtin_fmaf(w[k].w, t2[k].w, h2[k]);
                    }
                }
            }
            // the row's samples are in registers: the slot may go back as soon as every unit says so
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) asm volatile("" : "+v"(h01[k]), "+v"(h2[k]));
            ws_signal(done);
#pragma unroll
            for (uint32_t k = 0; k < NK; ++k) {
                if (on[k]) {
                    if (static_encode) {
                        const OutTables<BankedLut, DirectL2S> tbs{lut, DirectL2S{l2s_lds}};
                        store_pixel<false, 1>(a, img, j, strip.u0 + ul[k], h01[k].x, h01[k].y, h2[k], 1.0f, tbs);
                    } else {
                        const OutTables<BankedLut, ThresholdL2S> tb{lut, ThresholdL2S{nullptr, l2s_lds}};
                        store_pixel<false>(a, img, j, strip.u0 + ul[k], h01[k].x, h01[k].y, h2[k], 1.0f, tb);
                    }
                }
            }
        };
        for (;;) {
            while (c >= units) { c -= units; if (++f == f_on) { f = 0u; ++r; if (++s == R) { s = 0u; ++u; } } }
            if (r >= rows) break;
            const uint32_t img = img0 + f, j = j0 + r;
            uint32_t m[kWsUnitChunks];                           // the outputs' records do not depend on the row: requested before the wait
#pragma unroll
            for (uint32_t k = 0; k < kWsUnitChunks; ++k) {
                const uint32_t ulk = ((c * kWsUnitChunks + k) << 6) + lane_h;
                m[k] = hmeta2[ulk < n_u ? ulk : 0u];
            }
            if (!ws_wait_ge(vcnt + f * R + s, wpf * (u + 1u)) && lane == 0u) mark_stuck(img);
            const unsigned char* vrow = smem + L.ring + (f * R + s) * L.row_stride;
            const uint32_t nk = min(kWsUnitChunks, chunks - c * kWsUnitChunks);      // chunks of this unit (the row's last one may be short)
            static_assert(kWsUnitChunks == 4, "one case per unit size");
            switch (nk) {
            case 4: run_unit(std::integral_constant<uint32_t, 4>{}, img, j, vrow, hcnt + f * R + s, m); break;
            case 3: run_unit(std::integral_constant<uint32_t, 3>{}, img, j, vrow, hcnt + f * R + s, m); break;
            case 2: run_unit(std::integral_constant<uint32_t, 2>{}, img, j, vrow, hcnt + f * R + s, m); break;
            default: run_unit(std::integral_constant<uint32_t, 1>{}, img, j, vrow, hcnt + f * R + s, m); break;
            }
            c += n_h;
        }
        return;
    }

    // ======================================================= V waves =======================================================
    const uint32_t slot = __builtin_amdgcn_readfirstlane(wtid / T);
    if (slot >= f_on) return;                              // no frame for this slot: nobody waits for its rows either
    const uint32_t tid = wtid - slot * T;
    const uint32_t img = img0 + slot;
    const uint32_t s0 = a.band_begin[band], s1 = a.band_begin[band + 1];
    const uint32_t n_groups = strip.nquads;
    const bool lane_on = tid < n_groups;
    const uint32_t quad = lane_on ? tid : n_groups - 1u;   // lanes past the strip re-read its last quad: every row load is unconditional
    constexpr uint32_t BPP = YCC ? 1u : 4u;
    const uint32_t img_u = __builtin_amdgcn_readfirstlane(img);
    const uint8_t* src = a.in + (static_cast<size_t>(img_u) * a.in_image_bytes + static_cast<size_t>(strip.cx0) * BPP);
    const uint32_t lane_off0 = static_cast<uint32_t>(PX) * quad * BPP;

    constexpr int NP = PX * C / 2;
    f32x2 acc[K][NP];
#pragma unroll
    for (int s = 0; s < K; ++s)
#pragma unroll
        for (int i = 0; i < NP; ++i) acc[s][i] = f32x2{0.0f, 0.0f};
    auto acc_at = [&](int s, int p, int c) -> float {
        const int f = p * C + c;
        return (f & 1) ? acc[s][f >> 1].y : acc[s][f >> 1].x;
    };

    typedef uint32_t bgra_raw_t __attribute__((ext_vector_type(PX)));
    struct YccRaw { uint32_t y, cb, cr; };
    typedef std::conditional_t<YCC, YccRaw, bgra_raw_t> raw_t;
    auto fetch_row = [&](int y) -> raw_t {                               // y is wave-uniform; -1 = nothing needed (row 0 again: stays in L2)
        const uint32_t yy = y < 0 ? 0u : static_cast<uint32_t>(y);
        uint64_t rowp = reinterpret_cast<uint64_t>(src) + static_cast<uint64_t>(yy) * a.in_stride;
        asm("" : "+s"(rowp));                                           // scalar base + 32-bit lane offset (see resample_fused.hip)
        uint32_t lane_off = lane_off0;
        asm("" : "+v"(lane_off));
        typedef __attribute__((address_space(1))) const uint8_t gbyte;
        typedef __attribute__((address_space(1))) const uint32_t gword;
        typedef __attribute__((address_space(1))) const bgra_raw_t graw;
        gbyte* py = reinterpret_cast<gbyte*>(rowp);
        if constexpr (YCC) {
            return YccRaw{__builtin_nontemporal_load(reinterpret_cast<gword*>(py + lane_off)),
                          __builtin_nontemporal_load(reinterpret_cast<gword*>(py + (a.in_cb - a.in) + lane_off)),
                          __builtin_nontemporal_load(reinterpret_cast<gword*>(py + (a.in_cr - a.in) + lane_off))};
        } else {
            return __builtin_nontemporal_load(reinterpret_cast<graw*>(py + lane_off));
        }
    };

    const uint32_t lut_mul = 4u << a.lut_copies_log2;
    const uint32_t lut_lane = static_cast<uint32_t>(reinterpret_cast<uintptr_t>((lds_byte*)(smem + L.lut))) + ((wtid & ((1u << a.lut_copies_log2) - 1u)) << 2);
    auto convert = [&](const raw_t& q, f32x2 (&vv)[NP]) {             // sample -> working float (arithmetic contract step 1)
        float v[PX][C];
        uint32_t ad[PX][3];
        if constexpr (YCC) {
            // jdcolor.c ycc_rgb_convert, 16-bit fixed point (as jpeg_color_kernel), then the table gathers a BGRA byte would get
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
        } else {
            // the LDS byte address of channel k's table entry is one v_dot4_u32_u8 (see resample_fused.hip)
#pragma unroll
            for (int p = 0; p < PX; ++p)
#pragma unroll
                for (int k = 0; k < 3; ++k) ad[p][k] = __builtin_amdgcn_udot4(q[p], lut_mul << (8 * k), lut_lane, false);
        }
        __builtin_amdgcn_sched_barrier(0);                             // all addresses first, then all reads
#pragma unroll
        for (int p = 0; p < PX; ++p)
#pragma unroll
            for (int k = 0; k < 3; ++k) v[p][k] = *reinterpret_cast<lds_cfloat*>(static_cast<uintptr_t>(ad[p][k]));
#pragma unroll
        for (int i = 0; i < NP; ++i) vv[i] = f32x2{v[(2 * i) / C][(2 * i) % C], v[(2 * i + 1) / C][(2 * i + 1) % C]};
    };

    raw_t raw[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        raw[d] = fetch_row(steps[s0 + d].y);
        __builtin_amdgcn_sched_barrier(0);      // keep issue order raw[0..D-1]: the loop's vmcnt(D-1) relies on it
    }
    f32x2 vbuf[PIPE ? 2 : 1][NP];
    VStep rec[2];                               // the step records arrive a step ahead (scalar loads share lgkmcnt with the LDS reads)
    rec[0] = steps[s0];
    if (PIPE) {
        convert(raw[0], vbuf[0]);
        __builtin_amdgcn_sched_barrier(0);
        raw[0] = fetch_row((s0 + D < s1) ? steps[s0 + D].y : -1);
        __builtin_amdgcn_sched_barrier(0);
    }
    uint32_t ring_s = 0u, ring_u = 0u;          // slot of the next row to publish, and how often it has been used before
    unsigned char* ring_f = smem + L.ring + slot * R * L.row_stride;
    uint32_t* vcnt_f = vcnt + slot * R;
    uint32_t* hcnt_f = hcnt + slot * R;

    for (uint32_t sb = s0; sb < s1; sb += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            static_assert(D % 2 == 0, "the record's double buffer alternates with the step's parity");
            const int cur = d & 1, nxt = cur ^ 1;
            const uint32_t si = sb + d;
            // The next step's record is requested BEHIND this step's table gathers: scalar loads and LDS reads share lgkmcnt and
            // return out of order, so the wait for the gathers is a wait for every scalar load issued before it -- requested in
            // front of them (as the one-role kernel does, where four waves per SIMD cover it) a record's trip to the scalar
            // cache or L2 would sit in every step's dependent chain.  Behind them it has the step's multiply-adds to arrive.
            if (PIPE) {
                const int slot_next = (d + 1) % D;
                convert(raw[slot_next], vbuf[nxt]);
                __builtin_amdgcn_sched_barrier(0);
                rec[nxt] = steps[(si + 1 < s1) ? si + 1 : si];
                raw[slot_next] = fetch_row(rec[cur].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            } else {
                convert(raw[d], vbuf[0]);
                __builtin_amdgcn_sched_barrier(0);
                rec[nxt] = steps[(si + 1 < s1) ? si + 1 : si];
                raw[d] = fetch_row(rec[cur].y_ahead);
                __builtin_amdgcn_sched_barrier(0);
            }
            const VStep& st = rec[cur];
            f32x2 (&v)[NP] = vbuf[PIPE ? cur : 0];
            // every ring slot accumulates unconditionally: +0 weights on +0 accumulators are exact (see resample_fused.hip)
            if (st.y >= 0) {
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    const float w = st.w[s];
#pragma unroll
                    for (int i = 0; i < NP; ++i) acc[s][i] = __builtin_elementwise_fma(f32x2{w, w}, v[i], acc[s][i]);
                }
            }
            if (st.flush_slot >= 0) {
                // ---- an output row's vertical pass is complete: publish it ----
                if (ring_u != 0u && !ws_wait_ge(hcnt_f + ring_s, units * ring_u) && tid == 0u) mark_stuck(img);
                unsigned char* dst_row = ring_f + ring_s * L.row_stride;
#pragma unroll
                for (int s = 0; s < K; ++s) {
                    if (st.flush_slot == s) {
                        if (lane_on) {
                            float4* g4 = reinterpret_cast<float4*>(dst_row + tid * GP);
                            if constexpr (TWO) {
                                // two 24-byte groups of two pixels: (c0, c1) of the first, of the second, (c2, c2)
                                g4[0] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                g4[1] = make_float4(acc_at(s, 0, 2), acc_at(s, 1, 2), acc_at(s, 2, 0), acc_at(s, 2, 1));
                                g4[2] = make_float4(acc_at(s, 3, 0), acc_at(s, 3, 1), acc_at(s, 2, 2), acc_at(s, 3, 2));
                            } else {
                                g4[0] = make_float4(acc_at(s, 0, 0), acc_at(s, 0, 1), acc_at(s, 1, 0), acc_at(s, 1, 1));
                                g4[1] = make_float4(acc_at(s, 2, 0), acc_at(s, 2, 1), acc_at(s, 3, 0), acc_at(s, 3, 1));
                                g4[2] = make_float4(acc_at(s, 0, 2), acc_at(s, 1, 2), acc_at(s, 2, 2), acc_at(s, 3, 2));
                            }
                        }
#pragma unroll
                        for (int i = 0; i < NP; ++i) acc[s][i] = f32x2{0.0f, 0.0f};
                    }
                }
                ws_signal(vcnt_f + ring_s);
                if (++ring_s == R) { ring_s = 0u; ++ring_u; }
            }
        }
    }
}


template <int K, int FG, bool YCC>
static hipError_t launch_ws_variant(const ResampleArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    constexpr int D = kWsRowsInFlight;
    constexpr bool PIPE = false;
    static std::atomic<uint64_t> raised;     // the dynamic-LDS cap is sticky per kernel and device: raised once for each
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const uint64_t bit = dev >= 0 && dev < 64 ? 1ull << dev : 0ull;
    if (!(raised.load(std::memory_order_relaxed) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&ws_resample_kernel<K, FG, YCC, D, PIPE>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(kFusedLdsCap));
        if (e != hipSuccess) return e;
        raised.fetch_or(bit, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((ws_resample_kernel<K, FG, YCC, D, PIPE>), grid, block, lds, st, a, a.steps);
    return hipGetLastError();
}

#define IFHIP_CAT2(a, b) a##b
#define IFHIP_CAT(a, b) IFHIP_CAT2(a, b)

hipError_t IFHIP_CAT(launch_ws_k, IFHIP_FUSED_K)(const ResampleArgs& a, dim3 grid, dim3 block, size_t lds, hipStream_t st) {
    constexpr int K = IFHIP_FUSED_K;
    if (a.ycc) {
        switch (a.h_groups) {
        case 2: return launch_ws_variant<K, 2, true>(a, grid, block, lds, st);
        case 3: return launch_ws_variant<K, 3, true>(a, grid, block, lds, st);
        case 4: return launch_ws_variant<K, 4, true>(a, grid, block, lds, st);
        default: return hipErrorInvalidValue;
        }
    }
    switch (a.h_groups) {
    case 2: return launch_ws_variant<K, 2, false>(a, grid, block, lds, st);
    case 3: return launch_ws_variant<K, 3, false>(a, grid, block, lds, st);
    case 4: return launch_ws_variant<K, 4, false>(a, grid, block, lds, st);
    case 18: return launch_ws_variant<K, 18, false>(a, grid, block, lds, st);
    case 19: return launch_ws_variant<K, 19, false>(a, grid, block, lds, st);
    case 20: return launch_ws_variant<K, 20, false>(a, grid, block, lds, st);
    case 21: return launch_ws_variant<K, 21, false>(a, grid, block, lds, st);
    case 22: return launch_ws_variant<K, 22, false>(a, grid, block, lds, st);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace ifhip
