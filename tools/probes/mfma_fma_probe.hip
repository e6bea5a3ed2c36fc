// mfma_fma_probe.hip -- is v_mfma_f32_4x4x1_16b_f32's  d = a*b + c  the same function as fmaf(a, b, c) (one rounding, RNE,
// denormals kept)?  The vertical pass of the resampler is a chain of rank-1 updates acc[s] = fmaf(w[s], v, acc[s]); if the
// matrix pipe computes exactly that, it can take the chains off the VALU without changing one bit of the output.
// Build: hipcc --offload-arch=gfx950 -O2 -o mfma_fma_probe mfma_fma_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// One wave: lane L supplies a[L] (A operand) and b[L] (B operand) and c[L][0..3]; 4x4x1 with 16 blocks:
// d[L][i] = a[(L & ~3) + i] * b[L] + c[L][i].
__global__ void probe(const float* a, const float* b, const f32x4* c, f32x4* d_mfma, f32x4* d_fma, int n_waves) {
    const int wave = blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    if (wave >= n_waves) return;
    const int lane = threadIdx.x & 63;
    const int g = wave * 64 + lane;
    const float av = a[g], bv = b[g];
    const f32x4 cv = c[g];
    d_mfma[g] = __builtin_amdgcn_mfma_f32_4x4x1f32(av, bv, cv, 0, 0, 0);
    f32x4 r;
    for (int i = 0; i < 4; ++i) r[i] = __builtin_fmaf(a[wave * 64 + (lane & ~3) + i], bv, cv[i]);
    d_fma[g] = r;
}

int main() {
    const int n_waves = 1 << 16, n = n_waves * 64;
    std::vector<float> a(n), b(n);
    std::vector<f32x4> c(n), dm(n), df(n);
    std::mt19937_64 rng(12345);
    auto bits = [&](uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; };
    for (int i = 0; i < n; ++i) {
        const int kind = (i >> 6) & 7;     // per wave: a data class
        auto rnd_f = [&](int k) -> float {
            switch (k) {
            case 0: return std::ldexp(static_cast<float>(rng() >> 40) / (1 << 24), static_cast<int>(rng() % 8) - 6);       // (0,1) values like the pixel data
            case 1: return (static_cast<float>(rng() >> 40) / (1 << 24) - 0.5f) * 2.0f;                                   // weights in (-1,1)
            case 2: { uint32_t u = static_cast<uint32_t>(rng()); u &= 0x807fffffu; u |= ((rng() % 254) + 1) << 23; return bits(u); }   // any finite normal
            case 3: { uint32_t u = static_cast<uint32_t>(rng()) & 0x807fffffu; return bits(u); }                           // denormals / zeros
            default: return 0.0f;
            }
        };
        switch (kind) {
        case 0: case 1: a[i] = rnd_f(1); b[i] = rnd_f(0); for (int k = 0; k < 4; ++k) c[i][k] = rnd_f(0); break;                 // the resampler's range
        case 2: a[i] = rnd_f(1); b[i] = rnd_f(0); for (int k = 0; k < 4; ++k) c[i][k] = -a[(i & ~3) + 0] * b[i] * (1.0f + 1e-7f * (k + 1)); break;   // (a filled later) cancellation
        case 3: a[i] = rnd_f(2); b[i] = rnd_f(2); for (int k = 0; k < 4; ++k) c[i][k] = rnd_f(2); break;                           // wide exponents (overflow, underflow)
        case 4: a[i] = rnd_f(3); b[i] = rnd_f(1); for (int k = 0; k < 4; ++k) c[i][k] = rnd_f(3); break;                           // denormal operands
        case 5: a[i] = rnd_f(1) * 1e-20f; b[i] = rnd_f(0) * 1e-20f; for (int k = 0; k < 4; ++k) c[i][k] = rnd_f(3); break;         // denormal results
        case 6: a[i] = (rng() & 1) ? 0.0f : -0.0f; b[i] = rnd_f(0); for (int k = 0; k < 4; ++k) c[i][k] = (rng() & 1) ? 0.0f : -0.0f; break;   // signed zeros
        default: a[i] = rnd_f(1); b[i] = rnd_f(0); for (int k = 0; k < 4; ++k) c[i][k] = 0.0f; break;
        }
    }
    // exact-cancellation class: c = -(a*b rounded) so that the fused result is the rounding error of the product
    for (int i = 0; i < n; ++i)
        if (((i >> 6) & 7) == 2)
            for (int k = 0; k < 4; ++k) c[i][k] = -(a[(i & ~3) + k] * b[i]);
    float *da, *db;
    f32x4 *dc, *ddm, *ddf;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 16); hipMalloc(&ddm, n * 16); hipMalloc(&ddf, n * 16);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipMemcpy(dc, c.data(), n * 16, hipMemcpyHostToDevice);
    probe<<<n_waves / 4, 256>>>(da, db, dc, ddm, ddf, n_waves);
    hipMemcpy(dm.data(), ddm, n * 16, hipMemcpyDeviceToHost); hipMemcpy(df.data(), ddf, n * 16, hipMemcpyDeviceToHost);
    if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 2; }
    long mism[8] = {0}, host_mism[8] = {0}, total[8] = {0};
    for (int i = 0; i < n; ++i) {
        const int kind = (i >> 6) & 7;
        for (int k = 0; k < 4; ++k) {
            ++total[kind];
            uint32_t x, y, z;
            const float m = dm[i][k], f = df[i][k], h = std::fmaf(a[(i & ~3) + k], b[i], c[i][k]);
            std::memcpy(&x, &m, 4); std::memcpy(&y, &f, 4); std::memcpy(&z, &h, 4);
            const bool both_nan = std::isnan(m) && std::isnan(f);
            if (x != y && !both_nan) { if (mism[kind]++ < 3) printf("kind %d: mfma %a (%08x) vs v_fma %a (%08x)  a=%a b=%a c=%a\n", kind, m, x, f, y, a[(i & ~3) + k], b[i], c[i][k]); }
            if (y != z && !(std::isnan(f) && std::isnan(h))) ++host_mism[kind];
        }
    }
    const char* names[8] = {"pixel range", "pixel range", "exact cancellation", "wide exponents", "denormal operands", "denormal results", "signed zeros", "c = 0"};
    bool pixel_ok = true;
    for (int k = 0; k < 8; ++k) {
        printf("class %d (%s): %ld values, mfma != v_fma: %ld, v_fma != host fmaf: %ld\n", k, names[k], total[k], mism[k], host_mism[k]);
        if ((k <= 2 || k >= 6) && mism[k]) pixel_ok = false;
    }
    printf("RESULT: mfma_f32_4x4x1 %s fmaf on the resampler's value range\n", pixel_ok ? "EQUALS" : "DIFFERS FROM");
    return 0;
}
