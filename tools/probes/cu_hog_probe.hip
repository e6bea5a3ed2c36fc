// cu_hog_probe.hip -- stands in for RCCL's gather kernel on a box with ONE GPU (tools/exp_gather_overlap.py).
//
// What it imitates, from librccl.so.1.0.70200's gfx950 code object (llvm-readelf --notes; round 6): ncclDevKernel_Generic_{1,2,4}
// -- the kernel behind grouped send / recv, i.e. behind torch.distributed.gather -- has group_segment_fixed_size 37 664 bytes,
// 248 - 256 VGPRs and one workgroup per channel.  A workgroup of the fused resample kernel takes a CU's LDS (up to 160 KB) and
// 4 waves x 109 - 122 VGPRs on every SIMD, so an RCCL workgroup and a resample workgroup can never share a CU: while a gather
// runs, `channels` CUs are gone.  This kernel takes the same: `wgs` workgroups of 256 lanes with 37 664 bytes of LDS that spin
// for `micros` microseconds of wall clock.
//
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/probes/libcu_hog.so tools/probes/cu_hog_probe.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) cu_hog_kernel(uint64_t ticks, uint32_t* sink) {
    __shared__ uint32_t lds[37664 / 4];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    const uint64_t t0 = wall_clock64();                 // 100 MHz constant clock
    uint32_t acc = 0;
    while (wall_clock64() - t0 < ticks) {
        acc += lds[(threadIdx.x + acc) & 255u];
        __builtin_amdgcn_s_sleep(8);
    }
    if (acc == 0xffffffffu) sink[0] = acc;
}

extern "C" int cu_hog_launch(void* stream, int wgs, int micros, uint32_t* d_sink) {
    hipLaunchKernelGGL(cu_hog_kernel, dim3(wgs), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<uint64_t>(micros) * 100u, d_sink);
    return static_cast<int>(hipGetLastError());
}
