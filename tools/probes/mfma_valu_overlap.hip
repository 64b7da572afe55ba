// Stand-alone probe (not part of the library): do exact-fp32 MFMAs (v_mfma_f32_32x32x2_f32) of one wavefront and packed fp32
// FMAs (v_pk_fma_f32) of ANOTHER wavefront on the same SIMD execute concurrently, or do they share one pipe?  The guide prices
// both at 64 FLOP/clk/SIMD.  Two wavefronts per SIMD (512-thread workgroups, one per CU): mode 0 = both run the MFMA loop,
// mode 1 = both run the VALU loop, mode 2 = even wavefronts MFMA, odd wavefronts VALU.  Register-only operands, no memory.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o mfma_valu_overlap && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(512) void k_probe(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);      // waves 0-3 / 4-7 land on SIMDs 0-3 each: one of each kind per SIMD
    float r = 0.f;
    if (do_mfma) {
        f32x16 acc[5];
        for (int m = 0; m < 5; ++m) for (int i = 0; i < 16; ++i) acc[m][i] = 0.f;
        float a = 1.0f + threadIdx.x * 1e-6f, b = 0.5f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int m = 0; m < 5; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[m], 0, 0, 0);
        }
        for (int m = 0; m < 5; ++m) for (int i = 0; i < 16; ++i) r += acc[m][i];
    } else {
        f32x2 acc[20];
        for (int i = 0; i < 20; ++i) acc[i] = f32x2{0.f, 0.f};
        f32x2 w = {1.0f + threadIdx.x * 1e-6f, 0.999f}, x = {0.5f, 0.25f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u)          // 20 MFMAs x 64 cycles = 1280 cycles per iteration; 320 pk_fma x 4 cycles = 1280
#pragma unroll
                for (int i = 0; i < 20; ++i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(w), "v"(x));
        }
        for (int i = 0; i < 20; ++i) r += acc[i][0] + acc[i][1];
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = r;
}
int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode) {
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(k_probe, dim3(256), dim3(512), 0, 0, out, iters, mode);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double mfma_waves = mode == 0 ? 8 : mode == 2 ? 4 : 0, valu_waves = mode == 1 ? 8 : mode == 2 ? 4 : 0;
        const double flop = 256.0 * iters * (mfma_waves * 20 * 2.0 * 32 * 32 * 2 + valu_waves * 320 * 64 * 2 * 2.0);
        printf("mode %d (%s): %.3f ms, %.1f TFLOP/s fp32\n", mode, mode == 0 ? "MFMA + MFMA" : mode == 1 ? "VALU + VALU" : "MFMA + VALU", ms, flop / ms / 1e9);
    }
    return 0;
}
