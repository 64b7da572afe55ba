// build: /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 wg_placement.hip -o wg_placement   (then: gpurun -- tools/probes/wg_placement)
// Probe: where do the workgroups of a chain-shaped launch (256 threads, 72 KB LDS -> two per CU) land, and can a
// workgroup tell whether it is the first or the second on its CU?  Prints per block: XCC id, HW_ID fields, LDS_ALLOC.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <map>
__global__ __launch_bounds__(256, 2) void k(unsigned long long* o, int spin) {
    __shared__ float lds[72192 / 4];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    unsigned a = __builtin_amdgcn_s_getreg((6) | (0 << 6) | (31 << 11));    // HW_REG_LDS_ALLOC
    unsigned b = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));    // HW_REG_HW_ID
    unsigned x = __builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));   // HW_REG_XCC_ID
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) { __builtin_amdgcn_s_sleep(10); }
    if (threadIdx.x == 0) { o[blockIdx.x * 4] = t0; o[blockIdx.x * 4 + 1] = a; o[blockIdx.x * 4 + 2] = b; o[blockIdx.x * 4 + 3] = x; }
    if (lds[threadIdx.x] < 0) o[0] = 0;
}
int main() {
    const int nb = 2026;
    unsigned long long* d;
    hipMalloc(&d, nb * 32);
    hipLaunchKernelGGL(k, dim3(nb), dim3(256), 0, 0, d, 2000);   // 20 us at 100 MHz
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(nb * 4);
    hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost);
    unsigned long long tmin = ~0ull;
    for (int i = 0; i < nb; ++i) tmin = h[i * 4] < tmin ? h[i * 4] : tmin;
    std::map<unsigned long long, int> per_cu_first;
    int lds0 = 0, lds1 = 0, firstwave = 0;
    for (int i = 0; i < nb; ++i) {
        unsigned a = (unsigned)h[i * 4 + 1], b = (unsigned)h[i * 4 + 2], x = (unsigned)h[i * 4 + 3];
        if (i < 24 || (i >= 250 && i < 262) || (i >= 506 && i < 520))
            printf("blk %4d t=%6llu lds_alloc=%08x (base %u size %u) hw_id=%08x cu=%u sh=%u se=%u wave=%u simd=%u xcc=%u\n", i, h[i * 4] - tmin, a, a & 0xff,
                   (a >> 12) & 0x1ff, b, (b >> 8) & 0xf, (b >> 12) & 1, (b >> 13) & 7, b & 0xf, (b >> 4) & 3, x & 0xf);
        if (h[i * 4] - tmin < 500) { ++firstwave; if ((a & 0xff) == 0) ++lds0; else ++lds1; }
    }
    printf("first wave (t < 5us): %d blocks, lds base == 0: %d, != 0: %d\n", firstwave, lds0, lds1);
    // among first-wave blocks: block index ranges by lds base
    int lo0 = 1 << 30, hi0 = -1, lo1 = 1 << 30, hi1 = -1;
    for (int i = 0; i < nb; ++i) if (h[i * 4] - tmin < 500) {
        if ((h[i * 4 + 1] & 0xff) == 0) { lo0 = i < lo0 ? i : lo0; hi0 = i > hi0 ? i : hi0; } else { lo1 = i < lo1 ? i : lo1; hi1 = i > hi1 ? i : hi1; }
    }
    printf("block index range with lds base 0: [%d, %d]; with base != 0: [%d, %d]\n", lo0, hi0, lo1, hi1);
    return 0;
}
