// Stand-alone probe (not part of the library), companion of gather_depth.hip: C rows (HBM, read once) and V rows (L2 /
// Infinity Cache hits) gathered by the SAME wavefront.  Vector-memory loads of a wavefront return in issue order, so a
// fast V response queues behind every older, slower C load.  ORDER: 0 = c,v interleaved per 128-B tile (the shipped
// code); 1 = five C loads then five V loads per edge step; 2 = V then C; 3 = V of step k+1, then C of step k+2 (the C
// stream one step further ahead).  SPLIT = 1: even wavefronts load only C, odd ones only V (no mixing at all).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_order.hip -o gather_order && ./gather_order
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NFP = 160, K = 6;
template <int ORDER, int SPLIT>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ C, const float* __restrict__ V,
                                                 const int* __restrict__ snd, float* __restrict__ out, int n_recv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane >> 3, c = lane & 7;
    const bool doC = !SPLIT || (wave & 1) == 0, doV = !SPLIT || (wave & 1) == 1;
    f4 total = {0, 0, 0, 0};
    for (int tile = blockIdx.x; (long)tile * 128 < n_recv; tile += gridDim.x) {
        const int r0 = tile * 128 + (SPLIT ? (wave >> 1) * 64 : wave * 32);
        const int npass = SPLIT ? 8 : 4;
#pragma unroll 1
        for (int p = 0; p < npass; ++p) {
            const int i = min(r0 + 8 * p + rr, n_recv - 1);
            const int* sp = snd + (long)i * K;
            int s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] = sp[k];
            f4 cb[3][5], vb[2][5], acc[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = f4{0, 0, 0, 0};
            auto issueC = [&](int k, f4* d) {
                const float* cp = C + ((long)i * K + min(k, K - 1)) * NFP + 4 * c;
#pragma unroll
                for (int t = 0; t < 5; ++t) d[t] = doC ? *reinterpret_cast<const f4*>(cp + 32 * t) : f4{0, 0, 0, 0};
            };
            auto issueV = [&](int k, f4* d) {
                const float* vp = V + (long)s[min(k, K - 1)] * NFP + 4 * c;
#pragma unroll
                for (int t = 0; t < 5; ++t) d[t] = doV ? *reinterpret_cast<const f4*>(vp + 32 * t) : f4{0, 0, 0, 0};
            };
            auto issueCV = [&](int k, f4* dc, f4* dv) {
                const float* cp = C + ((long)i * K + min(k, K - 1)) * NFP + 4 * c;
                const float* vp = V + (long)s[min(k, K - 1)] * NFP + 4 * c;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    dc[t] = doC ? *reinterpret_cast<const f4*>(cp + 32 * t) : f4{0, 0, 0, 0};
                    dv[t] = doV ? *reinterpret_cast<const f4*>(vp + 32 * t) : f4{0, 0, 0, 0};
                }
            };
            auto sum = [&](const f4* dc, const f4* dv) {
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] += fmaxf(dc[t][e] + dv[t][e], 0.f);
            };
            if (ORDER == 3) {
                issueV(0, vb[0]); issueC(0, cb[0]); issueC(1, cb[1]);
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    issueV(k + 1, vb[(k + 1) & 1]);
                    issueC(k + 2, cb[(k + 2) % 3]);
                    sum(cb[k % 3], vb[k & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            } else {
                if (ORDER == 0) issueCV(0, cb[0], vb[0]);
                else if (ORDER == 1) { issueC(0, cb[0]); issueV(0, vb[0]); }
                else { issueV(0, vb[0]); issueC(0, cb[0]); }
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    if (ORDER == 0) issueCV(k + 1, cb[(k + 1) & 1], vb[(k + 1) & 1]);
                    else if (ORDER == 1) { issueC(k + 1, cb[(k + 1) & 1]); issueV(k + 1, vb[(k + 1) & 1]); }
                    else { issueV(k + 1, vb[(k + 1) & 1]); issueC(k + 1, cb[(k + 1) & 1]); }
                    sum(cb[k & 1], vb[k & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) total += acc[t];
        }
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}
template <int ORDER, int SPLIT>
static void run(const float* C, const float* V, const int* snd, float* out, int n_recv, int wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<ORDER, SPLIT>), dim3(grid), dim3(256), 0, 0, C, V, snd, out, n_recv);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_gather<ORDER, SPLIT>), dim3(grid), dim3(256), 0, 0, C, V, snd, out, n_recv);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)n_recv * K * 2 * 640;
    printf("order %d split %d, %d WG/CU: %.3f ms, %.2f TB/s gathered, %.1f GB/s per CU\n", ORDER, SPLIT, wgs_per_cu, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
    const int n_recv = 128 * 2026;
    float *C, *V, *out; int* snd;
    (void)hipMalloc(&C, (size_t)n_recv * K * NFP * 4); (void)hipMalloc(&V, (size_t)n_recv * NFP * 4);
    (void)hipMalloc(&snd, (size_t)n_recv * K * 4); (void)hipMalloc(&out, 256 * 4 * 512 * 4);
    (void)hipMemset(C, 0, (size_t)n_recv * K * NFP * 4); (void)hipMemset(V, 0, (size_t)n_recv * NFP * 4);
    std::vector<int> h((size_t)n_recv * K);
    unsigned x = 12345;
    for (int i = 0; i < n_recv; ++i)
        for (int k = 0; k < K; ++k) { x = x * 1664525u + 1013904223u; int j = i + (int)(x >> 8) % 401 - 200; h[(size_t)i * K + k] = j < 0 ? 0 : j >= n_recv ? n_recv - 1 : j; }
    (void)hipMemcpy(snd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<0, 0>(C, V, snd, out, n_recv, 1); run<1, 0>(C, V, snd, out, n_recv, 1); run<2, 0>(C, V, snd, out, n_recv, 1); run<3, 0>(C, V, snd, out, n_recv, 1);
    run<0, 0>(C, V, snd, out, n_recv, 2); run<1, 0>(C, V, snd, out, n_recv, 2); run<2, 0>(C, V, snd, out, n_recv, 2); run<3, 0>(C, V, snd, out, n_recv, 2);
    run<0, 1>(C, V, snd, out, n_recv, 1); run<0, 1>(C, V, snd, out, n_recv, 2);
    return 0;
}
