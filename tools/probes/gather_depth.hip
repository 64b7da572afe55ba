// Stand-alone probe (not part of the library): how does the look-ahead depth of a row gather shaped like the fused message
// passing (ag_mlp.hip: gather_agg) change what a CU pulls from HBM / L2?  Per wavefront: 32 receivers in 4 passes of 8 rows,
// K edges per receiver; every step loads, for 8 rows at once, the 640-B C row of the edge (slot-indexed: receiver-major,
// 6 rows apart between neighbouring receivers, each row read once) and the 640-B V row of its sender (a random neighbour
// within +-200 rows: reused ~K times, mostly L2 / Infinity Cache), five 128-B tiles each, 8 lanes x 16 B per row.
// DEPTH = number of edge steps in flight while one is summed - AS WRITTEN.  Finding: without a sched_barrier hipcc sinks
// every load next to its use, so all depths compile to no look-ahead and run alike (17.5 GB/s per CU at 4 waves); the
// pinned version is gather_order.hip (32 GB/s per CU).  Kept as the record of what the missing look-ahead costs, and for
// the C-only / V-only / contiguous-C modes.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_depth.hip -o gather_depth && ./gather_depth
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NFP = 160, K = 6;
struct Buf { f4 c[5], v[5]; };
template <int DEPTH, int WAVES, int CMODE, int VOFF>
__global__ __launch_bounds__(64 * WAVES) void k_gather(const float* __restrict__ C, const float* __restrict__ V,
                                                        const int* __restrict__ snd, float* __restrict__ out, int rows_per_wg,
                                                        int n_recv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane >> 3, c = lane & 7;
    f4 total = {0, 0, 0, 0};
    for (int tile = blockIdx.x; (long)tile * rows_per_wg < n_recv; tile += gridDim.x) {
        const int r0 = tile * rows_per_wg + wave * 32;
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int i = min(r0 + 8 * p + rr, n_recv - 1);
            const int* sp = snd + (long)i * K;
            int s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] = sp[k];
            Buf b[DEPTH + 1];
            f4 acc[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = f4{0, 0, 0, 0};
            auto issue = [&](int k, Buf& d) {
                // CMODE 0: the shipped shape (8 rows x 128 B per instruction); 1: the same bytes as contiguous 1-KB pieces
                // (what a receiver-major stream of the wave's C rows would look like); 2: no C loads at all (V only)
                const float* cp = (CMODE == 1 || CMODE == 4) ? C + ((long)(r0 + 8 * p) * K + k * 8) * NFP + lane * 4
                                             : C + ((long)i * K + k) * NFP + 4 * c;
                const float* vp = V + (long)s[k] * NFP + 4 * c;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    d.c[t] = CMODE == 2 ? f4{0, 0, 0, 0} : CMODE >= 3 ? *reinterpret_cast<const f4*>(cp + (CMODE == 4 ? 256 : 32) * t) : __builtin_nontemporal_load(reinterpret_cast<const f4*>(cp + (CMODE == 1 ? 256 : 32) * t));
                    d.v[t] = VOFF ? f4{0, 0, 0, 0} : *reinterpret_cast<const f4*>(vp + 32 * t);
                }
            };
#pragma unroll
            for (int k = 0; k < DEPTH && k < K; ++k) issue(k, b[k % (DEPTH + 1)]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (k + DEPTH < K) issue(k + DEPTH, b[(k + DEPTH) % (DEPTH + 1)]);
                const Buf& d = b[k % (DEPTH + 1)];
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] += fmaxf(d.c[t][e] + d.v[t][e], 0.f);
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) total += acc[t];
        }
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}
template <int DEPTH, int WAVES, int CMODE, int VOFF = 0>
static void run(const float* C, const float* V, const int* snd, float* out, int n_recv, int wgs_per_cu) {
    const int rows_per_wg = 32 * WAVES, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<DEPTH, WAVES, CMODE, VOFF>), dim3(grid), dim3(64 * WAVES), 0, 0, C, V, snd, out, rows_per_wg, n_recv);
    hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_gather<DEPTH, WAVES, CMODE, VOFF>), dim3(grid), dim3(64 * WAVES), 0, 0, C, V, snd, out, rows_per_wg, n_recv);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    const double bytes = (double)n_recv * K * ((CMODE == 2 || VOFF) ? 1 : 2) * 640;
    printf("depth %d, C mode %d%s, %d waves/WG, %d WG/CU: %.3f ms, %.2f TB/s gathered, %.1f GB/s per CU\n", DEPTH, CMODE, VOFF ? " (no V)" : "", WAVES, wgs_per_cu, ms,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256);
}
int main() {
    const int n_recv = 128 * 2026;                       // one 128-candidate launch
    float *C, *V, *out; int* snd;
    hipMalloc(&C, (size_t)n_recv * K * NFP * 4); hipMalloc(&V, (size_t)n_recv * NFP * 4);
    hipMalloc(&snd, (size_t)n_recv * K * 4); hipMalloc(&out, 256 * 4 * 512 * 4);
    hipMemset(C, 0, (size_t)n_recv * K * NFP * 4); hipMemset(V, 0, (size_t)n_recv * NFP * 4);
    std::vector<int> h((size_t)n_recv * K);
    unsigned x = 12345;
    for (int i = 0; i < n_recv; ++i)
        for (int k = 0; k < K; ++k) { x = x * 1664525u + 1013904223u; int j = i + (int)(x >> 8) % 401 - 200; h[(size_t)i * K + k] = j < 0 ? 0 : j >= n_recv ? n_recv - 1 : j; }
    hipMemcpy(snd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<1, 4, 0>(C, V, snd, out, n_recv, 1); run<3, 4, 0>(C, V, snd, out, n_recv, 1); run<1, 4, 0>(C, V, snd, out, n_recv, 2); run<1, 4, 0>(C, V, snd, out, n_recv, 4);
    run<1, 4, 3>(C, V, snd, out, n_recv, 1); run<1, 4, 3>(C, V, snd, out, n_recv, 2);                 // temporal C loads
    run<1, 4, 1>(C, V, snd, out, n_recv, 1); run<1, 4, 1>(C, V, snd, out, n_recv, 2);                 // contiguous C
    run<1, 4, 2>(C, V, snd, out, n_recv, 1); run<1, 4, 2>(C, V, snd, out, n_recv, 2);                 // V only
    run<1, 4, 0, 1>(C, V, snd, out, n_recv, 1); run<3, 4, 0, 1>(C, V, snd, out, n_recv, 1); run<1, 4, 0, 1>(C, V, snd, out, n_recv, 2); run<3, 4, 0, 1>(C, V, snd, out, n_recv, 2);   // C only
    run<1, 4, 1, 1>(C, V, snd, out, n_recv, 1); run<3, 4, 1, 1>(C, V, snd, out, n_recv, 2); run<3, 4, 4, 1>(C, V, snd, out, n_recv, 2); run<3, 4, 4, 1>(C, V, snd, out, n_recv, 4);   // contiguous C only
    return 0;
}
