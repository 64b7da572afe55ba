// Stand-alone probe (not part of the library), companion of gather_order.hip: does dropping the 40-B row padding pay?
// Activation rows are 150 floats + the bias slot at a 160-float pitch (640 B = 5 whole 128-B lines).  At a 152-float pitch
// (608 B) a row is 4.75 lines: 5 % fewer bytes, but only every fourth row starts on a line boundary, so the eight lanes
// that fetch one 128-B piece of a row touch two lines three times out of four.  Same access shape as the fused gather (8
// rows x 128 B per instruction, C rows slot-indexed and read once, V rows of random near neighbours, one step of look-ahead).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/gather_pitch.hip -o gather_pitch && ./gather_pitch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int K = 6;
template <int PITCH>
__global__ __launch_bounds__(256) void k_gather(const float* __restrict__ C, const float* __restrict__ V,
                                                 const int* __restrict__ snd, float* __restrict__ out, int n_recv) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, rr = lane >> 3, c = lane & 7;
    const bool last_on = PITCH == 160 || c < 6;              // the fifth piece of a 152-float row is 96 B
    f4 total = {0, 0, 0, 0};
    for (int tile = blockIdx.x; (long)tile * 128 < n_recv; tile += gridDim.x) {
        const int r0 = tile * 128 + wave * 32;
#pragma unroll 1
        for (int p = 0; p < 4; ++p) {
            const int i = min(r0 + 8 * p + rr, n_recv - 1);
            const int* sp = snd + (long)i * K;
            int s[K];
#pragma unroll
            for (int k = 0; k < K; ++k) s[k] = sp[k];
            f4 cb[2][5], vb[2][5], acc[5];
#pragma unroll
            for (int t = 0; t < 5; ++t) acc[t] = f4{0, 0, 0, 0};
            auto issueCV = [&](int k, f4* dc, f4* dv) {
                const float* cp = C + ((long)i * K + min(k, K - 1)) * PITCH + 4 * c;
                const float* vp = V + (long)s[min(k, K - 1)] * PITCH + 4 * c;
#pragma unroll
                for (int t = 0; t < 5; ++t) {
                    const bool on = t < 4 || last_on;
                    dc[t] = on ? *reinterpret_cast<const f4*>(cp + 32 * t) : f4{0, 0, 0, 0};
                    dv[t] = on ? *reinterpret_cast<const f4*>(vp + 32 * t) : f4{0, 0, 0, 0};
                }
            };
            issueCV(0, cb[0], vb[0]);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                issueCV(k + 1, cb[(k + 1) & 1], vb[(k + 1) & 1]);
#pragma unroll
                for (int t = 0; t < 5; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[t][e] += fmaxf(cb[k & 1][t][e] + vb[k & 1][t][e], 0.f);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int t = 0; t < 5; ++t) total += acc[t];
        }
    }
    out[(long)blockIdx.x * blockDim.x + threadIdx.x] = total[0] + total[1] + total[2] + total[3];
}
template <int PITCH>
static void run(const float* C, const float* V, const int* snd, float* out, int n_recv, int wgs_per_cu) {
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k_gather<PITCH>), dim3(grid), dim3(256), 0, 0, C, V, snd, out, n_recv);
    (void)hipEventRecord(e0);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k_gather<PITCH>), dim3(grid), dim3(256), 0, 0, C, V, snd, out, n_recv);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("pitch %d floats (%d B), %d WG/CU: %.3f ms per %d receivers x %d edges\n", PITCH, PITCH * 4, wgs_per_cu, ms, n_recv, K);
}
int main() {
    const int n_recv = 128 * 2026;
    float *C, *V, *out; int* snd;
    (void)hipMalloc(&C, (size_t)n_recv * K * 160 * 4 + 4096); (void)hipMalloc(&V, (size_t)n_recv * 160 * 4 + 4096);
    (void)hipMalloc(&snd, (size_t)n_recv * K * 4); (void)hipMalloc(&out, 256 * 4 * 512 * 4);
    (void)hipMemset(C, 0, (size_t)n_recv * K * 160 * 4 + 4096); (void)hipMemset(V, 0, (size_t)n_recv * 160 * 4 + 4096);
    std::vector<int> h((size_t)n_recv * K);
    unsigned x = 12345;
    for (int i = 0; i < n_recv; ++i)
        for (int k = 0; k < K; ++k) { x = x * 1664525u + 1013904223u; int j = i + (int)(x >> 8) % 401 - 200; h[(size_t)i * K + k] = j < 0 ? 0 : j >= n_recv ? n_recv - 1 : j; }
    (void)hipMemcpy(snd, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep)
        for (int wg = 1; wg <= 2; ++wg) { run<160>(C, V, snd, out, n_recv, wg); run<152>(C, V, snd, out, n_recv, wg); }
    return 0;
}
