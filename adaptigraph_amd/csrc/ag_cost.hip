// Per-candidate cost kernels: replace reference src/planning/losses.py:4-92 and the particle
// reductions of running_cost (src/planning/plan.py:41-44).  gfx950 only.
//
// All of them are "one workgroup per (candidate, look-ahead step) row, sweep the particles, tree-reduce": HBM/LDS-bound
// pairwise or pointwise passes.  chamfer is the heavy one (N*M distance evaluations per row in both directions): both
// clouds sit in LDS as SoA, every lane owns one point and scans the other cloud through LDS broadcasts - the same
// tiling idea as the edge builder.  No atomics: fixed reduction trees, bit-reproducible.
#include "ag_common.h"

namespace ag {

constexpr int CT = 256;

__device__ __forceinline__ float block_sum(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = CT / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_min(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = CT / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fminf(red[tid], red[tid + o]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    const int tid = threadIdx.x;
    red[tid] = v;
    __syncthreads();
    for (int o = CT / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]);
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

// ---- chamfer(x, y) = mean_j min_i |x_i - y_j| + mean_i min_j |x_i - y_j|        losses.py:4-10
// x (R,N,3); y (By,M,3) with By == 1 (one target for every row, plan.py:146) or By == R; optional validity masks
// (mean_chamfer, losses.py:12-24, keeps only masked-in points of both clouds).
struct ChamferDev {
    const float* x; const float* y; const uint8_t* xm; const uint8_t* ym; float* out;
    int R, N, M, By;
};
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int CH_PT = 4;
// Squared distance from CH_PT owned points (registers) to every point of the other cloud (SoA in LDS, length padded to
// even with points parked at BIG).  The sweep is VALU-bound, so it runs on the packed fp32 pipe: two points of the other
// cloud per step (v_pk_add/mul/fma_f32) and one v_min3_f32 folds both distances - 7 instructions per 2 evaluations.
// A parked point gives dx*dx = +inf, which min() ignores; no select in the loop.
__device__ __forceinline__ void chamfer_sweep(const float* os, int opad, const float (&qx)[CH_PT], const float (&qy)[CH_PT],
                                              const float (&qz)[CH_PT], float (&m)[CH_PT]) {
    const f2* ox = reinterpret_cast<const f2*>(os);
    const f2* oy = reinterpret_cast<const f2*>(os + opad);
    const f2* oz = reinterpret_cast<const f2*>(os + 2 * opad);
    for (int i = 0; i < opad / 2; ++i) {
        const f2 px = ox[i], py = oy[i], pz = oz[i];
#pragma unroll
        for (int k = 0; k < CH_PT; ++k) {
            const f2 dx = px - qx[k], dy = py - qy[k], dz = pz - qz[k];
            const f2 d = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
            m[k] = fminf(fminf(m[k], d.x), d.y);
        }
    }
}
__global__ __launch_bounds__(CT) void k_chamfer(ChamferDev a) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    __shared__ float red[CT];
    const int r = blockIdx.x, tid = threadIdx.x;
    const int Np = (a.N + 1) & ~1, Mp = (a.M + 1) & ~1;
    float* xs = sm;                 // [3][Np]
    float* ys = sm + 3 * Np;        // [3][Mp]
    const float* xr = a.x + (long)r * a.N * 3;
    const float* yr = a.y + (long)(a.By == 1 ? 0 : r) * a.M * 3;
    const uint8_t* xm = a.xm ? a.xm + (long)r * a.N : nullptr;
    const uint8_t* ym = a.ym ? a.ym + (long)(a.By == 1 ? 0 : r) * a.M : nullptr;
    const float BIG = 3.0e38f;
    for (int i = tid; i < Np; i += CT) {
        const bool v = i < a.N && (xm ? xm[i] != 0 : true);   // masked-out points (and the pad) are parked at "infinity"
        xs[i] = v ? xr[3 * i] : BIG; xs[Np + i] = v ? xr[3 * i + 1] : BIG; xs[2 * Np + i] = v ? xr[3 * i + 2] : BIG;
    }
    for (int j = tid; j < Mp; j += CT) {
        const bool v = j < a.M && (ym ? ym[j] != 0 : true);
        ys[j] = v ? yr[3 * j] : BIG; ys[Mp + j] = v ? yr[3 * j + 1] : BIG; ys[2 * Mp + j] = v ? yr[3 * j + 2] : BIG;
    }
    __syncthreads();
    float sum_y = 0.f, cnt_y = 0.f, sum_x = 0.f, cnt_x = 0.f;
    // register tiling: a lane owns CH_PT points of one cloud, so every LDS read of two points of the other cloud feeds
    // 2*CH_PT distance evaluations
    for (int j0 = tid * CH_PT; j0 < a.M; j0 += CT * CH_PT) {     // for every y point the nearest x
        float qx[CH_PT], qy[CH_PT], qz[CH_PT], m[CH_PT];
#pragma unroll
        for (int k = 0; k < CH_PT; ++k) {
            const int j = min(j0 + k, a.M - 1);
            qx[k] = ys[j]; qy[k] = ys[Mp + j]; qz[k] = ys[2 * Mp + j]; m[k] = BIG;
        }
        chamfer_sweep(xs, Np, qx, qy, qz, m);
#pragma unroll
        for (int k = 0; k < CH_PT; ++k)
            if (j0 + k < a.M && qx[k] < BIG) { sum_y += sqrtf(m[k]); cnt_y += 1.f; }
    }
    for (int i0 = tid * CH_PT; i0 < a.N; i0 += CT * CH_PT) {     // for every x point the nearest y
        float qx[CH_PT], qy[CH_PT], qz[CH_PT], m[CH_PT];
#pragma unroll
        for (int k = 0; k < CH_PT; ++k) {
            const int i = min(i0 + k, a.N - 1);
            qx[k] = xs[i]; qy[k] = xs[Np + i]; qz[k] = xs[2 * Np + i]; m[k] = BIG;
        }
        chamfer_sweep(ys, Mp, qx, qy, qz, m);
#pragma unroll
        for (int k = 0; k < CH_PT; ++k)
            if (i0 + k < a.N && qx[k] < BIG) { sum_x += sqrtf(m[k]); cnt_x += 1.f; }
    }
    const float sy = block_sum(sum_y, red), cy = block_sum(cnt_y, red);
    const float sx = block_sum(sum_x, red), cx = block_sum(cnt_x, red);
    if (tid == 0) a.out[r] = sy / cy + sx / cx;
}

// ---- per-row particle statistics of a (R,N,3) state tensor: box_loss (losses.py:26-35) and the x/z bounds that
// running_cost turns into the bounding-box penalty (plan.py:41-51).  out (R,5) = [box_loss, xmin, xmax, zmin, zmax]
struct StatsDev { const float* state; float* out; int R, N; int has_box; float bx0, bx1, bz0, bz1; };
__global__ __launch_bounds__(CT) void k_state_stats(StatsDev a) {
    __shared__ float red[CT];
    const int r = blockIdx.x, tid = threadIdx.x;
    const float* s = a.state + (long)r * a.N * 3;
    float acc = 0.f, xmin = 3.0e38f, xmax = -3.0e38f, zmin = 3.0e38f, zmax = -3.0e38f;
    for (int i = tid; i < a.N; i += CT) {
        const float x = s[3 * i], z = s[3 * i + 2];
        xmin = fminf(xmin, x); xmax = fmaxf(xmax, x); zmin = fminf(zmin, z); zmax = fmaxf(zmax, z);
        if (a.has_box) {
            const float xd = fmaxf(a.bx0 - x, 0.f) + fmaxf(x - a.bx1, 0.f);
            const float zd = fmaxf(a.bz0 - z, 0.f) + fmaxf(z - a.bz1, 0.f);
            acc += sqrtf(xd * xd + zd * zd);
        }
    }
    const float t = block_sum(acc, red);
    const float a0 = block_min(xmin, red), a1 = block_max(xmax, red), a2 = block_min(zmin, red), a3 = block_max(zmax, red);
    if (tid == 0) {
        float* o = a.out + (long)r * 5;
        o[0] = t / (float)a.N; o[1] = a0; o[2] = a1; o[3] = a2; o[4] = a3;
    }
}

// ---- collision penalties (losses.py:37-92).  One workgroup per (b,h).  kind 0 rope, 1 cloth, 2 granular.
// out (B,H,2): [exp(-max(dmin - size, 0)*100), min(dmax, 0.4*ratio)] - the second entry is only meaningful for cloth,
// whose final value needs the batch-global maximum of it (losses.py:62).
struct PenDev {
    const float* state_pred; const float* action; const float* state_init; float* out;
    int B, H, N, kind; float ratio;
};
__global__ __launch_bounds__(CT) void k_penalty(PenDev a) {
    __shared__ float red[CT];
    const int bh = blockIdx.x, b = bh / a.H, h = bh % a.H, tid = threadIdx.x;
    const float* act = a.action + (long)bh * 4;
    // cloth always looks at the initial cloud (losses.py:55); rope / granular at the cloud BEFORE step h (:42-43, :83-84)
    const float* s = (a.kind == 1 || h == 0) ? a.state_init : a.state_pred + ((long)b * a.H + (h - 1)) * a.N * 3;
    const float x0 = act[0], z0 = act[1];
    float px[9], pz[9];
    int npt = 1;
    px[0] = x0; pz[0] = z0;
    if (a.kind == 2) {                                       // 9 points along the pusher blade (losses.py:70-82)
        const float rad = 0.05f * a.ratio;
        const float dx = rad * sinf(act[2]), dz = -rad * cosf(act[2]);
        const float c[9] = {-1.f, -0.75f, -0.5f, -0.25f, 0.f, 0.25f, 0.5f, 0.75f, 1.f};
        npt = 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            px[k] = c[k] < 0.f ? x0 - (-c[k]) * dx : x0 + c[k] * dx;
            pz[k] = c[k] < 0.f ? z0 - (-c[k]) * dz : z0 + c[k] * dz;
        }
    }
    float dmin = 3.0e38f, dmax = 0.f;
    for (int i = tid; i < a.N; i += CT) {
        const float sx = s[3 * i], sz = s[3 * i + 2];
        for (int k = 0; k < npt; ++k) {
            const float ex = px[k] - sx, ez = pz[k] - sz;
            const float d = sqrtf(ex * ex + ez * ez);
            dmin = fminf(dmin, d); dmax = fmaxf(dmax, d);
        }
    }
    const float mn = block_min(dmin, red), mx = block_max(dmax, red);
    if (tid == 0) {
        const float size = (a.kind == 1 ? 0.005f : 0.02f) * a.ratio;
        a.out[(long)bh * 2 + 0] = expf(-fmaxf(mn - size, 0.f) * 100.f);
        a.out[(long)bh * 2 + 1] = fminf(mx, 0.4f * a.ratio);
    }
}

// ---- what is left of running_cost once the particle reductions are done (plan.py:35-53), in ONE launch instead of ~30
// element-wise torch kernels on (B,H) tensors (a planner call evaluates 80 batches: the launches were 15 % of it):
//   error_weight = fp32(2 / (double(max error) + 1e-6))                                               plan.py:37
//   box_penalty[b,h] = max over the four sides of exp(-max(side violation, 0) * 100)                    plan.py:41-51
//   reward[b] = -error_weight * error[b,-1] - 5 * mean_h penalty[b,h] - 5 * mean_h box_penalty[b,h]    plan.py:53
// fp32 operations in the reference's order; the mean is sum * fp32(1/H) as torch's mean kernel forms it.  One workgroup:
// it needs the batch-global error maximum first (given by the caller when the batch is sharded and the maximum all-reduced).
struct RewardDev {
    const float* error; const float* pen; const float* stats; const float* emax; float* out;
    int B, H; float bx0, bx1, bz0, bz1;
};
constexpr int RW = 1024;
__global__ __launch_bounds__(RW) void k_reward(RewardDev a) {
    __shared__ float red[RW];
    const int tid = threadIdx.x;
    float mx;
    if (a.emax) mx = a.emax[0];
    else {
        float m = -3.4e38f;
        bool nan = false;
        for (long i = tid; i < (long)a.B * a.H; i += RW) { const float e = a.error[i]; nan |= e != e; m = fmaxf(m, e); }
        red[tid] = nan ? __builtin_nanf("") : m;
        __syncthreads();
        for (int o = RW / 2; o > 0; o >>= 1) {
            if (tid < o) { const float x = red[tid], y = red[tid + o]; red[tid] = (x != x || y != y) ? __builtin_nanf("") : fmaxf(x, y); }
            __syncthreads();
        }
        mx = red[0];                                         // (torch.max propagates NaN: so does this)
    }
    const float ew = (float)(2.0 / ((double)mx + 1e-6));
    const float inv_h = (float)(1.0 / (double)a.H);
    for (int b = tid; b < a.B; b += RW) {
        float ps = 0.f, bs = 0.f;
        for (int h = 0; h < a.H; ++h) {
            const float* st = a.stats + ((long)b * a.H + h) * 5;
            const float v0 = fmaxf(st[1] - a.bx0, 0.f), v1 = fmaxf(a.bx1 - st[2], 0.f);
            const float v2 = fmaxf(st[3] - a.bz0, 0.f), v3 = fmaxf(a.bz1 - st[4], 0.f);
            const float e0 = expf(-v0 * 100.0f), e1 = expf(-v1 * 100.0f), e2 = expf(-v2 * 100.0f), e3 = expf(-v3 * 100.0f);
            bs += fmaxf(fmaxf(e0, e1), fmaxf(e2, e3));
            ps += a.pen[(long)b * a.H + h];
        }
        a.out[b] = -ew * a.error[(long)b * a.H + a.H - 1] - 5.0f * (ps * inv_h) - 5.0f * (bs * inv_h);
    }
}
hipError_t launch_reward(const float* error, const float* pen, const float* stats, const float* emax, const double* bbox4, int B,
                         int H, float* out, hipStream_t st) {
    RewardDev a{error, pen, stats, emax, out, B, H, (float)bbox4[0], (float)bbox4[1], (float)bbox4[2], (float)bbox4[3]};
    hipLaunchKernelGGL(k_reward, dim3(1), dim3(RW), 0, st, a);
    return hipGetLastError();
}
// cloth_penalty's tail (losses.py:62-63): 1 - e0 - 0.2 * e1 / max_batch(e1), from the (B,H,2) output of k_penalty
__global__ __launch_bounds__(RW) void k_cloth_combine(const float* __restrict__ raw, const float* __restrict__ dmax_in, long n,
                                                      float* __restrict__ out) {
    __shared__ float red[RW];
    const int tid = threadIdx.x;
    float mx;
    if (dmax_in) mx = dmax_in[0];
    else {
        float m = -3.4e38f;
        for (long i = tid; i < n; i += RW) m = fmaxf(m, raw[2 * i + 1]);
        red[tid] = m;
        __syncthreads();
        for (int o = RW / 2; o > 0; o >>= 1) { if (tid < o) red[tid] = fmaxf(red[tid], red[tid + o]); __syncthreads(); }
        mx = red[0];
    }
    for (long i = tid; i < n; i += RW) out[i] = 1.0f - raw[2 * i] - (raw[2 * i + 1] / mx) * 0.2f;
}
hipError_t launch_cloth_combine(const float* raw, const float* dmax, long n, float* out, hipStream_t st) {
    hipLaunchKernelGGL(k_cloth_combine, dim3(1), dim3(RW), 0, st, raw, dmax, n, out);
    return hipGetLastError();
}

hipError_t launch_chamfer(const float* x, const float* y, const uint8_t* xm, const uint8_t* ym, int R, int N, int M,
                          int By, float* out, hipStream_t st) {
    ChamferDev a{x, y, xm, ym, out, R, N, M, By};
    const size_t lds = (size_t)(3 * ((N + 1) & ~1) + 3 * ((M + 1) & ~1)) * 4;
    static unsigned long long attr_devices = 0;              // per-device function attribute (see ag_edges.hip)
    int dev_id = 0;
    if (hipGetDevice(&dev_id) != hipSuccess) dev_id = 0;
    if (dev_id >= 64 || !(attr_devices >> dev_id & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_chamfer),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 2048);
        if (e != hipSuccess) return e;
        if (dev_id < 64) attr_devices |= 1ull << dev_id;
    }
    hipLaunchKernelGGL(k_chamfer, dim3(R), dim3(CT), lds, st, a);
    return hipGetLastError();
}
hipError_t launch_state_stats(const float* state, int R, int N, const float* box4, float* out, hipStream_t st) {
    StatsDev a{state, out, R, N, box4 ? 1 : 0, box4 ? box4[0] : 0.f, box4 ? box4[1] : 0.f, box4 ? box4[2] : 0.f,
               box4 ? box4[3] : 0.f};
    hipLaunchKernelGGL(k_state_stats, dim3(R), dim3(CT), 0, st, a);
    return hipGetLastError();
}
hipError_t launch_penalty(const float* state_pred, const float* action, const float* state_init, int B, int H, int N,
                          int kind, float ratio, float* out, hipStream_t st) {
    PenDev a{state_pred, action, state_init, out, B, H, N, kind, ratio};
    hipLaunchKernelGGL(k_penalty, dim3(B * H), dim3(CT), 0, st, a);
    return hipGetLastError();
}
size_t chamfer_max_points() { return (160 * 1024 - 2048) / 12 - 2; }   // both clouds, each padded to even length

}  // namespace ag
